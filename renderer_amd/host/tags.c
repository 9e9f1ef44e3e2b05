/*
 * Identity tags of the three shader families for the GPU build.
 *
 * program_create() recognises a family by the addresses of its vertex and
 * fragment shader functions (the reference passes blinn_/pbr_/skybox_*_shader,
 * blinn_shader.c:372, pbr_shader.c:608, skybox_shader.c:85).  The device
 * pipeline implements those shaders itself (csrc/shade.h, csrc/kernels.h), so
 * the host-side functions exist only to be pointed at.  Calling one is a bug:
 * this library has no CPU shading path.
 */
#include <stdio.h>
#include <stdlib.h>
#include "../../include/mi355r.h"

static rz_vec4 never(const char *name) {
    rz_vec4 zero = {0, 0, 0, 0};
    fprintf(stderr, "mi355r: %s was called on the host; shaders run on the GPU only\n", name);
    abort();
    return zero;
}

rz_vec4 blinn_vertex_shader(void *a, void *v, void *u) { (void)a; (void)v; (void)u; return never("blinn_vertex_shader"); }
rz_vec4 blinn_fragment_shader(void *v, void *u, int *d, int b) { (void)v; (void)u; (void)d; (void)b; return never("blinn_fragment_shader"); }
rz_vec4 pbr_vertex_shader(void *a, void *v, void *u) { (void)a; (void)v; (void)u; return never("pbr_vertex_shader"); }
rz_vec4 pbr_fragment_shader(void *v, void *u, int *d, int b) { (void)v; (void)u; (void)d; (void)b; return never("pbr_fragment_shader"); }
rz_vec4 skybox_vertex_shader(void *a, void *v, void *u) { (void)a; (void)v; (void)u; return never("skybox_vertex_shader"); }
rz_vec4 skybox_fragment_shader(void *v, void *u, int *d, int b) { (void)v; (void)u; (void)d; (void)b; return never("skybox_fragment_shader"); }
