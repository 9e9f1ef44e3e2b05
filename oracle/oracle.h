/*
 * TEST INFRASTRUCTURE -- declarations private to the CPU oracle build.
 * Nothing under renderer_amd/ includes this file.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include "../include/mi355r.h"

/* the three shader pairs (oracle/shaders.c) */
rz_vec4 blinn_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 blinn_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);
rz_vec4 pbr_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 pbr_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);
rz_vec4 skybox_vertex_shader(void *attribs, void *varyings, void *uniforms);
rz_vec4 skybox_fragment_shader(void *varyings, void *uniforms, int *discard, int backface);

/* varying blocks (blinn_shader.h:16-21, pbr_shader.h:29-37, skybox_shader.h:12-14) */
typedef struct {
    rz_vec3 world_position, depth_position;
    rz_vec2 texcoord;
    rz_vec3 normal;
} o_blinn_varyings;

typedef struct {
    rz_vec3 world_position, depth_position;
    rz_vec4 clip_position;
    rz_vec3 world_normal, world_tangent, world_bitangent;
    rz_vec2 texcoord;
} o_pbr_varyings;

typedef struct { rz_vec3 direction; } o_skybox_varyings;

/* workload counters, accumulated over every draw since the last reset:
 * triangles submitted, triangles rasterised after clip/cull, bounding-box
 * pixel tests, covered pixels, fragments shaded (graphics.c:437 calls) */
typedef struct {
    unsigned long long triangles, prims, tests, covered, shaded;
} oracle_stats;

/* CPU binning oracle (oracle/binning.c): the ordered per-bin triangle lists
 * that follow from the triangles' pixel boxes.  counts[bins]; entries (may be
 * NULL) receives {triangle, fan index} pairs; returns the number of entries. */
size_t oracle_bin(const unsigned *boxes, const unsigned *fans, int num_tris, int width, int height, int edge,
                  unsigned *counts, unsigned *entries);

void oracle_get_stats(oracle_stats *out);
void oracle_reset_stats(void);

#endif
