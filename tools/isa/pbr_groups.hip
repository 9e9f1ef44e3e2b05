// Instruction count of a pbr fragment by group, each group a kernel of its own (inputs from memory, a few
// stores), compiled with the library's flags; tools/isa/pbr_groups.sh counts the instructions of each.
#include "../../renderer_amd/csrc/kernels.h"

#define IO const float *__restrict__ in, float *__restrict__ out
#define LANE (blockIdx.x * 64 + threadIdx.x)

__global__ void p0_empty(IO) { out[LANE] = in[LANE]; }

__global__ void p1_varyings(IO, const float *m0, const float *m1, const float *m2) {   // graphics.c:431-434, 21 floats
    float v[24];
    interpolate_lane<21>(m0, m1, m2, in[LANE], in[LANE + 64], in[LANE + 128], in[LANE + 192], v);
    for (int i = 0; i < 21; i++) out[LANE + 64 * i] = v[i];
}

__global__ void p2_surface(IO, const DrawRec *D) {     // pbr_shader.c:148-264: six texels, decode, normal map
    float v[24];
    for (int i = 0; i < 21; i++) v[i] = in[LANE + 64 * i];
    Surface s = pbr_surface<true>(*D, v, false);
    out[LANE] = s.diffuse.x + s.diffuse.y + s.diffuse.z + s.specular.x + s.specular.y + s.specular.z + s.alpha + s.roughness +
                s.normal.x + s.normal.y + s.normal.z + s.occlusion + s.emission.x + s.emission.y + s.emission.z;
}

__global__ void p3_ambient(IO, const DrawRec *D) {     // pbr_shader.c:281-309
    Surface s;
    s.diffuse = mk3(in[LANE], in[LANE + 64], in[LANE + 128]); s.specular = mk3(in[LANE + 192], in[LANE + 256], in[LANE + 320]);
    s.roughness = in[LANE + 384]; s.occlusion = in[LANE + 448];
    f3 n = mk3(in[LANE + 512], in[LANE + 576], in[LANE + 640]), view = mk3(in[LANE + 704], in[LANE + 768], in[LANE + 832]);
    f3 c = pbr_ambient<true>(s, D->ibl, n, view);
    out[LANE] = c.x; out[LANE + 64] = c.y; out[LANE + 128] = c.z;
}

__global__ void p4_light(IO) {                          // pbr_shader.c:334-393
    Surface s;
    s.diffuse = mk3(in[LANE], in[LANE + 64], in[LANE + 128]); s.specular = mk3(in[LANE + 192], in[LANE + 256], in[LANE + 320]);
    s.roughness = in[LANE + 384];
    f3 n = mk3(in[LANE + 512], in[LANE + 576], in[LANE + 640]), view = mk3(in[LANE + 704], in[LANE + 768], in[LANE + 832]);
    f3 light = mk3(in[LANE + 896], in[LANE + 960], in[LANE + 1024]);
    f3 c = pbr_light(s, light, n, view);
    out[LANE] = c.x; out[LANE + 64] = c.y; out[LANE + 128] = c.z;
}

__global__ void p5_display(IO) {                        // pbr_shader.c:401-406: aces + pow(x, 1/2.2), three channels
    f4 c = display_encode(mk3(in[LANE], in[LANE + 64], in[LANE + 128]), 1.0f);
    out[LANE] = c.x; out[LANE + 64] = c.y; out[LANE + 128] = c.z;
}

__global__ void p6_whole(IO, const DrawRec *D, uint32_t *color) {
    float v[24];
    for (int i = 0; i < 21; i++) v[i] = in[LANE + 64 * i];
    bool discard = false;
    f4 c = fs_pbr<true>(*D, v, false, &discard);
    c = mk4(d_sat(c.x), d_sat(c.y), d_sat(c.z), d_sat(c.w));
    color[LANE] = pack_rgb(c, color[LANE]);
}
