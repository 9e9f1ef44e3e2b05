// Instruction count of a skybox fragment by group: every group of the per-fragment work of
// rz_sky_kernel's one-triangle-per-wave path as a kernel of its own (inputs from memory, one store),
// compiled with the library's flags; tools/isa/sky_groups.sh counts the VALU instructions of each and
// subtracts the empty kernel's.
#include "../../renderer_amd/csrc/kernels.h"

#define IO const float *__restrict__ in, float *__restrict__ out
#define LANE (blockIdx.x * 64 + threadIdx.x)

__global__ void g0_empty(IO) { out[LANE] = in[LANE]; }

__global__ void g1_weights(IO, TriRec r) {              // graphics.c:385-397 at the pixel centre
    float w0, w1, w2;
    pixel_weights(r, (int)in[LANE], (int)in[LANE + 64], w0, w1, w2);
    out[LANE] = w0; out[LANE + 64] = w1; out[LANE + 128] = w2;
}

__global__ void g2_perspective(IO, TriRec r) {          // graphics.c:426-429: 3 products, a sum, a correctly rounded reciprocal
    float w0 = in[LANE], w1 = in[LANE + 64], w2 = in[LANE + 128];
    float q0 = r.rw[0] * w0, q1 = r.rw[1] * w1, q2 = r.rw[2] * w2;
    out[LANE] = 1 / (q0 + q1 + q2);
    out[LANE + 64] = q0; out[LANE + 128] = q1; out[LANE + 192] = q2;
}

__global__ void g3_varyings(IO, const float *m0, const float *m1, const float *m2) {   // graphics.c:431-434, 3 floats
    float v[4];
    interpolate_lane<3>(m0, m1, m2, in[LANE], in[LANE + 64], in[LANE + 128], in[LANE + 192], v);
    out[LANE] = v[0]; out[LANE + 64] = v[1]; out[LANE + 128] = v[2];
}

__global__ void g4_cube_face(IO, int *face) {           // texture.c:210-255 incl. the two correctly rounded divisions
    float s, t;
    face[LANE] = cube_face(mk3(in[LANE], in[LANE + 64], in[LANE + 128]), &s, &t);
    out[LANE] = s; out[LANE + 64] = t;
}

__global__ void g5_texel_address(IO, DevTex tex, uint32_t *cell) {  // texture.c:160-167 + the tiled layout
    float s = in[LANE], t = 1 - in[LANE + 64];
    cell[LANE] = texel_cell(tex.w, tex.h, tex.flags, s - floorf(s), t - floorf(t));
}

__global__ void g6_fetch_decode(const uint32_t *__restrict__ cell, float *__restrict__ out, const DrawRec *D, const int *face) {
    DevTex f = D->maps[0];
    f.data = D->maps[face[LANE]].data;
    f4 c = decode_texel<false>(f, load_cell<false>(f.data, f.flags, cell[LANE]));
    out[LANE] = c.x; out[LANE + 64] = c.y; out[LANE + 128] = c.z;
}

__global__ void g7_saturate_pack(IO, uint32_t *color) { // graphics.c:452, 466-468 with maths.c:33
    f4 c = mk4(d_sat(in[LANE]), d_sat(in[LANE + 64]), d_sat(in[LANE + 128]), 1.0f);
    color[LANE] = pack_rgb(c, color[LANE]);
}

__global__ void g8_whole_fragment(IO, TriRec m, const float *m0, const float *m1, const float *m2, const DrawRec *D, uint32_t *color) {
    float w0, w1, w2, v[4];
    pixel_weights(m, (int)in[LANE], (int)in[LANE + 64], w0, w1, w2);
    const float q0 = m.rw[0] * w0, q1 = m.rw[1] * w1, q2 = m.rw[2] * w2;
    const float norm = 1 / (q0 + q1 + q2);
    interpolate_lane<3>(m0, m1, m2, q0, q1, q2, norm, v);
    f4 c = fs_skybox(*D, v);
    c = mk4(d_sat(c.x), d_sat(c.y), d_sat(c.z), d_sat(c.w));
    color[LANE] = pack_rgb(c, color[LANE]);
}

// the fragment as the sky kernel's own-visibility path shades it: weights kept from the depth test, display bytes
__global__ void g9_bytes_fragment(IO, TriRec m, const float *m0, const float *m1, const float *m2, const DrawRec *D, uint32_t *color) {
    float v[4];
    const float w0 = in[LANE], w1 = in[LANE + 64], w2 = in[LANE + 128];
    const float q0 = m.rw[0] * w0, q1 = m.rw[1] * w1, q2 = m.rw[2] * w2;
    const float norm = 1 / (q0 + q1 + q2);
    interpolate_lane<3>(m0, m1, m2, q0, q1, q2, norm, v);
    color[LANE] = sky_fragment(*D, v, color[LANE]);
}
