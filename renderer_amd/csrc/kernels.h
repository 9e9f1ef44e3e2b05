/*
 * The gfx950 kernels of the triangle pipeline, one pass (= draws into one
 * framebuffer) per launch sequence:
 *
 *   rz_vertex_kernel   vertex shaders incl. skeletal skinning, 1 thread / vertex
 *   rz_setup_kernel    clip -> fan -> cull -> viewport -> bbox, 1 thread / triangle
 *   rz_bin_kernel      ordered 64x64-pixel bin lists, 1 workgroup / bin
 *   rz_raster_kernel   coverage, early-Z, varyings, fragment shaders, blend;
 *                      1 wave / 8x8 pixels, depth + colour held in registers
 *   rz_clear_kernel, rz_depth_to_texture_kernel, rz_color_to_texture_kernel,
 *   rz_texture_tile_kernel   framebuffer/texture utilities
 *
 * All arithmetic that decides coverage, depth or colour follows the reference
 * expression order (renderer/core/graphics.c, cited inline).
 */
#ifndef RZ_KERNELS_H
#define RZ_KERNELS_H

#include "shade.h"

/* largest d with first_tri[d] <= t */
RZ_DEV int find_draw(const uint32_t *first_tri, int num_draws, uint32_t t) {
    int lo = 0, hi = num_draws - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (first_tri[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* ========================================================================
 * K1  vertex shaders (blinn_shader.c:9-99, pbr_shader.c:10-111,
 *     skybox_shader.c:8-19)
 * ======================================================================== */

/* sum_i w_i * M_i over the (up to) four influences with w_i > 0, accumulated
 * from zero in influence order (maths.c:526-549 / 362-385) */
template <int N>
RZ_DEV void blend_joints(const float *pool, const float *joint, const float *weight, float *out) {
    for (int k = 0; k < N; k++) out[k] = 0;
    for (int i = 0; i < 4; i++) {
        float w = weight[i];
        if (w > 0) {
            const float *src = pool + (size_t)(int)joint[i] * N;
            for (int k = 0; k < N; k++) out[k] += w * src[k];
        }
    }
}

/* C = A * B, accumulator from zero, k ascending (maths.c:564 / 399) */
template <int N>
RZ_DEV void mat_mul(const float *a, const float *b, float *c) {
    for (int r = 0; r < N; r++) for (int col = 0; col < N; col++) {
        float acc = 0;
        for (int k = 0; k < N; k++) acc += a[r * N + k] * b[k * N + col];
        c[r * N + col] = acc;
    }
}

__global__ void __launch_bounds__(256) rz_vertex_kernel(PassParams P) {
    uint32_t gv = blockIdx.x * 256u + threadIdx.x;
    if (gv >= 3u * (uint32_t)P.num_tris) return;
    int d = find_draw(P.first_tri, P.num_draws, gv / 3u);
    const DrawRec &D = P.draws[d];
    uint32_t local = gv - 3u * P.first_tri[d];
    const float4 *src = (const float4*)((const char*)D.vertices + (size_t)local * 80u);
    float a[20];                    /* position 0-2, texcoord 3-4, normal 5-7, tangent 8-11, joint 12-15, weight 16-19 */
    for (int i = 0; i < 5; i++) {
        float4 q = src[i];
        a[4 * i] = q.x; a[4 * i + 1] = q.y; a[4 * i + 2] = q.z; a[4 * i + 3] = q.w;
    }
    float *vary = P.vary + (size_t)gv * RZ_VARY_STRIDE;
    f4 pos = mk4(a[0], a[1], a[2], 1.0f);
    f4 clip;

    if (D.family == 2) {                                        /* skybox_shader.c:8-19 */
        clip = m4v4(D.camera_vp, pos);
        clip.z = clip.w * (1 - RZ_EPS);
        vary[0] = a[0]; vary[1] = a[1]; vary[2] = a[2];
    } else {
        float model[16];
        if (D.joints) {                                         /* get_model_matrix */
            float skin[16];
            blend_joints<16>(D.joints, a + 12, a + 16, skin);
            mat_mul<4>(D.model, skin, model);
        } else {
            for (int k = 0; k < 16; k++) model[k] = D.model[k];
        }
        f4 world = m4v4(model, pos);
        f4 light_space = m4v4(D.light_vp, world);
        if (D.flags & RZ_DRAW_SHADOW_PASS) {                    /* shadow_vertex_shader */
            clip = light_space;
            int tc = D.family == 0 ? 6 : 19;
            vary[tc] = a[3]; vary[tc + 1] = a[4];
        } else {
            float nmat[9];
            if (D.joint_normals) {                              /* get_normal_matrix */
                float skin[9];
                blend_joints<9>(D.joint_normals, a + 12, a + 16, skin);
                mat_mul<3>(D.normal, skin, nmat);
            } else {
                for (int k = 0; k < 9; k++) nmat[k] = D.normal[k];
            }
            clip = m4v4(D.camera_vp, world);
            f3 n = m3v3(nmat, mk3(a[5], a[6], a[7]));
            vary[0] = world.x; vary[1] = world.y; vary[2] = world.z;
            vary[3] = light_space.x; vary[4] = light_space.y; vary[5] = light_space.z;
            if (D.family == 0) {                                /* blinn common_vertex_shader */
                n = unit3(n);
                vary[6] = a[3]; vary[7] = a[4];
                vary[8] = n.x; vary[9] = n.y; vary[10] = n.z;
            } else {                                            /* pbr common_vertex_shader */
                vary[6] = clip.x; vary[7] = clip.y; vary[8] = clip.z; vary[9] = clip.w;
                n = unit3(n);
                vary[10] = n.x; vary[11] = n.y; vary[12] = n.z;
                if (D.maps[3].data) {
                    float tm[9] = {model[0], model[1], model[2], model[4], model[5], model[6],
                                   model[8], model[9], model[10]};
                    f3 t = unit3(m3v3(tm, mk3(a[8], a[9], a[10])));
                    f3 b = scale3(cross3(n, t), a[11]);
                    vary[13] = t.x; vary[14] = t.y; vary[15] = t.z;
                    vary[16] = b.x; vary[17] = b.y; vary[18] = b.z;
                } else {
                    /* the reference leaves tangent/bitangent untouched here;
                     * nothing reads them without a normal map */
                    vary[13] = 0; vary[14] = 0; vary[15] = 0;
                    vary[16] = 0; vary[17] = 0; vary[18] = 0;
                }
                vary[19] = a[3]; vary[20] = a[4];
            }
        }
    }
    ((float4*)P.clip)[gv] = make_float4(clip.x, clip.y, clip.z, clip.w);
}

/* ========================================================================
 * K2  triangle setup: clip (graphics.c:161-313), fan + cull (:584-607,
 *     :324-332, :494), viewport (:338-343), bounding box (:355-364)
 * ======================================================================== */

enum { PLANE_W, PLANE_PX, PLANE_NX, PLANE_PY, PLANE_NY, PLANE_PZ, PLANE_NZ };

RZ_DEV bool plane_inside(f4 v, int plane) {
    switch (plane) {
    case PLANE_W:  return v.w >= RZ_EPS;
    case PLANE_PX: return v.x <= +v.w;
    case PLANE_NX: return v.x >= -v.w;
    case PLANE_PY: return v.y <= +v.w;
    case PLANE_NY: return v.y >= -v.w;
    case PLANE_PZ: return v.z <= +v.w;
    default:       return v.z >= -v.w;
    }
}

RZ_DEV float plane_crossing(f4 a, f4 b, int plane) {            /* graphics.c:183-203 */
    switch (plane) {
    case PLANE_W:  return (a.w - RZ_EPS) / (a.w - b.w);
    case PLANE_PX: return (a.w - a.x) / ((a.w - a.x) - (b.w - b.x));
    case PLANE_NX: return (a.w + a.x) / ((a.w + a.x) - (b.w + b.x));
    case PLANE_PY: return (a.w - a.y) / ((a.w - a.y) - (b.w - b.y));
    case PLANE_NY: return (a.w + a.y) / ((a.w + a.y) - (b.w + b.y));
    case PLANE_PZ: return (a.w - a.z) / ((a.w - a.z) - (b.w - b.z));
    default:       return (a.w + a.z) / ((a.w + a.z) - (b.w + b.z));
    }
}

RZ_DEV bool fully_visible(f4 v) {                               /* graphics.c:275 */
    return fabsf(v.x) <= v.w && fabsf(v.y) <= v.w && fabsf(v.z) <= v.w;
}

/* Fills rec from three clip-space corners; false when the triangle is culled.
 * Also returns its pixel bounding box (empty boxes have x0 > x1). */
RZ_DEV bool make_tri(const f4 c[3], int width, int height, bool double_sided, uint32_t draw,
                     TriRec &rec, int &x0, int &x1, int &y0, int &y1) {
    f3 ndc[3];
    for (int i = 0; i < 3; i++) ndc[i] = div3(mk3(c[i].x, c[i].y, c[i].z), c[i].w);    /* :486-489 */
    float area = ndc[0].x * ndc[1].y - ndc[0].y * ndc[1].x +
                 ndc[1].x * ndc[2].y - ndc[1].y * ndc[2].x +
                 ndc[2].x * ndc[0].y - ndc[2].y * ndc[0].x;                             /* :328-330 */
    bool backface = area <= 0;
    if (backface && !double_sided) return false;
    float sx[3], sy[3];
    for (int i = 0; i < 3; i++) {
        rec.rw[i] = 1 / c[i].w;                                                         /* :499-501 */
        sx[i] = (ndc[i].x + 1) * 0.5f * (float)width;                                   /* :338-343 */
        sy[i] = (ndc[i].y + 1) * 0.5f * (float)height;
        rec.z[i] = (ndc[i].z + 1) * 0.5f;
    }
    rec.ax = sx[0]; rec.ay = sy[0];
    rec.abx = sx[1] - sx[0]; rec.aby = sy[1] - sy[0];
    rec.acx = sx[2] - sx[0]; rec.acy = sy[2] - sy[0];
    rec.factor = 1 / (rec.abx * rec.acy - rec.aby * rec.acx);                           /* :392 */
    float minx = d_min(d_min(sx[0], sx[1]), sx[2]), maxx = d_max(d_max(sx[0], sx[1]), sx[2]);
    float miny = d_min(d_min(sy[0], sy[1]), sy[2]), maxy = d_max(d_max(sy[0], sy[1]), sy[2]);
    if (!(minx == minx && maxx == maxx && miny == miny && maxy == maxy)) {
        x0 = 1; x1 = 0; y0 = 1; y1 = 0;         /* NaN corner: the reference's loop is empty too */
    } else {
        /* clamp in float first: keeps the int conversion in range for any input */
        x0 = (int)d_max(floorf(minx), 0.0f);
        y0 = (int)d_max(floorf(miny), 0.0f);
        x1 = (int)d_min(ceilf(maxx), (float)(width - 1));
        y1 = (int)d_min(ceilf(maxy), (float)(height - 1));
    }
    rec.meta = draw | (backface ? 1u << 16 : 0u);
    if (x0 > x1 || y0 > y1) { rec.bx = RZ_BBOX_EMPTY_X; rec.by = RZ_BBOX_EMPTY_X; }
    else { rec.bx = (uint32_t)x0 | ((uint32_t)x1 << 16); rec.by = (uint32_t)y0 | ((uint32_t)y1 << 16); }
    return true;
}

RZ_DEV void store_tri(TriRec *dst, const TriRec &r) {
    float4 *o = (float4*)dst;
    o[0] = make_float4(r.ax, r.ay, r.abx, r.aby);
    o[1] = make_float4(r.acx, r.acy, r.factor, r.z[0]);
    o[2] = make_float4(r.z[1], r.z[2], r.rw[0], r.rw[1]);
    o[3] = make_float4(r.rw[2], __uint_as_float(r.meta), __uint_as_float(r.bx), __uint_as_float(r.by));
}

__global__ void __launch_bounds__(256) rz_setup_kernel(PassParams P) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint32_t)P.num_tris) return;
    int d = find_draw(P.first_tri, P.num_draws, t);
    const DrawRec &D = P.draws[d];
    bool double_sided = (D.flags & RZ_DRAW_DOUBLE_SIDED) != 0;
    f4 c[3];
    for (int i = 0; i < 3; i++) {
        float4 q = ((const float4*)P.clip)[3u * t + i];
        c[i] = mk4(q.x, q.y, q.z, q.w);
    }
    TriRec rec;
    int x0, x1, y0, y1;
    uint2 empty = make_uint2(RZ_BBOX_EMPTY_X, RZ_BBOX_EMPTY_X);

    if (fully_visible(c[0]) && fully_visible(c[1]) && fully_visible(c[2])) {
        if (make_tri(c, P.width, P.height, double_sided, (uint32_t)d, rec, x0, x1, y0, y1)) {
            store_tri(&P.tri[t], rec);
            P.bbox[t] = make_uint2(rec.bx, rec.by);
            if (P.counters) atomicAdd(&P.counters[1], 1ull);
        } else {
            P.bbox[t] = empty;
        }
        return;
    }

    /* Sutherland-Hodgman against W, +X, -X, +Y, -Y, +Z, -Z (graphics.c:304-310),
     * positions and every varying float lerped with the same ratio */
    const int nv = D.num_varyings;
    f4 pos[2][RZ_MAX_POLY];
    float var[2][RZ_MAX_POLY][21];
    for (int i = 0; i < 3; i++) {
        pos[0][i] = c[i];
        const float *src = P.vary + (size_t)(3u * t + i) * RZ_VARY_STRIDE;
        for (int k = 0; k < nv; k++) var[0][i][k] = src[k];
    }
    int n = 3, cur = 0;
    for (int plane = PLANE_W; plane <= PLANE_NZ && n >= 3; plane++) {
        int dst = cur ^ 1, out = 0;
        for (int i = 0; i < n; i++) {
            int j = (i - 1 + n) % n;
            f4 a = pos[cur][j], b = pos[cur][i];
            bool a_in = plane_inside(a, plane), b_in = plane_inside(b, plane);
            if (a_in != b_in && out < RZ_MAX_POLY) {
                float r = plane_crossing(a, b, plane);
                pos[dst][out] = mk4(d_mix(a.x, b.x, r), d_mix(a.y, b.y, r), d_mix(a.z, b.z, r), d_mix(a.w, b.w, r));
                for (int k = 0; k < nv; k++) var[dst][out][k] = d_mix(var[cur][j][k], var[cur][i][k], r);
                out++;
            }
            if (b_in && out < RZ_MAX_POLY) {
                pos[dst][out] = b;
                for (int k = 0; k < nv; k++) var[dst][out][k] = var[cur][i][k];
                out++;
            }
        }
        n = out;
        cur = dst;
    }
    if (n < 3) { P.bbox[t] = empty; return; }

    uint32_t block = atomicAdd(&P.control[0], 1u);
    if (block >= (uint32_t)P.fan_capacity) {
        atomicOr(&P.control[2], (uint32_t)RZ_ERR_FAN_OVERFLOW);
        P.bbox[t] = empty;
        return;
    }
    for (int i = 0; i < n; i++) {
        float *dstv = P.fan_vary + ((size_t)block * RZ_MAX_POLY + i) * RZ_VARY_STRIDE;
        for (int k = 0; k < nv; k++) dstv[k] = var[cur][i][k];
    }
    int fan = 0;
    int ux0 = 65535, ux1 = -1, uy0 = 65535, uy1 = -1;
    for (int k = 0; k + 2 < n; k++) {
        f4 fc[3] = {pos[cur][0], pos[cur][k + 1], pos[cur][k + 2]};
        /* a culled fan triangle ends the whole fan (graphics.c:604-606) */
        if (!make_tri(fc, P.width, P.height, double_sided, (uint32_t)d, rec, x0, x1, y0, y1)) break;
        store_tri(&P.fan_tri[(size_t)block * RZ_MAX_FAN + k], rec);
        if (x0 <= x1 && y0 <= y1) {
            ux0 = x0 < ux0 ? x0 : ux0; ux1 = x1 > ux1 ? x1 : ux1;
            uy0 = y0 < uy0 ? y0 : uy0; uy1 = y1 > uy1 ? y1 : uy1;
        }
        fan++;
    }
    if (P.counters) atomicAdd(&P.counters[1], (unsigned long long)fan);
    if (fan == 0 || ux0 > ux1) { P.bbox[t] = empty; return; }
    TriRec head;
    head.ax = __uint_as_float(block);
    head.ay = head.abx = head.aby = head.acx = head.acy = head.factor = 0;
    head.z[0] = head.z[1] = head.z[2] = head.rw[0] = head.rw[1] = head.rw[2] = 0;
    head.meta = (uint32_t)d | (1u << 17) | ((uint32_t)fan << 18);
    head.bx = (uint32_t)ux0 | ((uint32_t)ux1 << 16);
    head.by = (uint32_t)uy0 | ((uint32_t)uy1 << 16);
    store_tri(&P.tri[t], head);
    P.bbox[t] = make_uint2(head.bx, head.by);
}

/* ========================================================================
 * K3  binning: for each 64x64-pixel bin the triangles whose bounding box
 *     touches it, in ascending triangle index (= submission order).
 *     Integer only.  One workgroup per bin streams the packed boxes twice:
 *     count, reserve a contiguous range with one atomic, then fill in order.
 * ======================================================================== */

RZ_DEV bool box_overlaps(uint2 bb, int rx0, int ry0, int rx1, int ry1) {
    int x0 = (int)(bb.x & 0xffffu), x1 = (int)(bb.x >> 16);
    int y0 = (int)(bb.y & 0xffffu), y1 = (int)(bb.y >> 16);
    return x0 <= rx1 && x1 >= rx0 && y0 <= ry1 && y1 >= ry0;
}

__global__ void __launch_bounds__(256) rz_bin_kernel(PassParams P) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const int bin = blockIdx.x;
    const int rx0 = (bin % P.bins_x) * RZ_BIN, ry0 = (bin / P.bins_x) * RZ_BIN;
    const int rx1 = rx0 + RZ_BIN - 1, ry1 = ry0 + RZ_BIN - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t T = (uint32_t)P.num_tris;

    uint32_t mine = 0;
    for (uint32_t t = threadIdx.x; t < T; t += 256u) mine += box_overlaps(P.bbox[t], rx0, ry0, rx1, ry1) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        uint32_t start = total ? atomicAdd(&P.control[1], total) : 0u;
        if (start + total > P.bin_capacity) {
            atomicOr(&P.control[2], (uint32_t)RZ_ERR_BIN_OVERFLOW);
            total = 0;
        }
        P.bin_start[bin] = start;
        P.bin_count[bin] = total;
        s_base = total ? start : 0xffffffffu;
    }
    __syncthreads();
    uint32_t cursor = s_base;
    if (cursor == 0xffffffffu) return;
    __syncthreads();

    for (uint32_t base = 0; base < T; base += 256u) {
        uint32_t t = base + threadIdx.x;
        bool hit = t < T && box_overlaps(P.bbox[t], rx0, ry0, rx1, ry1);
        unsigned long long m = __ballot(hit);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        uint32_t chunk = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (hit) {
            uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            P.bin_list[cursor + before + rank] = t;
        }
        cursor += chunk;
        __syncthreads();
    }
}

/* ========================================================================
 * K4  rasterise + shade.  One wave owns an 8x8 pixel block for the whole
 *     pass: its depth and colour live in two registers per lane, triangles
 *     are taken in ascending index from the bin list, so per-pixel order is
 *     the submission order by construction (graphics.c:512-532, :437-470).
 * ======================================================================== */

RZ_DEV TriRec load_tri(const TriRec *src) {
    const float4 *q = (const float4*)src;
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    TriRec r;
    r.ax = a.x; r.ay = a.y; r.abx = a.z; r.aby = a.w;
    r.acx = b.x; r.acy = b.y; r.factor = b.z; r.z[0] = b.w;
    r.z[1] = c.x; r.z[2] = c.y; r.rw[0] = c.z; r.rw[1] = c.w;
    r.rw[2] = d.x; r.meta = __float_as_uint(d.y); r.bx = __float_as_uint(d.z); r.by = __float_as_uint(d.w);
    return r;
}

RZ_DEV uint32_t pack_rgb(f4 c, uint32_t keep) {                 /* float_to_uchar, maths.c:33 */
    uint32_t r = (uint32_t)(unsigned char)(int)(c.x * 255);
    uint32_t g = (uint32_t)(unsigned char)(int)(c.y * 255);
    uint32_t b = (uint32_t)(unsigned char)(int)(c.z * 255);
    return r | (g << 8) | (b << 16) | (keep & 0xff000000u);
}

template <int NV>
RZ_DEV void interpolate(const float *v0, const float *v1, const float *v2,
                        float q0, float q1, float q2, float norm, float *out) {
    /* (s0*q0 + s1*q1 + s2*q2) * norm per float (graphics.c:431-434) */
    for (int k = 0; k < NV; k++) out[k] = (v0[k] * q0 + v1[k] * q1 + v2[k] * q2) * norm;
}

/* one triangle against this wave's 8x8 block */
RZ_DEV void raster_one(const PassParams &P, const TriRec &r, const float *v0, const float *v1, const float *v2,
                       int px, int py, bool valid, float &zcur, uint32_t &ccur, unsigned long long *shaded) {
    int x0 = (int)(r.bx & 0xffffu), x1 = (int)(r.bx >> 16);
    int y0 = (int)(r.by & 0xffffu), y1 = (int)(r.by >> 16);
    /* barycentrics at the pixel centre (graphics.c:385-397, 514-519) */
    float apx = ((float)px + 0.5f) - r.ax, apy = ((float)py + 0.5f) - r.ay;
    float s = (r.acy * apx - r.acx * apy) * r.factor;
    float t = (r.abx * apy - r.aby * apx) * r.factor;
    float w0 = 1 - s - t, w1 = s, w2 = t;
    bool inside = valid && px >= x0 && px <= x1 && py >= y0 && py <= y1 &&
                  w0 > -RZ_EPS && w1 > -RZ_EPS && w2 > -RZ_EPS;
    float depth = r.z[0] * w0 + r.z[1] * w1 + r.z[2] * w2;      /* :403-408 */
    bool pass = inside && depth <= zcur;                        /* :523 */
    if (!__any(pass)) return;

    const DrawRec &D = P.draws[RZ_META_DRAW(r.meta)];
    bool backface = RZ_META_BACKFACE(r.meta) != 0;
    if (pass) {
        float q0 = r.rw[0] * w0, q1 = r.rw[1] * w1, q2 = r.rw[2] * w2;      /* :426-429 */
        float norm = 1 / (q0 + q1 + q2);
        bool discard = false;
        f4 color;
        if (D.family == 2) {
            float v[3];
            interpolate<3>(v0, v1, v2, q0, q1, q2, norm, v);
            color = fs_skybox(D, v);
        } else if (D.family == 0) {
            float v[11];
            interpolate<11>(v0, v1, v2, q0, q1, q2, norm, v);
            color = fs_blinn(D, v, backface, &discard);
        } else {
            float v[21];
            interpolate<21>(v0, v1, v2, q0, q1, q2, norm, v);
            color = fs_pbr(D, v, backface, &discard);
        }
        if (!discard) {                                         /* draw_fragment, :449-469 */
            color = mk4(d_sat(color.x), d_sat(color.y), d_sat(color.z), d_sat(color.w));
            if (D.flags & RZ_DRAW_BLEND) {
                float dr = (float)(ccur & 0xffu) / 255.0f;
                float dg = (float)((ccur >> 8) & 0xffu) / 255.0f;
                float db = (float)((ccur >> 16) & 0xffu) / 255.0f;
                color.x = color.x * color.w + dr * (1 - color.w);
                color.y = color.y * color.w + dg * (1 - color.w);
                color.z = color.z * color.w + db * (1 - color.w);
            }
            ccur = pack_rgb(color, ccur);
            zcur = depth;
        }
    }
    if (shaded) {
        unsigned long long m = __ballot(pass);
        if ((threadIdx.x & 63) == 0) atomicAdd(shaded, (unsigned long long)__popcll(m));
    }
}

__global__ void __launch_bounds__(256) rz_raster_kernel(PassParams P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (P.width + RZ_TILE - 1) / RZ_TILE;
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int bx0 = tile_x * RZ_TILE + (wave & 1) * 8, by0 = tile_y * RZ_TILE + (wave >> 1) * 8;
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool valid = px < P.width && py < P.height;
    if (bx0 >= P.width || by0 >= P.height) return;             /* whole wave off-screen */
    const size_t pixel = (size_t)py * P.width + px;
    const int bin = (by0 / RZ_BIN) * P.bins_x + bx0 / RZ_BIN;
    const uint32_t count = P.bin_count[bin];
    if (count == 0 && !P.clear_color_pending && !P.clear_depth_pending) return;

    float zcur = 1.0f;
    uint32_t ccur = 0;
    if (valid) {
        zcur = P.clear_depth_pending ? P.clear_depth : P.depth[pixel];
        ccur = P.clear_color_pending ? P.clear_color : P.color[pixel];
    }
    const uint32_t *list = P.bin_list + P.bin_start[bin];
    const int bx1 = bx0 + 7, by1 = by0 + 7;
    unsigned long long *shaded = P.counters ? P.counters + 2 : nullptr;

    for (uint32_t base = 0; base < count; base += 64u) {
        uint32_t mine = 0xffffffffu;
        bool hit = false;
        if (base + lane < count) {
            mine = list[base + lane];
            hit = box_overlaps(P.bbox[mine], bx0, by0, bx1, by1);
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)mine, j);
            TriRec r = load_tri(&P.tri[t]);
            if (!RZ_META_CLIPPED(r.meta)) {
                const float *v = P.vary + (size_t)(3u * t) * RZ_VARY_STRIDE;
                raster_one(P, r, v, v + RZ_VARY_STRIDE, v + 2 * RZ_VARY_STRIDE, px, py, valid, zcur, ccur, shaded);
            } else {
                uint32_t block = __float_as_uint(r.ax);
                int fan = (int)RZ_META_FAN(r.meta);
                const float *v = P.fan_vary + (size_t)block * RZ_MAX_POLY * RZ_VARY_STRIDE;
                for (int k = 0; k < fan; k++) {
                    TriRec f = load_tri(&P.fan_tri[(size_t)block * RZ_MAX_FAN + k]);
                    raster_one(P, f, v, v + (k + 1) * RZ_VARY_STRIDE, v + (k + 2) * RZ_VARY_STRIDE,
                               px, py, valid, zcur, ccur, shaded);
                }
            }
        }
    }
    if (valid) {
        P.depth[pixel] = zcur;
        P.color[pixel] = ccur;
    }
}

/* ========================================================================
 * K6  framebuffer / texture utilities
 * ======================================================================== */

/* framebuffer_clear_color / _depth (graphics.c:38-55) when no draw follows */
__global__ void rz_clear_kernel(uint32_t *color, float *depth, size_t n, int do_color, uint32_t c,
                                int do_depth, float d) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (do_color) color[i] = c;
    if (do_depth) depth[i] = d;
}

/* texture_from_depthbuffer (texture.c:147-158): the texel is (d, d, d, 1);
 * stored as one float, expanded on fetch */
__global__ void rz_depth_to_texture_kernel(const float *depth, float *texels, int w, int h, int flags, int mbits) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, mbits, x, y)] = depth[(size_t)y * w + x];
}

/* texture_from_colorbuffer (texture.c:130-145): byte / 255.0f per channel */
__global__ void rz_color_to_texture_kernel(const uint32_t *color, float4 *texels, int w, int h, int flags, int mbits) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    uint32_t c = color[(size_t)y * w + x];
    texels[texel_index(w, flags, mbits, x, y)] =
        make_float4((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f,
                    (float)((c >> 16) & 0xffu) / 255.0f, (float)(c >> 24) / 255.0f);
}

/* row-major float4 image -> device layout (Morton where the size allows) */
__global__ void rz_texture_tile_kernel(const float4 *linear, float4 *texels, int w, int h, int flags, int mbits) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, mbits, x, y)] = linear[(size_t)y * w + x];
}

/* device layout -> row-major float4 (texture_sync) */
__global__ void rz_texture_untile_kernel(const void *texels, float4 *linear, int w, int h, int flags, int mbits) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    uint32_t i = texel_index(w, flags, mbits, x, y);
    float4 v;
    if (flags & RZ_TEX_R32F) { float d = ((const float*)texels)[i]; v = make_float4(d, d, d, 1.0f); }
    else v = ((const float4*)texels)[i];
    linear[(size_t)y * w + x] = v;
}

#endif
