/*
 * The gfx950 kernels of the triangle pipeline.  A launch covers a batch of
 * passes (a pass = draws into one framebuffer; grid y or z = pass):
 *
 *   rz_vertex_kernel       vertex shaders incl. skeletal skinning, 1 thread / vertex; cull -> viewport -> bbox of
 *                          triangles that need no clipping
 *   rz_clip_kernel         homogeneous clipping -> fan for the rest, 1 wave / triangle
 *   rz_bin_count / _scan / _fill_kernel   ordered lists per 64x64-pixel bin
 *   rz_strip_kernel        passes without blending: coverage + depth, leaves depth and one
 *                          fragment reference per pixel; 1 wave / strip of 16x8-pixel blocks
 *   rz_sky_kernel, rz_shade_list_kernel   per-family fragment shading of those references
 *   rz_raster_list_kernel  passes with blending: coverage, depth and ordered in-kernel shading; 1 wave / 8x8 pixels,
 *                          over a list of tiles
 *   rz_blend_visibility_kernel   the same passes in large framebuffers: visibility first, ordered shading only
 *                          for the tiles where blended fragments pile up
 *   rz_clear_kernel, rz_*_to_texture(s)_kernel, rz_texture_*tile*_kernel   framebuffer / texture utilities
 *
 * All arithmetic that decides coverage, depth or colour follows the reference
 * expression order (renderer/core/graphics.c, cited inline).
 */
#ifndef RZ_KERNELS_H
#define RZ_KERNELS_H

#include <type_traits>
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

/* first kernel of a batch: what the passes' kernels count in or add into starts at zero */
__global__ void __launch_bounds__(256) rz_begin_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    const uint32_t gv = blockIdx.x * 256u + threadIdx.x;
    /* the pass's device-written scalars start at zero -- unless another pass of the flush ran out of
     * scratch and this one may depend on it: then it is marked to be left out and run again later */
    if (gv < 8u) P.control[gv] = (gv == 2u && P.level > 0 && *P.flush_flag != 0u) ? (uint32_t)RZ_ERR_SKIPPED : 0u;
    if (blockIdx.y == 0 && gv < RZ_LISTS * RZ_QUEUE_WAYS) P.queue_counts[gv * RZ_QUEUE_PAD] = 0;    /* ... and the batch's work-list lengths */
    for (uint32_t t = gv; t < (uint32_t)(P.tiles_x * P.tiles_y); t += gridDim.x * 256u) P.tile_listed[t] = 0;
    if (P.sky_draws && gv < (uint32_t)RZ_SKY_TABLE) P.sky_table[gv] = 0xffffffffu;      /* RZ_REF_NONE: no sky triangle yet */
    {   /* ... and the pass's (chunk, bin) matrix, which rz_bin_count_kernel adds into */
        const uint32_t cells = (((uint32_t)P.num_tris + 63u) / 64u) * (uint32_t)(P.bins_x * P.bins_y);
        for (uint32_t t = gv; t < cells; t += gridDim.x * 256u) P.bin_counts[t] = 0;
    }
}

RZ_DEV bool make_tri(const f4 c[3], int width, int height, bool double_sided, uint32_t draw,
                     TriRec &rec, int &x0, int &x1, int &y0, int &y1);
RZ_DEV bool fully_visible(f4 v);
RZ_DEV uint2 pack_bin_box(uint32_t bx, uint32_t by, uint32_t entries);
RZ_DEV void store_tri(TriRec *dst, const TriRec &r);

/* what the rasteriser needs to know about a triangle's draw without loading it (or-ed into the record's meta word) */
RZ_DEV uint32_t draw_class_bits(const DrawRec &D) {
    const bool blend = (D.flags & RZ_DRAW_BLEND) != 0;
    return (blend ? 1u << 22 : 0u) | (D.alpha_cutoff > 0 ? 1u << 23 : 0u) | ((uint32_t)D.family << 24) |
           ((D.flags & RZ_DRAW_SHADOW_PASS) && !blend ? 1u << 26 : 0u);
}

/* the groups of four varyings anything reads of a draw's triangles, as a bit mask.  A shadow pass's fragments are
 * black whatever their varyings (blinn_shader.c:101-115, pbr_shader.c:113-136): only the alpha cutoff's texture
 * coordinates are ever looked at. */
RZ_DEV uint32_t draw_vary_groups(const DrawRec &D) {
    if (D.flags & RZ_DRAW_SHADOW_PASS) {
        if (D.flags & RZ_DRAW_BLEND) return (1u << ((D.num_varyings + 3) / 4)) - 1u;
        if (!(D.alpha_cutoff > 0)) return 0u;
        return 1u << ((D.family == 0 ? RZ_BV_TEXCOORD : RZ_PV_TEXCOORD) / 4);
    }
    return (1u << ((D.num_varyings + 3) / 4)) - 1u;                 /* 1 skybox, 3 blinn, 6 pbr */
}

/* Vertex shaders and triangle set-up in one kernel.  A workgroup takes 85 triangles: thread 3k + i shades
 * corner i of triangle k (thread 255 idles) and leaves its clip-space position in LDS; thread k then does
 * triangle k's set-up from the three positions -- trivial accept (graphics.c:275-277), cull, record and pixel
 * box, or the queue of the clipper -- and tells the corners' threads what became of it.
 * Varyings are stored for triangles that live (half of a closed mesh's are culled: nothing ever reads
 * theirs), clip-space positions only for triangles the clipper will read them of.
 *
 * A wave's corners nearly always belong to one draw.  The wave takes the draws its lanes belong to one after the
 * other, so that a draw's record -- four matrices, flags, pointers -- comes through scalar loads once per wave; read
 * per lane those 60 words were three times the vertex itself and the kernel waited on the vector memory path. */
#define RZ_VS_TRIS 85
__global__ void __launch_bounds__(256) rz_vertex_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    __shared__ float4 s_clip[256];
    __shared__ int s_fate[RZ_VS_TRIS + 1];          /* 0 culled, 1 alive, 2 for the clipper */
    __shared__ int s_draw[RZ_VS_TRIS + 1];
    const uint32_t first = blockIdx.x * (uint32_t)RZ_VS_TRIS;
    if (first >= (uint32_t)P.num_tris) return;                                  /* uniform over the workgroup */
    const uint32_t tri = first + threadIdx.x / 3u, corner = threadIdx.x % 3u;
    const bool mine = threadIdx.x < 3u * RZ_VS_TRIS && tri < (uint32_t)P.num_tris;
    const uint32_t gv = 3u * tri + corner;
    /* the draw of the workgroup's first triangle by a search every lane makes alike; a lane past that draw's end
     * searches for itself */
    int d = find_draw(P.first_tri, P.num_draws, first);
    if (mine && tri >= P.first_tri[d + 1]) d = find_draw(P.first_tri, P.num_draws, tri);
    f4 clip = mk4(0, 0, 0, 0);
    float vary[RZ_VARY_STRIDE];     /* put together in registers, stored as whole groups of four below */
    for (int k = 0; k < RZ_VARY_STRIDE; k++) vary[k] = 0;
    uint32_t groups = 0;
    unsigned long long rest = __ballot(mine);
    while (rest) {
    const int d_now = __builtin_amdgcn_readlane(d, __ffsll((long long)rest) - 1);
    const bool now = mine && d == d_now;
    rest &= ~__ballot(now);
    const DrawRec &D = P.draws[d_now];
    if (now) {
    groups = draw_vary_groups(D);
    uint32_t local = gv - 3u * P.first_tri[d_now];
    const float4 RZ_GLOBAL *src = (const float4 RZ_GLOBAL*)(D.vertices.as<const char>() + (size_t)local * 80u);
    float a[20];                    /* position 0-2, texcoord 3-4, normal 5-7, tangent 8-11, joint 12-15, weight 16-19 */
    for (int i = 0; i < 5; i++) {
        float4 q = src[i];
        a[4 * i] = q.x; a[4 * i + 1] = q.y; a[4 * i + 2] = q.z; a[4 * i + 3] = q.w;
    }
    f4 pos = mk4(a[0], a[1], a[2], 1.0f);

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
            int tc = D.family == 0 ? RZ_BV_TEXCOORD : RZ_PV_TEXCOORD;
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
            vary[0] = world.x; vary[1] = world.y; vary[2] = world.z;           /* RZ_PV_WORLD == 0 */
            if (D.family == 0) {                                /* blinn common_vertex_shader */
                vary[3] = light_space.x; vary[4] = light_space.y; vary[5] = light_space.z;
                n = unit3(n);
                vary[6] = a[3]; vary[7] = a[4];
                vary[8] = n.x; vary[9] = n.y; vary[10] = n.z;
            } else {                                            /* pbr common_vertex_shader */
                vary[RZ_PV_CLIP] = clip.x; vary[RZ_PV_CLIP + 1] = clip.y; vary[RZ_PV_CLIP + 2] = clip.z; vary[RZ_PV_CLIP + 3] = clip.w;
                vary[RZ_PV_DEPTH] = light_space.x; vary[RZ_PV_DEPTH + 1] = light_space.y; vary[RZ_PV_DEPTH + 2] = light_space.z;
                n = unit3(n);
                vary[RZ_PV_NORMAL] = n.x; vary[RZ_PV_NORMAL + 1] = n.y; vary[RZ_PV_NORMAL + 2] = n.z;
                if (D.maps[3].data) {
                    float tm[9] = {model[0], model[1], model[2], model[4], model[5], model[6],
                                   model[8], model[9], model[10]};
                    f3 t = unit3(m3v3(tm, mk3(a[8], a[9], a[10])));
                    f3 b = scale3(cross3(n, t), a[11]);
                    vary[RZ_PV_TANGENT] = t.x; vary[RZ_PV_TANGENT + 1] = t.y; vary[RZ_PV_TANGENT + 2] = t.z;
                    vary[RZ_PV_BITANGENT] = b.x; vary[RZ_PV_BITANGENT + 1] = b.y; vary[RZ_PV_BITANGENT + 2] = b.z;
                } else {
                    /* the reference leaves tangent/bitangent untouched here;
                     * nothing reads them without a normal map */
                    for (int k = 0; k < 3; k++) { vary[RZ_PV_TANGENT + k] = 0; vary[RZ_PV_BITANGENT + k] = 0; }
                }
                vary[RZ_PV_TEXCOORD] = a[3]; vary[RZ_PV_TEXCOORD + 1] = a[4];
            }
        }
    }
    }
    }
    s_clip[threadIdx.x] = make_float4(clip.x, clip.y, clip.z, clip.w);
    if (mine && corner == 0u) s_draw[threadIdx.x / 3u] = d;
    __syncthreads();
    /* the set-up, triangle k in thread k: the workgroup's first two waves, two lanes in three at work (in the corners'
     * own threads it kept one lane in three of all four waves busy) */
    if (threadIdx.x < (uint32_t)RZ_VS_TRIS && first + threadIdx.x < (uint32_t)P.num_tris) {
        const uint32_t t = first + threadIdx.x;
        const int dt = s_draw[threadIdx.x];
        const DrawRec &D = P.draws[dt];
        const bool double_sided = (D.flags & RZ_DRAW_DOUBLE_SIDED) != 0;
        const uint32_t class_bits = draw_class_bits(D);
        f4 c[3];
        for (int i = 0; i < 3; i++) {
            const float4 q = s_clip[3u * threadIdx.x + i];
            c[i] = mk4(q.x, q.y, q.z, q.w);
        }
        const uint2 empty = make_uint2(RZ_BBOX_EMPTY_X, RZ_BBOX_EMPTY_X);
        int fate = 0;
        if (fully_visible(c[0]) && fully_visible(c[1]) && fully_visible(c[2])) {        /* graphics.c:290-300 */
            TriRec rec;
            int x0, x1, y0, y1;
            if (make_tri(c, P.width, P.height, double_sided, (uint32_t)dt, rec, x0, x1, y0, y1)) {
                rec.meta |= class_bits;
                store_tri(&P.tri[t], rec);
                if (D.sky_slot) {       /* a sky box on the pass's table: its place there instead of one in the bins */
                    P.bbox[t] = empty;
                    if (x0 <= x1 && y0 <= y1)
                        P.sky_table[((uint32_t)(D.sky_slot - 1) * RZ_SKY_FACES + (t - P.first_tri[dt])) * RZ_MAX_FAN] = t;
                } else {
                    P.bbox[t] = pack_bin_box(rec.bx, rec.by, 1u);
                }
                if (P.counters) atomicAdd(&P.counters[1], 1ull);
                fate = 1;
            } else {
                P.bbox[t] = empty;
            }
        } else {
            /* needs the clipper: queued for rz_clip_kernel, which fills tri/bbox */
            P.bbox[t] = empty;
            P.clip_list[atomicAdd(&P.control[3], 1u)] = t;
            fate = 2;
        }
        s_fate[threadIdx.x] = fate;
    }
    __syncthreads();
    if (!mine) return;
    const int fate = s_fate[threadIdx.x / 3u];
    if (fate == 0) return;
    if (fate == 2) {
        P.clip.as<float4>()[gv] = make_float4(clip.x, clip.y, clip.z, clip.w);
        groups = (1u << ((P.draws[d].num_varyings + 3) / 4)) - 1u;          /* the clipper carries every varying along */
    }
    float4 RZ_GLOBAL *out = (float4 RZ_GLOBAL*)(P.vary + (size_t)gv * RZ_VARY_STRIDE);
    for (int i = 0; i < RZ_VARY_STRIDE / 4; i++)
        if ((groups >> i) & 1u) out[i] = make_float4(vary[4 * i], vary[4 * i + 1], vary[4 * i + 2], vary[4 * i + 3]);
}

/* ========================================================================
 * K0  skeletal poses (renderer/core/skeleton.c:231-341), one wave per pose
 * ======================================================================== */

/* a skeleton on the device, one block of 32-bit words:
 *   [0] joints J, [1..3] unused, parent[J], inverse_bind[J][16], tracks[J][3] (PoseTrack), key data;
 * a track's times and values sit at word offsets from the block's start */
struct PoseTrack { int32_t count; int32_t ascending; uint32_t times_at; uint32_t values_at; };
struct PoseJob { const uint32_t *skeleton; float time; int32_t pad; float *joints; float *normals; };
struct AttachJob { DrawRec *draw; const float *joint; };
#define RZ_POSE_MAX_JOINTS 256

/* skeleton.c:231-313: the first key, the last key, or the first segment [t_i, t_i+1) that holds the
 * time with its parameter (t - t_i) / (t_i+1 - t_i); -1 / -2 = clamped to an end.  Key times that
 * ascend strictly (checked when the skeleton is uploaded) are searched by bisection, which finds
 * that same segment. */
RZ_DEV int pose_segment(const PoseTrack &t, const float *times, float time, float *param) {
    if (time <= times[0]) return -1;
    if (time >= times[t.count - 1]) return -2;
    if (t.ascending) {
        int lo = 0, hi = t.count - 1;                       /* times[lo] <= time < times[hi] */
        while (hi - lo > 1) {
            const int mid = (lo + hi) / 2;
            if (time >= times[mid]) lo = mid; else hi = mid;
        }
        *param = (time - times[lo]) / (times[lo + 1] - times[lo]);
        return lo;
    }
    for (int i = 0; i < t.count - 1; i++) {
        const float a = times[i], b = times[i + 1];
        if (time >= a && time < b) { *param = (time - a) / (b - a); return i; }
    }
    return -2;                                              /* not reached for a well-formed track */
}

RZ_DEV f3 pose_vec3(const PoseTrack &t, const uint32_t *block, float time, f3 fallback) {
    if (t.count == 0) return fallback;
    const float *times = (const float*)block + t.times_at, *v = (const float*)block + t.values_at;
    float u = 0;
    const int s = pose_segment(t, times, time, &u);
    if (s == -1) return mk3(v[0], v[1], v[2]);
    if (s == -2) { v += (size_t)(t.count - 1) * 3; return mk3(v[0], v[1], v[2]); }
    v += (size_t)s * 3;
    return mk3(d_mix(v[0], v[3], u), d_mix(v[1], v[4], u), d_mix(v[2], v[5], u));     /* vec3_lerp */
}

/* quat_slerp (maths.c:291-318): acos and sin in double, narrowed, as the C library calls there are */
RZ_DEV f4 pose_quat(const PoseTrack &t, const uint32_t *block, float time) {
    if (t.count == 0) return mk4(0, 0, 0, 1);
    const float *times = (const float*)block + t.times_at, *v = (const float*)block + t.values_at;
    float u = 0;
    const int s = pose_segment(t, times, time, &u);
    if (s == -1) return mk4(v[0], v[1], v[2], v[3]);
    if (s == -2) { v += (size_t)(t.count - 1) * 4; return mk4(v[0], v[1], v[2], v[3]); }
    v += (size_t)s * 4;
    const f4 a = mk4(v[0], v[1], v[2], v[3]);
    f4 b = mk4(v[4], v[5], v[6], v[7]);
    float cosv = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (cosv < 0) { b = mk4(-b.x, -b.y, -b.z, -b.w); cosv = -cosv; }
    if (cosv > 1 - RZ_EPS) return mk4(d_mix(a.x, b.x, u), d_mix(a.y, b.y, u), d_mix(a.z, b.z, u), d_mix(a.w, b.w, u));
    const float angle = (float)acos((double)cosv);
    const float sinv = (float)sin((double)angle);
    const float fa = (float)sin((double)((1 - u) * angle)) / sinv;
    const float fb = (float)sin((double)(u * angle)) / sinv;
    return mk4(fa * a.x + fb * b.x, fa * a.y + fb * b.y, fa * a.z + fb * b.z, fa * a.w + fb * b.w);
}

/* mat3_inverse_transpose (maths.c:432-467): adjugate over determinant, cofactor signs as there */
RZ_DEV void inverse_transpose3(const float *m, float *out) {
    float adj[9];
    adj[0] = +(m[4] * m[8] - m[7] * m[5]);
    adj[1] = -(m[3] * m[8] - m[6] * m[5]);
    adj[2] = +(m[3] * m[7] - m[6] * m[4]);
    adj[3] = -(m[1] * m[8] - m[7] * m[2]);
    adj[4] = +(m[0] * m[8] - m[6] * m[2]);
    adj[5] = -(m[0] * m[7] - m[6] * m[1]);
    adj[6] = +(m[1] * m[5] - m[4] * m[2]);
    adj[7] = -(m[0] * m[5] - m[3] * m[2]);
    adj[8] = +(m[0] * m[4] - m[3] * m[1]);
    const float a = +m[0] * (m[4] * m[8] - m[5] * m[7]);
    const float b = -m[1] * (m[3] * m[8] - m[5] * m[6]);
    const float c = +m[2] * (m[3] * m[7] - m[4] * m[6]);
    const float det = a + b + c;
    const float inv = 1 / det;
    for (int k = 0; k < 9; k++) out[k] = adj[k] * inv;
}

__global__ void __launch_bounds__(64) rz_pose_kernel(const PoseJob *__restrict__ jobs) {
    const PoseJob job = jobs[blockIdx.x];
    __shared__ float s_world[RZ_POSE_MAX_JOINTS][16];
    const uint32_t *block = job.skeleton;
    const int J = (int)block[0];
    const int *parent = (const int*)(block + 4);
    const float *bind = (const float*)(block + 4 + J);
    const PoseTrack *tracks = (const PoseTrack*)(block + 4 + J + 16 * J);
    const int lane = threadIdx.x;
    /* every joint's own transform, mat4_from_trs (maths.c:519): T * (R * S) as full products */
    for (int j = lane; j < J; j += 64) {
        const f3 t = pose_vec3(tracks[3 * j], block, job.time, mk3(0, 0, 0));
        const f4 q = pose_quat(tracks[3 * j + 1], block, job.time);
        const f3 s = pose_vec3(tracks[3 * j + 2], block, job.time, mk3(1, 1, 1));
        const float xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
        const float yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w, zz = q.z * q.z, zw = q.z * q.w;
        const float rm[16] = {1 - 2 * (yy + zz), 2 * (xy - zw), 2 * (xz + yw), 0,
                              2 * (xy + zw), 1 - 2 * (xx + zz), 2 * (yz - xw), 0,
                              2 * (xz - yw), 2 * (yz + xw), 1 - 2 * (xx + yy), 0,
                              0, 0, 0, 1};                                          /* maths.c:492 */
        const float sm[16] = {s.x, 0, 0, 0, 0, s.y, 0, 0, 0, 0, s.z, 0, 0, 0, 0, 1};
        const float tm[16] = {1, 0, 0, t.x, 0, 1, 0, t.y, 0, 0, 1, t.z, 0, 0, 0, 1};
        float rs[16], local[16];
        mat_mul<4>(rm, sm, rs);
        mat_mul<4>(tm, rs, local);
        for (int k = 0; k < 16; k++) s_world[j][k] = local[k];
    }
    __syncthreads();
    /* joint-to-model: parent's * own, joints in index order (a parent comes before its children), sixteen
     * lanes one element each */
    for (int j = 0; j < J; j++) {
        const int p = parent[j];
        if (p < 0) continue;                                /* uniform */
        float acc = 0;
        if (lane < 16) {
            const int r = lane >> 2, c = lane & 3;
            for (int k = 0; k < 4; k++) acc += s_world[p][r * 4 + k] * s_world[j][k * 4 + c];
        }
        __syncthreads();
        if (lane < 16) s_world[j][lane] = acc;
        __syncthreads();
    }
    /* joint matrix = joint-to-model * inverse bind; normal matrix = its upper 3x3, inverse-transposed */
    for (int j = lane; j < J; j += 64) {
        float world[16], ib[16], jm[16], n3[9], nm[9];
        for (int k = 0; k < 16; k++) { world[k] = s_world[j][k]; ib[k] = bind[(size_t)j * 16 + k]; }
        mat_mul<4>(world, ib, jm);
        for (int k = 0; k < 16; k++) job.joints[(size_t)j * 16 + k] = jm[k];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) n3[r * 3 + c] = jm[r * 4 + c];
        inverse_transpose3(n3, nm);
        for (int k = 0; k < 9; k++) job.normals[(size_t)j * 9 + k] = nm[k];
    }
}

/* rigid node animation (blinn_shader.c:270-283, pbr_shader.c:512-525): the draw's model matrix becomes
 * model * joint[attached], its normal matrix the inverse transpose of that product's upper 3x3 */
__global__ void __launch_bounds__(64) rz_attach_kernel(const AttachJob *__restrict__ jobs, int count) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    DrawRec *D = jobs[i].draw;
    float model[16], joint[16], out[16], n3[9], nm[9];
    for (int k = 0; k < 16; k++) { model[k] = D->model[k]; joint[k] = jobs[i].joint[k]; }
    mat_mul<4>(model, joint, out);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) n3[r * 3 + c] = out[r * 4 + c];
    inverse_transpose3(n3, nm);
    for (int k = 0; k < 16; k++) D->model[k] = out[k];
    for (int k = 0; k < 9; k++) D->normal[k] = nm[k];
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
    rec.meta = draw | (backface ? 1u << 16 : 0u);       /* class bits are or-ed in by the caller */
    if (x0 > x1 || y0 > y1) { rec.bx = RZ_BBOX_EMPTY_X; rec.by = RZ_BBOX_EMPTY_X; }
    else { rec.bx = (uint32_t)x0 | ((uint32_t)x1 << 16); rec.by = (uint32_t)y0 | ((uint32_t)y1 << 16); }
    return true;
}

/* What the binning kernel reads per input triangle: x0 | x1 << 16 and
 * y0 | y1 << 14 | entries << 28, entries = list entries the triangle adds to a
 * bin it touches (1, or the length of its fan).  Framebuffers are at most
 * 16384 pixels high, so 14 bits per y bound are enough. */
RZ_DEV uint2 pack_bin_box(uint32_t bx, uint32_t by, uint32_t entries) {
    return make_uint2(bx, (by & 0x3fffu) | ((by >> 16) << 14) | (entries << 28));
}

RZ_DEV void store_tri(TriRec *dst, const TriRec &r) {
    float4 *o = (float4*)dst;
    o[0] = make_float4(r.ax, r.ay, r.abx, r.aby);
    o[1] = make_float4(r.acx, r.acy, r.factor, r.z[0]);
    o[2] = make_float4(r.z[1], r.z[2], r.rw[0], r.rw[1]);
    o[3] = make_float4(r.rw[2], __uint_as_float(r.meta), __uint_as_float(r.bx), __uint_as_float(r.by));
}

#define RZ_BCAST_F(v, j) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (j)))
#define RZ_BCAST_I(v, j) __builtin_amdgcn_readlane((int)(v), (j))

/* K2b  clipping, one WAVE per queued triangle.  Lane c carries one float
 * "channel" of every polygon corner through the seven Sutherland-Hodgman
 * passes: channels 0-3 are the clip-space position, 4.. the varyings.  Inside
 * tests and crossing ratios depend on positions only, so every lane computes
 * them identically from LDS broadcasts and then lerps its own channel with
 * the same ratio -- exactly what clip_against_plane does per float
 * (graphics.c:205-253).  Planes in the reference order W +X -X +Y -Y +Z -Z
 * (:304-310); the fan and its early end on a culled triangle as in :584-607. */
__global__ void __launch_bounds__(64) rz_clip_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    __shared__ float poly[2][RZ_MAX_POLY][32];
    __shared__ int s_box[4][RZ_MAX_FAN];
    const int lane = threadIdx.x;
    const uint32_t queued = P.control[3];
    for (uint32_t item = blockIdx.x; item < queued; item += gridDim.x) {
        const uint32_t t = P.clip_list[item];
        const int d = find_draw(P.first_tri, P.num_draws, t);
        const DrawRec &D = P.draws[d];
        const int nv = D.num_varyings;
        const bool double_sided = (D.flags & RZ_DRAW_DOUBLE_SIDED) != 0;
        const uint32_t class_bits = draw_class_bits(D);
        __syncthreads();
        if (lane < 32) {
            for (int i = 0; i < 3; i++) {
                float v = 0;
                if (lane < 4) v = P.clip[(size_t)(3u * t + i) * 4 + lane];
                else if (lane < 4 + nv) v = P.vary[(size_t)(3u * t + i) * RZ_VARY_STRIDE + (lane - 4)];
                poly[0][i][lane] = v;
            }
        }
        __syncthreads();
        int n = 3, cur = 0;
        for (int plane = PLANE_W; plane <= PLANE_NZ && n >= 3; plane++) {
            const int dst = cur ^ 1;
            /* every lane's own channel of all corners, and in lane e the edge that ends in corner e (positions of
             * corners e - 1 and e, which side of the plane they lie on, the crossing's ratio): one trip to LDS and one
             * division per plane, where walking the edges one after the other made a trip and a division per edge */
            float vals[RZ_MAX_POLY];
#pragma unroll
            for (int i = 0; i < RZ_MAX_POLY; i++) vals[i] = i < n ? poly[cur][i][lane & 31] : 0.0f;
            const int e = lane < n ? lane : 0, j = (e - 1 + n) % n;
            const float4 pa = *(const float4*)&poly[cur][j][0], pb = *(const float4*)&poly[cur][e][0];
            const f4 a = mk4(pa.x, pa.y, pa.z, pa.w), b = mk4(pb.x, pb.y, pb.z, pb.w);
            const bool a_in = plane_inside(a, plane), b_in = plane_inside(b, plane);
            float r = 0;
            if (a_in != b_in) r = plane_crossing(a, b, plane);
            const uint32_t cross = (uint32_t)__ballot(lane < n && a_in != b_in), inside = (uint32_t)__ballot(lane < n && b_in);
            float last = vals[0];
#pragma unroll
            for (int i = 1; i < RZ_MAX_POLY; i++) if (i == n - 1) last = vals[i];
            int out = 0;
#pragma unroll
            for (int i = 0; i < RZ_MAX_POLY; i++) {
                if (i >= n) break;                                  /* uniform, as every branch of this loop */
                if (((cross >> i) & 1u) && out < RZ_MAX_POLY) {
                    const float ratio = RZ_BCAST_F(r, i);
                    if (lane < 32) poly[dst][out][lane] = d_mix(i == 0 ? last : vals[i > 0 ? i - 1 : 0], vals[i], ratio);
                    out++;
                }
                if (((inside >> i) & 1u) && out < RZ_MAX_POLY) {
                    if (lane < 32) poly[dst][out][lane] = vals[i];
                    out++;
                }
            }
            n = out;
            cur = dst;
            __syncthreads();
        }
        if (n < 3) continue;                    /* bbox stays empty */

        uint32_t block = 0;
        if (lane == 0) block = atomicAdd(&P.control[0], 1u);
        block = (uint32_t)__builtin_amdgcn_readfirstlane((int)block);
        if (block >= (uint32_t)P.fan_capacity) {
            if (lane == 0) { atomicOr(&P.control[2], (uint32_t)RZ_ERR_FAN_OVERFLOW); atomicOr(P.flush_flag, 1u); }
            continue;                           /* control[0] goes on counting the blocks the pass needs */
        }
        if (lane >= 4 && lane < 4 + nv) {
            for (int i = 0; i < n; i++)
                P.fan_vary[((size_t)block * RZ_MAX_POLY + i) * RZ_VARY_STRIDE + (lane - 4)] = poly[cur][i][lane];
        }
        /* lane k builds fan triangle (0, k+1, k+2) */
        const int k = lane;
        TriRec rec;
        int x0 = 1, x1 = 0, y0 = 1, y1 = 0;
        bool alive = false;
        if (k + 2 < n) {
            f4 fc[3];
            fc[0] = mk4(poly[cur][0][0], poly[cur][0][1], poly[cur][0][2], poly[cur][0][3]);
            fc[1] = mk4(poly[cur][k + 1][0], poly[cur][k + 1][1], poly[cur][k + 1][2], poly[cur][k + 1][3]);
            fc[2] = mk4(poly[cur][k + 2][0], poly[cur][k + 2][1], poly[cur][k + 2][2], poly[cur][k + 2][3]);
            alive = make_tri(fc, P.width, P.height, double_sided, (uint32_t)d, rec, x0, x1, y0, y1);
        }
        unsigned long long culled = __ballot(k + 2 < n && !alive);
        const int fan = culled ? (__ffsll((long long)culled) - 1) : (n - 2);
        if (k < fan) {
            rec.meta |= class_bits;
            store_tri(&P.fan_tri[(size_t)block * RZ_MAX_FAN + k], rec);
            bool any_px = x0 <= x1 && y0 <= y1;
            if (D.sky_slot && any_px)   /* a sky box on the pass's table */
                P.sky_table[((uint32_t)(D.sky_slot - 1) * RZ_SKY_FACES + (t - P.first_tri[d])) * RZ_MAX_FAN + (uint32_t)k] =
                    0x80000000u | (block * RZ_MAX_FAN + (uint32_t)k);
            s_box[0][k] = any_px ? x0 : 65535; s_box[1][k] = any_px ? x1 : -1;
            s_box[2][k] = any_px ? y0 : 65535; s_box[3][k] = any_px ? y1 : -1;
        }
        __syncthreads();
        if (lane == 0) {
            int ux0 = 65535, ux1 = -1, uy0 = 65535, uy1 = -1;
            for (int q = 0; q < fan; q++) {
                ux0 = min(ux0, s_box[0][q]); ux1 = max(ux1, s_box[1][q]);
                uy0 = min(uy0, s_box[2][q]); uy1 = max(uy1, s_box[3][q]);
            }
            if (P.counters) atomicAdd(&P.counters[1], (unsigned long long)fan);
            if (fan > 0 && ux0 <= ux1 && uy0 <= uy1) {
                TriRec head;
                head.ax = __uint_as_float(block);
                head.ay = head.abx = head.aby = head.acx = head.acy = head.factor = 0;
                head.z[0] = head.z[1] = head.z[2] = head.rw[0] = head.rw[1] = head.rw[2] = 0;
                head.meta = (uint32_t)d | (1u << 17) | ((uint32_t)fan << 18);
                head.bx = (uint32_t)ux0 | ((uint32_t)ux1 << 16);
                head.by = (uint32_t)uy0 | ((uint32_t)uy1 << 16);
                store_tri(&P.tri[t], head);
                if (!D.sky_slot) P.bbox[t] = pack_bin_box(head.bx, head.by, (uint32_t)fan);    /* (a sky box on the pass's table is kept out of the bins) */
            }
        }
    }
}

/* ========================================================================
 * K3  binning: for each 64x64-pixel bin the triangles whose bounding box
 *     touches it, in ascending triangle index (= submission order).
 *     Integer only.  One workgroup per bin streams the packed boxes twice:
 *     count, reserve a contiguous range with one atomic, then fill in order.
 * ======================================================================== */

/* a fragment / list reference: a triangle index, or RZ_REF_FAN | index into fan_tri */
#define RZ_REF_NONE 0xffffffffu
#define RZ_REF_FAN 0x80000000u
#define RZ_REF_FAMILY_BITS 0x30000000u  /* in the reference plane only: the fragment's shader family */

RZ_DEV bool box_overlaps(uint2 bb, int rx0, int ry0, int rx1, int ry1) {
    int x0 = (int)(bb.x & 0xffffu), x1 = (int)(bb.x >> 16);
    int y0 = (int)(bb.y & 0xffffu), y1 = (int)(bb.y >> 16);
    return x0 <= rx1 && x1 >= rx0 && y0 <= ry1 && y1 >= ry0;
}

/* The triangle range is cut into chunks of 64 (one wave each).  counts[c][b]
 * = list entries chunk c adds to bin b.  Three small kernels:
 *   count  every triangle adds its entry count to the bins its box touches
 *   scan   per bin, exclusive prefix over the chunks -> where each chunk's
 *          entries start in the bin's list; one atomic reserves the list
 *   fill   every chunk writes its entries bin by bin, lanes in order
 * so a list is ascending in (chunk, lane, fan index) = submission order. */

struct BinRect { int bx0, bx1, by0, by1; uint32_t entries; };

/* bins touched by a packed box (pack_bin_box); entries == 0: none */
RZ_DEV BinRect bin_rect_of(uint2 bb) {
    BinRect r;
    int x0 = (int)(bb.x & 0xffffu), x1 = (int)(bb.x >> 16);
    int y0 = (int)(bb.y & 0x3fffu), y1 = (int)((bb.y >> 14) & 0x3fffu);
    r.entries = x0 <= x1 ? bb.y >> 28 : 0u;
    r.bx0 = x0 / RZ_BIN; r.bx1 = x1 / RZ_BIN; r.by0 = y0 / RZ_BIN; r.by1 = y1 / RZ_BIN;
    return r;
}


/* Cross-lane steps of scans and reductions go through the VALU's data-parallel primitives (DPP): the
 * operand comes from a neighbouring lane's register inside the instruction itself, where __shfl makes a
 * trip through the LDS crossbar (ds_bpermute) and waits for it.  A lane whose source lies outside its row
 * of 16, or whose row the mask leaves out, gets `otherwise`. */
#define RZ_DPP_ROW_SHR(n) (0x110 + (n))     /* lane i reads lane i - n of its row */
#define RZ_DPP_ROW_BCAST15 0x142            /* every lane of a row reads lane 15 of the row before */
#define RZ_DPP_ROW_BCAST31 0x143            /* rows 2 and 3 read lane 31 */
template <int CTRL, int ROW_MASK = 0xf>
RZ_DEV int dpp_from(int otherwise, int v) { return __builtin_amdgcn_update_dpp(otherwise, v, CTRL, ROW_MASK, 0xf, false); }

/* all 64 lanes must be active */
RZ_DEV uint32_t wave_exclusive_scan(uint32_t v, uint32_t *total) {
    int x = (int)v;
    x += dpp_from<RZ_DPP_ROW_SHR(1)>(0, x);
    x += dpp_from<RZ_DPP_ROW_SHR(2)>(0, x);
    x += dpp_from<RZ_DPP_ROW_SHR(4)>(0, x);
    x += dpp_from<RZ_DPP_ROW_SHR(8)>(0, x);                 /* inclusive within each row of 16 */
    x += dpp_from<RZ_DPP_ROW_BCAST15, 0xa>(0, x);           /* rows 1 and 3 add the row before them */
    x += dpp_from<RZ_DPP_ROW_BCAST31, 0xc>(0, x);           /* rows 2 and 3 add the first half */
    *total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
    return (uint32_t)x - v;
}

/* the rows' results end up in their last lanes and are combined on the scalar side */
RZ_DEV int wave_min_i(int v) {
    v = min(v, dpp_from<RZ_DPP_ROW_SHR(1)>(v, v));
    v = min(v, dpp_from<RZ_DPP_ROW_SHR(2)>(v, v));
    v = min(v, dpp_from<RZ_DPP_ROW_SHR(4)>(v, v));
    v = min(v, dpp_from<RZ_DPP_ROW_SHR(8)>(v, v));
    return min(min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               min(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

RZ_DEV int wave_max_i(int v) {
    v = max(v, dpp_from<RZ_DPP_ROW_SHR(1)>(v, v));
    v = max(v, dpp_from<RZ_DPP_ROW_SHR(2)>(v, v));
    v = max(v, dpp_from<RZ_DPP_ROW_SHR(4)>(v, v));
    v = max(v, dpp_from<RZ_DPP_ROW_SHR(8)>(v, v));
    return max(max(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31)),
               max(__builtin_amdgcn_readlane(v, 47), __builtin_amdgcn_readlane(v, 63)));
}

#define RZ_FILL_FEW_BINS 16         /* a chunk whose joint bin rectangle is at most this large is counted and filled triangle-parallel */
__global__ void __launch_bounds__(256) rz_bin_count_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if ((t & ~63u) >= (uint32_t)P.num_tris) return;             /* uniform over the wave */
    BinRect r;
    r.entries = 0; r.bx0 = r.bx1 = r.by0 = r.by1 = 0;
    if (t < (uint32_t)P.num_tris) r = bin_rect_of(P.bbox[t]);
    uint32_t *row = P.bin_counts + (size_t)(t >> 6) * (uint32_t)(P.bins_x * P.bins_y);
    const int w = r.bx1 - r.bx0 + 1, area = w * (r.by1 - r.by0 + 1);
    /* A wave is a chunk and the chunk's row of the matrix is its own.  A dense mesh puts all 64 triangles into a
     * handful of bins: an atomic per triangle and bin then hits the same few words 64 times over.  So when the
     * chunk's joint bin rectangle is small its bins are taken one after the other, the triangles that touch a bin
     * counted with a ballot (summed when a clipped triangle adds a whole fan), and lane i stores bin i's count
     * (azura at 800x600: 51 -> 8 us per launch of 32 passes). */
    const unsigned long long active = __ballot(r.entries != 0);
    if (!active) return;
    const int ux0 = wave_min_i(r.entries ? r.bx0 : 0x7fffffff), ux1 = wave_max_i(r.entries ? r.bx1 : -1);
    const int uy0 = wave_min_i(r.entries ? r.by0 : 0x7fffffff), uy1 = wave_max_i(r.entries ? r.by1 : -1);
    const int uw = ux1 - ux0 + 1, ubins = uw * (uy1 - uy0 + 1);
    if (ubins <= RZ_FILL_FEW_BINS) {
        const bool fans = __ballot(r.entries > 1u) != 0ull;
        uint32_t mine = 0;
        for (int i = 0; i < ubins; i++) {
            const int bx = ux0 + i % uw, by = uy0 + i / uw;
            const bool touch = r.entries != 0 && bx >= r.bx0 && bx <= r.bx1 && by >= r.by0 && by <= r.by1;
            uint32_t total;
            if (fans) (void)wave_exclusive_scan(touch ? r.entries : 0u, &total);
            else total = (uint32_t)__popcll(__ballot(touch));
            if (lane == i) mine = total;
        }
        if (lane < ubins && mine) row[(uy0 + lane / uw) * P.bins_x + ux0 + lane % uw] = mine;
        return;
    }
    /* a triangle over many bins (a sky box face) is spread over the wave's lanes */
    unsigned long long big = __ballot(r.entries != 0 && area > 8);
    if (r.entries != 0 && area <= 8) {
        for (int by = r.by0; by <= r.by1; by++)
            for (int bx = r.bx0; bx <= r.bx1; bx++) atomicAdd(&row[by * P.bins_x + bx], r.entries);
    }
    while (big) {
        const int L = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int x0 = RZ_BCAST_I(r.bx0, L), y0 = RZ_BCAST_I(r.by0, L), wL = RZ_BCAST_I(w, L), aL = RZ_BCAST_I(area, L);
        const uint32_t e = (uint32_t)RZ_BCAST_I(r.entries, L);
        for (int i = lane; i < aL; i += 64) atomicAdd(&row[(y0 + i / wL) * P.bins_x + x0 + i % wL], e);
    }
}

/* 1024 threads = 64 bins x 16 segments of the chunk range */
#define RZ_SCAN_SEGMENTS 16
__global__ void __launch_bounds__(64 * RZ_SCAN_SEGMENTS) rz_bin_scan_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    __shared__ uint32_t s_part[RZ_SCAN_SEGMENTS][64];
    const int bins = P.bins_x * P.bins_y;
    if ((int)(blockIdx.x * 64u) >= bins) return;
    const int bin = (int)(blockIdx.x * 64u) + (threadIdx.x & 63), segment = threadIdx.x >> 6;
    const uint32_t chunks = ((uint32_t)P.num_tris + 63u) / 64u;
    const uint32_t per = (chunks + RZ_SCAN_SEGMENTS - 1u) / RZ_SCAN_SEGMENTS;
    const uint32_t c0 = min((uint32_t)segment * per, chunks), c1 = min(c0 + per, chunks);
    uint32_t *col = P.bin_counts + bin;
    uint32_t sum = 0;
    if (bin < bins) {
        for (uint32_t c = c0; c < c1; c += 8u) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (c + u) < c1 ? col[(size_t)(c + u) * bins] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) sum += v[u];
        }
    }
    s_part[segment][threadIdx.x & 63] = sum;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int q = 0; q < RZ_SCAN_SEGMENTS; q++) {
        uint32_t v = s_part[q][threadIdx.x & 63];
        if (q < segment) before += v;
        total += v;
    }
    /* every segment's thread of a bin needs the list's start: segment 0 reserves it */
    __syncthreads();
    if (segment == 0 && bin < bins) {
        uint32_t start = total ? atomicAdd(&P.control[1], total) : 0u;
        if (start + total > P.bin_capacity) {      /* control[1] still ends up as the entries the pass needs */
            atomicOr(&P.control[2], (uint32_t)RZ_ERR_BIN_OVERFLOW);
            atomicOr(P.flush_flag, 1u);
            total = 0;
        }
        P.bin_range[bin] = make_uint2(start, total);
        s_part[0][threadIdx.x & 63] = total ? start : 0xffffffffu;
        /* a pass with blended draws: the bin's 32x8 tiles go on the work list of the kernel that rasterises it, unless
         * there is nothing for that kernel to do in them */
        if (P.scan_list >= 0 && (total != 0u || P.clear_color_pending || P.clear_depth_pending)) {
            const uint32_t tx_first = (uint32_t)(bin % P.bins_x) * (RZ_BIN / RZ_TILE_W), ty_first = (uint32_t)(bin / P.bins_x) * (RZ_BIN / RZ_TILE_H);
            for (uint32_t ty = ty_first; ty < min(ty_first + RZ_BIN / RZ_TILE_H, (uint32_t)P.tiles_y); ty++)
                for (uint32_t tx = tx_first; tx < min(tx_first + RZ_BIN / RZ_TILE_W, (uint32_t)P.tiles_x); tx++) {
                    const uint32_t tile = ty * (uint32_t)P.tiles_x + tx;
                    /* by the tile's index in the batch: no part can get more than its even share */
                    const uint32_t part = (uint32_t)P.scan_list * RZ_QUEUE_WAYS + ((P.tile_off + tile) & (RZ_QUEUE_WAYS - 1u));
                    const uint32_t at = atomicAdd(&P.queue_counts[part * RZ_QUEUE_PAD], 1u);
                    if (at < P.queue_capacity)
                        P.queue[(size_t)part * P.queue_capacity + at] = ((uint32_t)blockIdx.y << 20) | (ty << 9) | tx;
                    else
                        atomicOr(&P.control[2], (uint32_t)RZ_ERR_QUEUE_OVERFLOW);
                }
        }
    }
    if (segment == 0) {             /* the pass's longest list, for the host's choice of strip height next time */
        uint32_t longest = bin < bins ? total : 0u;
        longest = (uint32_t)wave_max_i((int)longest);           /* entry counts stay far below 2^31 */
        if ((threadIdx.x & 63) == 0 && longest) atomicMax(&P.control[4], longest);
    }
    __syncthreads();
    const uint32_t start = s_part[0][threadIdx.x & 63];
    if (bin >= bins || start == 0xffffffffu) return;    /* empty or overflowed bin: its entries are never written */
    uint32_t run = start + before;
    for (uint32_t c = c0; c < c1; c += 8u) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (c + u) < c1 ? col[(size_t)(c + u) * bins] : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if ((c + u) < c1) col[(size_t)(c + u) * bins] = run;    /* count -> first entry of the chunk */
            run += v[u];
        }
    }
}

#define RZ_FILL_FAN_SLOTS 64

/* what the list walk needs of an entry, from the last quarter of its record */
RZ_DEV uint2 entry_box(float4 q3) {
    return make_uint2(__float_as_uint(q3.z), __float_as_uint(q3.w) | (RZ_META_CUTOFF(__float_as_uint(q3.y)) << 31));
}

/* a triangle's entries (one, or its fan's) into a bin's list from position `at` on; fan records come from the
 * chunk's staging area in LDS when they fitted there */
RZ_DEV void fill_entries(const PassParams &P, uint32_t at, uint32_t ref, uint32_t entries, float4 r0, float4 r1, float4 r2, float4 r3,
                         uint32_t slot, const float4 (*s_fan)[4], const TriRec *__restrict__ fan_tri) {
    if (!(ref & RZ_REF_FAN)) {
        float4 *dst = (float4*)&P.bin_list[at];
        dst[0] = make_float4(__uint_as_float(ref), 0.0f, 0.0f, 0.0f);
        dst[1] = r0; dst[2] = r1; dst[3] = r2; dst[4] = r3;
        P.bin_boxes[at] = entry_box(r3);
        return;
    }
    for (uint32_t k = 0; k < entries; k++) {
        float4 q0, q1, q2, q3;
        if (slot != 0xffffffffu) {
            q0 = s_fan[slot + k][0]; q1 = s_fan[slot + k][1]; q2 = s_fan[slot + k][2]; q3 = s_fan[slot + k][3];
        } else {
            const float4 *src = (const float4*)&fan_tri[(ref & ~RZ_REF_FAN) + k];
            q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
        }
        float4 *dst = (float4*)&P.bin_list[at + k];
        dst[0] = make_float4(__uint_as_float(ref + k), 0.0f, 0.0f, 0.0f);
        dst[1] = q0; dst[2] = q1; dst[3] = q2; dst[4] = q3;
        P.bin_boxes[at + k] = entry_box(q3);
    }
}

/* One workgroup of WAVES waves per chunk.  Lanes first hold the
 * chunk's 64 triangles (box, record), then turn into BINS: the chunk's joint
 * bin rectangle is cut into rounds of 64 bins, wave w takes rounds w,
 * w + WAVES, ..., lane i of a round one bin, and walks the 64
 * triangles in order (their data broadcast with readlane), appending those
 * that touch its bin.  No lane waits on another and the only dependent loads
 * are the per-bin start offsets.  Every entry carries a copy of the
 * triangle's record. */
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) rz_bin_fill_kernel(const PassParams *__restrict__ passes, uint32_t num_passes) {
    /* Workgroups start in index order: the pass is the fastest part of the index and a pass's chunks are taken
     * from its last one down.  The chunks that run longest are those of triangles over many bins, i.e. the sky
     * box, which a frame draws last (test_helper.c:497-503): they start first in every pass instead of making
     * up the launch's tail. */
    const PassParams &P = passes[blockIdx.x % num_passes];
    const uint32_t chunks = ((uint32_t)P.num_tris + 63u) / 64u, turn = blockIdx.x / num_passes;
    if (turn >= chunks) return;                                 /* uniform over the workgroup */
    const uint32_t chunk = chunks - 1u - turn;
    if (P.control[2] != 0u) return;                             /* out of scratch, or left out: the pass is run again later */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t = chunk * 64u + lane;
    const int bins = P.bins_x * P.bins_y;
    BinRect r;
    r.entries = 0; r.bx0 = r.bx1 = r.by0 = r.by1 = 0;
    if (t < (uint32_t)P.num_tris) r = bin_rect_of(P.bbox[t]);
    const unsigned long long active = __ballot(r.entries != 0);
    if (!active) return;                                        /* same for every wave of the workgroup */
    const int ux0 = wave_min_i(r.entries ? r.bx0 : 0x7fffffff), ux1 = wave_max_i(r.entries ? r.bx1 : -1);
    const int uy0 = wave_min_i(r.entries ? r.by0 : 0x7fffffff), uy1 = wave_max_i(r.entries ? r.by1 : -1);
    const int uw = ux1 - ux0 + 1, ubins = uw * (uy1 - uy0 + 1);
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    uint32_t ref = t;
    if (r.entries) {
        const float4 *src = (const float4*)&P.tri[t];
        r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
        if (RZ_META_CLIPPED(__float_as_uint(r3.y)))
            ref = RZ_REF_FAN | (__float_as_uint(r0.x) * RZ_MAX_FAN);     /* first fan record */
    }
    /* the chunk's fan records, staged in LDS by wave 0 so that the walk below has no dependent
     * global loads (beyond RZ_FILL_FAN_SLOTS records per chunk they are read from memory) */
    __shared__ float4 s_fan[RZ_FILL_FAN_SLOTS][4];
    uint32_t slot = 0xffffffffu;
    {
        const bool clipped = r.entries && (ref & RZ_REF_FAN);
        const uint32_t need = clipped ? r.entries : 0u;
        uint32_t total;
        const uint32_t first = wave_exclusive_scan(need, &total);
        if (clipped && first + need <= RZ_FILL_FAN_SLOTS) {
            slot = first;
            if (wave == 0) {
                for (uint32_t k = 0; k < need; k++) {
                    const float4 *src = (const float4*)&P.fan_tri[(ref & ~RZ_REF_FAN) + k];
                    for (int q = 0; q < 4; q++) s_fan[slot + k][q] = src[q];
                }
            }
        }
        if (total) __syncthreads();                             /* total is the same in every wave */
    }
    const uint32_t *row = P.bin_counts + (size_t)chunk * (uint32_t)bins;
    const TriRec *__restrict__ fan_tri = P.fan_tri;

    if (ubins <= RZ_FILL_FEW_BINS) {
        /* The chunk's triangles lie in a few bins (a dense mesh: most chunks).  Lanes stay
         * triangles and the few bins are taken one after the other: the triangles that touch the
         * bin find their places behind the chunk's start in its list by a prefix over the lanes
         * (ascending lane = submission order, as below) and write their entries side by side. */
        if (wave != 0) return;
        /* lane i asks for bin i's start and whether its list is kept, all bins in one memory round trip
         * (asked for bin after bin inside the loop they were one dependent round trip each) */
        uint32_t my_start = 0u, my_kept = 0u;
        if (lane < ubins) {
            const int b = (uy0 + lane / uw) * P.bins_x + ux0 + lane % uw;
            my_start = row[b];
            my_kept = P.bin_range[b].y;         /* zero: the list is empty or overflowed, nothing is kept */
        }
        for (int i = 0; i < ubins; i++) {
            const int bx = ux0 + i % uw, by = uy0 + i / uw;
            const bool touch = r.entries != 0 && bx >= r.bx0 && bx <= r.bx1 && by >= r.by0 && by <= r.by1;
            const unsigned long long touching = __ballot(touch);
            const uint32_t start = (uint32_t)RZ_BCAST_I(my_start, i);
            if (!touching || RZ_BCAST_I(my_kept, i) == 0) continue;
            uint32_t before;
            if (__ballot(touch && r.entries != 1u) == 0) {
                before = __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
            } else {
                uint32_t total;
                before = wave_exclusive_scan(touch ? r.entries : 0u, &total);
            }
            if (touch) fill_entries(P, start + before, ref, r.entries, r0, r1, r2, r3, slot, s_fan, fan_tri);
        }
        return;
    }

    const int w = r.bx1 - r.bx0 + 1, h = r.by1 - r.by0 + 1;
    if (__ballot(r.entries != 0 && (w > 3 || h > 3)) == 0ull) {
        /* Small triangles that lie far apart (consecutive triangles of different parts of a mesh): the joint rectangle
         * is large and nearly empty.  Lanes stay triangles; each walks the (at most 3 x 3) bins of its own box in
         * ascending bin index, the wave takes the lowest bin any lane is at, the lanes at that bin find their places
         * with a prefix over the lanes as above and move on.  One turn per DISTINCT bin the chunk touches, where the
         * rounds below make 64 turns per 64 bins of the rectangle.  The starts of a lane's bins are asked for up front,
         * in one memory round trip. */
        if (wave != 0) return;
        uint32_t start[9];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const bool in_box = r.entries != 0 && q % 3 < w && q / 3 < h;
            const int b = in_box ? (r.by0 + q / 3) * P.bins_x + r.bx0 + q % 3 : 0;     /* (every lane loads: nothing waits on a branch) */
            const uint32_t first = row[b], kept = P.bin_range[b].y;
            start[q] = in_box && kept != 0u ? first : 0xffffffffu;      /* not mine, or a list that is not kept (empty or overflowed) */
        }
        int px = 0, py = 0;
        bool more = r.entries != 0;
        for (;;) {
            const int mine = more ? (r.by0 + py) * P.bins_x + r.bx0 + px : 0x7fffffff;
            const int b = wave_min_i(mine);
            if (b == 0x7fffffff) break;
            const bool touch = mine == b;
            const unsigned long long touching = __ballot(touch);
            uint32_t before;
            if (__ballot(touch && r.entries != 1u) == 0) {
                before = __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
            } else {
                uint32_t total;
                before = wave_exclusive_scan(touch ? r.entries : 0u, &total);
            }
            if (touch) {
                const int at_slot = py * 3 + px;
                uint32_t at = start[0];
#pragma unroll
                for (int q = 1; q < 9; q++) if (at_slot == q) at = start[q];
                if (at != 0xffffffffu) fill_entries(P, at + before, ref, r.entries, r0, r1, r2, r3, slot, s_fan, fan_tri);
                if (++px == w) { px = 0; py++; }
                more = py < h;
            }
        }
        return;
    }

    for (int base = wave * 64; base < ubins; base += 64 * WAVES) {
        const int i = base + lane;
        const bool have = i < ubins;
        const int bx = ux0 + i % uw, by = uy0 + i / uw;
        const int b = by * P.bins_x + bx;
        uint32_t cursor = 0;
        bool keep = false;
        if (have) {
            cursor = row[b];
            keep = P.bin_range[b].y != 0;       /* zero: the list is empty or overflowed, nothing is kept */
        }
        unsigned long long todo = active;
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int x0 = RZ_BCAST_I(r.bx0, L), x1 = RZ_BCAST_I(r.bx1, L);
            const int y0 = RZ_BCAST_I(r.by0, L), y1 = RZ_BCAST_I(r.by1, L);
            const bool hit = keep && bx >= x0 && bx <= x1 && by >= y0 && by <= y1;
            if (!__any(hit)) continue;
            const uint32_t e = (uint32_t)RZ_BCAST_I(r.entries, L);
            const uint32_t refL = (uint32_t)RZ_BCAST_I(ref, L);
            if (!(refL & RZ_REF_FAN)) {
                const float4 q0 = make_float4(RZ_BCAST_F(r0.x, L), RZ_BCAST_F(r0.y, L), RZ_BCAST_F(r0.z, L), RZ_BCAST_F(r0.w, L));
                const float4 q1 = make_float4(RZ_BCAST_F(r1.x, L), RZ_BCAST_F(r1.y, L), RZ_BCAST_F(r1.z, L), RZ_BCAST_F(r1.w, L));
                const float4 q2 = make_float4(RZ_BCAST_F(r2.x, L), RZ_BCAST_F(r2.y, L), RZ_BCAST_F(r2.z, L), RZ_BCAST_F(r2.w, L));
                const float4 q3 = make_float4(RZ_BCAST_F(r3.x, L), RZ_BCAST_F(r3.y, L), RZ_BCAST_F(r3.z, L), RZ_BCAST_F(r3.w, L));
                if (hit) {
                    float4 *dst = (float4*)&P.bin_list[cursor];
                    dst[0] = make_float4(__uint_as_float(refL), 0.0f, 0.0f, 0.0f);
                    dst[1] = q0; dst[2] = q1; dst[3] = q2; dst[4] = q3;
                    P.bin_boxes[cursor] = entry_box(q3);
                    cursor += 1u;
                }
            } else {
                const uint32_t slotL = (uint32_t)RZ_BCAST_I(slot, L);
                for (uint32_t k = 0; k < e; k++) {
                    float4 q0, q1, q2, q3;
                    if (slotL != 0xffffffffu) {
                        q0 = s_fan[slotL + k][0]; q1 = s_fan[slotL + k][1];
                        q2 = s_fan[slotL + k][2]; q3 = s_fan[slotL + k][3];
                    } else {
                        const float4 *src = (const float4*)&fan_tri[(refL & ~RZ_REF_FAN) + k];  /* uniform address */
                        q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];
                    }
                    if (hit) {
                        float4 *dst = (float4*)&P.bin_list[cursor + k];
                        dst[0] = make_float4(__uint_as_float(refL + k), 0.0f, 0.0f, 0.0f);
                        dst[1] = q0; dst[2] = q1; dst[3] = q2; dst[4] = q3;
                        P.bin_boxes[cursor + k] = entry_box(q3);
                    }
                }
                if (hit) cursor += e;
            }
        }
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

/* What draw_fragment leaves in the framebuffer for a skybox fragment (graphics.c:452-468; the program
 * does not blend): the shader's colour is its cube map's texel, so saturating it, scaling by 255 and
 * truncating is a function of the texel alone -- worked out once per texel when the map is first drawn
 * (rz_texture_display_kernel; the six faces in one plane, so that a lane's face is an offset, not a
 * pointer to fetch), the fragment then reads 4 bytes instead of 16 and converts nothing. */
/* where a direction's texel sits in the cube's plane of display bytes (RZ_DRAW_SKY_BYTES draws):
 * maps[0] describes all six faces, one behind the other; size and layout are scalars */
RZ_DEV uint32_t sky_bytes_index(const DrawRec &D, const float *v) {
    float s, t;
    const int face = cube_face(mk3(v[0], v[1], v[2]), &s, &t);
    const int w = D.maps[0].w, h = D.maps[0].h, flags = D.maps[0].flags;
    const float x = s, y = 1 - t;                                   /* texture.c:257, 160-167 */
    return __umul24((uint32_t)face, (uint32_t)(w * h)) + texel_cell(w, h, flags, x - floorf(x), y - floorf(y));
}
RZ_DEV uint32_t sky_fragment(const DrawRec &D, const float *v, uint32_t below) {
    if (D.flags & RZ_DRAW_SKY_BYTES) {
        const uint32_t word = D.maps[0].data.as<const uint32_t>()[sky_bytes_index(D, v)];
        return (word & 0x00ffffffu) | (below & 0xff000000u);
    }
    f4 color = fs_skybox(D, v);
    color = mk4(d_sat(color.x), d_sat(color.y), d_sat(color.z), d_sat(color.w));
    return pack_rgb(color, below);
}

/* ---- deferred, order-exact shading ---------------------------------------
 *
 * The reference runs the fragment shader for every fragment that passes the
 * depth test, in submission order (graphics.c:523-529).  Its result only
 * matters for the fragments that are still visible at the end, so each lane
 * (= pixel) keeps a short stack of REFERENCES to the triangles whose fragments
 * make up its colour and shades them when the list ends:
 *   - an opaque fragment replaces the stack (everything below it is dead);
 *   - a blended fragment is pushed on top (it reads the colour below it);
 *   - fragments the shader would discard (alpha cutoff) never enter: the
 *     cutoff test is evaluated at once, it needs only alpha;
 *   - depth is updated at once in both cases (draw_fragment writes depth for
 *     blended fragments too, graphics.c:469).
 * When a blended fragment meets a full stack the wave resolves all stacks and
 * goes on.  Resolving shades level by level, all lanes of one draw together
 * whatever triangle each lane refers to, so the shader runs on full waves and
 * at exactly one place in the kernel.
 */


/* (s0*q0 + s1*q1 + s2*q2) * norm per float (graphics.c:431-434), per lane */
template <int NV>
RZ_DEV void interpolate_lane(const float *v0, const float *v1, const float *v2,
                             float q0, float q1, float q2, float norm, float *out) {
    constexpr int N4 = (NV + 3) / 4;            /* the 96-byte stride makes whole float4 reads safe */
    for (int i = 0; i < N4; i++) {
        float4 a = ((const float4*)v0)[i], b = ((const float4*)v1)[i], c = ((const float4*)v2)[i];
        if (4 * i + 0 < NV) out[4 * i + 0] = (a.x * q0 + b.x * q1 + c.x * q2) * norm;
        if (4 * i + 1 < NV) out[4 * i + 1] = (a.y * q0 + b.y * q1 + c.y * q2) * norm;
        if (4 * i + 2 < NV) out[4 * i + 2] = (a.z * q0 + b.z * q1 + c.z * q2) * norm;
        if (4 * i + 3 < NV) out[4 * i + 3] = (a.w * q0 + b.w * q1 + c.w * q2) * norm;
    }
}

/* The same for corners that are the same for every lane (one triangle under the whole wave).  The varyings
 * were written by an earlier launch and nothing in this one changes them: read through the constant
 * address space, which is how the compiler is told so -- the loads then go through the scalar unit; as
 * ordinary global loads they went through the vector memory pipeline (its data path was the sky kernel's
 * bound), because the choice between a triangle's and a fan's varyings hides where the pointer came from. */
#define RZ_CONSTANT __attribute__((address_space(4)))
typedef float rz_float4 __attribute__((ext_vector_type(4)));
RZ_DEV void interpolate_uniform3(const float *v0, const float *v1, const float *v2,
                                 float q0, float q1, float q2, float norm, float *out) {
    const rz_float4 a = *(const rz_float4 RZ_CONSTANT*)v0, b = *(const rz_float4 RZ_CONSTANT*)v1, c = *(const rz_float4 RZ_CONSTANT*)v2;
    out[0] = (a.x * q0 + b.x * q1 + c.x * q2) * norm;
    out[1] = (a.y * q0 + b.y * q1 + c.y * q2) * norm;
    out[2] = (a.z * q0 + b.z * q1 + c.z * q2) * norm;
}

/* the same for floats FIRST .. NV - 1 only (whole groups of four are fetched) */
template <int FIRST, int NV>
RZ_DEV void interpolate_lane_from(const float *v0, const float *v1, const float *v2,
                                  float q0, float q1, float q2, float norm, float *out) {
    for (int i = FIRST / 4; i < (NV + 3) / 4; i++) {
        float4 a = ((const float4*)v0)[i], b = ((const float4*)v1)[i], c = ((const float4*)v2)[i];
        if (4 * i + 0 >= FIRST && 4 * i + 0 < NV) out[4 * i + 0] = (a.x * q0 + b.x * q1 + c.x * q2) * norm;
        if (4 * i + 1 >= FIRST && 4 * i + 1 < NV) out[4 * i + 1] = (a.y * q0 + b.y * q1 + c.y * q2) * norm;
        if (4 * i + 2 >= FIRST && 4 * i + 2 < NV) out[4 * i + 2] = (a.z * q0 + b.z * q1 + c.z * q2) * norm;
        if (4 * i + 3 >= FIRST && 4 * i + 3 < NV) out[4 * i + 3] = (a.w * q0 + b.w * q1 + c.w * q2) * norm;
    }
}

/* barycentric weights of the pixel centre (graphics.c:385-397, 514) */
RZ_DEV void pixel_weights(const TriRec &r, int px, int py, float &w0, float &w1, float &w2) {
    float apx = ((float)px + 0.5f) - r.ax, apy = ((float)py + 0.5f) - r.ay;
    float s = (r.acy * apx - r.acx * apy) * r.factor;
    float t = (r.abx * apy - r.aby * apx) * r.factor;
    w0 = 1 - s - t; w1 = s; w2 = t;
}

/* Shades the fragments referenced by `want` lanes and merges them into ccur.
 * Each lane carries its own triangle record and varying pointers; lanes of one
 * draw are shaded together under that draw's (wave-uniform) record.
 * FAMILY >= 0 compiles only that family's shader (per-family kernels),
 * FAMILY < 0 all three. */
template <int FAMILY, bool NARROW = false, class TABLES = GlobalTables>
RZ_DEV void shade_lanes(const PassParams &P, const DrawRec *__restrict__ draw_table, bool want,
                        const TriRec &m, const float *m0, const float *m1, const float *m2,
                        int px, int py, uint32_t &ccur) {
    const int lane = threadIdx.x & 63;
    uint32_t mydraw = want ? RZ_META_DRAW(m.meta) : 0xffffffffu;
    unsigned long long rest = __ballot(want);
    while (rest) {
        int leader = __ffsll((long long)rest) - 1;
        uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)mydraw, leader);
        bool sel = want && mydraw == d;
        rest &= ~__ballot(sel);
        const DrawRec &D = draw_table[d];
        if (sel) {
            float w0, w1, w2;
            pixel_weights(m, px, py, w0, w1, w2);
            float q0 = m.rw[0] * w0, q1 = m.rw[1] * w1, q2 = m.rw[2] * w2;          /* graphics.c:426-429 */
            float norm = 1 / (q0 + q1 + q2);
            bool backface = RZ_META_BACKFACE(m.meta) != 0;
            bool discard = false;
            bool continue_to_count = false;                         /* the colour is in ccur already */
            f4 color = mk4(0, 0, 0, 0);
            if ((FAMILY < 0 || FAMILY == 2) && D.family == 2) {
                float v[4];
                interpolate_lane<3>(m0, m1, m2, q0, q1, q2, norm, v);
                if (D.flags & RZ_DRAW_SKY_BYTES) {                  /* never set for a program that blends */
                    ccur = sky_fragment(D, v, ccur);
                    continue_to_count = true;
                } else
                    color = fs_skybox(D, v);
            } else if ((FAMILY < 0 || FAMILY == 0) && D.family == 0) {
                float v[12];
                interpolate_lane<11>(m0, m1, m2, q0, q1, q2, norm, v);
                color = fs_blinn<NARROW, TABLES>(D, v, backface, &discard);
            } else if ((FAMILY < 0 || FAMILY == 1) && D.family == 1) {
                float v[24];
                interpolate_lane<RZ_PV_ALWAYS>(m0, m1, m2, q0, q1, q2, norm, v);
                for (int k = RZ_PV_ALWAYS; k < RZ_PV_COUNT; k++) v[k] = 0;
                if (pbr_reads_positions(D))             /* uniform over the draw */
                    interpolate_lane_from<RZ_PV_ALWAYS, RZ_PV_COUNT>(m0, m1, m2, q0, q1, q2, norm, v);
                color = fs_pbr<NARROW, TABLES>(D, v, backface, &discard);
            }
            /* draw_fragment (graphics.c:452-468); a discard was settled during rasterisation */
            if (!continue_to_count) {
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
            }
        }
        if (P.counters) {       /* [8] fragments actually shaded on the device */
            unsigned long long n = (unsigned long long)__popcll(__ballot(sel));
            if (lane == 0) {
                atomicAdd(P.counters + 8, n);
                if (D.family == 2) atomicAdd(P.counters + 9, n);    /* [9] of those, skybox */
            }
        }
    }
}

/* the three corner varying blocks a reference stands for */
RZ_DEV void ref_varyings(uint32_t ref, const float *__restrict__ vary_table, const float *__restrict__ fan_vary_table,
                         const float *&m0, const float *&m1, const float *&m2) {
    if (ref & RZ_REF_FAN) {
        uint32_t idx = ref & ~RZ_REF_FAN, blk = idx / RZ_MAX_FAN, k = idx % RZ_MAX_FAN;
        const float *fv = fan_vary_table + (size_t)blk * RZ_MAX_POLY * RZ_VARY_STRIDE;
        m0 = fv; m1 = fv + (k + 1) * RZ_VARY_STRIDE; m2 = fv + (k + 2) * RZ_VARY_STRIDE;
    } else {
        m0 = vary_table + (size_t)(3u * ref) * RZ_VARY_STRIDE; m1 = m0 + RZ_VARY_STRIDE; m2 = m1 + RZ_VARY_STRIDE;
    }
}

/* loads the record and varying pointers a fragment reference stands for */
RZ_DEV void fetch_ref(uint32_t ref, const TriRec *__restrict__ tri_table, const TriRec *__restrict__ fan_table,
                      const float *__restrict__ vary_table, const float *__restrict__ fan_vary_table,
                      TriRec &m, const float *&m0, const float *&m1, const float *&m2) {
    m = load_tri((ref & RZ_REF_FAN) ? &fan_table[ref & ~RZ_REF_FAN] : &tri_table[ref]);
    ref_varyings(ref, vary_table, fan_vary_table, m0, m1, m2);
}

/* Conservative "this triangle cannot cover any pixel centre of the block":
 * each barycentric weight is an affine function of the pixel, so its maximum
 * over the block sits at a corner; the corner values are computed with the
 * rasteriser's own expressions, and a weight is taken to be below the
 * -epsilon threshold everywhere only if it is so at all four corners by more
 * than a bound on the rounding error of those expressions (16 ulp of the
 * magnitudes involved, twice: corner and pixel).  NaN compares false, so
 * degenerate triangles are never rejected here -- the pixel test drops them. */
RZ_DEV bool block_outside(float4 c0, float4 c1, float x0, float x1, float y0, float y1) {
    const float ax = c0.x, ay = c0.y, abx = c0.z, aby = c0.w, acx = c1.x, acy = c1.y, factor = c1.z;
    const float u = 1.9073486e-06f;             /* 2^-19 */
    float dx0 = x0 - ax, dx1 = x1 - ax, dy0 = y0 - ay, dy1 = y1 - ay;
    float s00 = (acy * dx0 - acx * dy0) * factor, s10 = (acy * dx1 - acx * dy0) * factor;
    float s01 = (acy * dx0 - acx * dy1) * factor, s11 = (acy * dx1 - acx * dy1) * factor;
    float t00 = (abx * dy0 - aby * dx0) * factor, t10 = (abx * dy0 - aby * dx1) * factor;
    float t01 = (abx * dy1 - aby * dx0) * factor, t11 = (abx * dy1 - aby * dx1) * factor;
    float mx = fmaxf(fabsf(dx0), fabsf(dx1)), my = fmaxf(fabsf(dy0), fabsf(dy1)), f = fabsf(factor);
    float es = u * (fabsf(acy) * mx + fabsf(acx) * my) * f;
    float et = u * (fabsf(abx) * my + fabsf(aby) * mx) * f;
    float smax = fmaxf(fmaxf(s00, s10), fmaxf(s01, s11)), tmax = fmaxf(fmaxf(t00, t10), fmaxf(t01, t11));
    float w00 = 1 - s00 - t00, w10 = 1 - s10 - t10, w01 = 1 - s01 - t01, w11 = 1 - s11 - t11;
    float wmax = fmaxf(fmaxf(w00, w10), fmaxf(w01, w11));
    float sabs = fmaxf(fmaxf(fabsf(s00), fabsf(s10)), fmaxf(fabsf(s01), fabsf(s11)));
    float tabs = fmaxf(fmaxf(fabsf(t00), fabsf(t10)), fmaxf(fabsf(t01), fabsf(t11)));
    float ew = es + et + u * (1 + sabs + tabs);
    /* all of smax, es, ... finite is required for a rejection */
    bool finite = (smax - smax == 0) && (tmax - tmax == 0) && (wmax - wmax == 0) && (es - es == 0) && (et - et == 0) && (ew - ew == 0);
    return finite && (smax + es < -RZ_EPS || tmax + et < -RZ_EPS || wmax + ew < -RZ_EPS);
}


/* The ordered rasteriser, for a pass (or the part of a pass) from its first blended draw on: coverage, depth
 * and shading in one kernel, with the two-level stack above.  FAMILY >= 0 when every draw of every pass of the
 * batch is of that one family (only its shader is compiled in, which halves the registers of a blinn-only
 * kernel), NARROW when all their images are 4-byte formats.
 *
 * The read-only tables come in as separate const __restrict__ arguments so
 * that wave-uniform reads of them are issued as scalar loads.  Per 64 list
 * entries the candidates' records are fetched lane-parallel (one memory round
 * trip for all of them) and then broadcast one at a time in list order. */
template <int FAMILY, bool NARROW>
RZ_DEV void raster_tile(const PassParams &P, const uint32_t pass_index, const uint32_t tile_x, const uint32_t tile_y,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const BinEntry *__restrict__ list_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base) {
    if ((int)tile_x >= P.tiles_x || (int)tile_y >= P.tiles_y) return;
    if (P.control[2] != 0u) return;                             /* out of scratch, or left out: the pass is run again later */
    /* this pass's slices of the batch-wide tables */
    const TriRec *tri_table = tri_base + P.tri_off;
    const TriRec *fan_table = fan_base + (size_t)P.fan_off * RZ_MAX_FAN;
    const BinEntry *list_table = list_base + P.list_off;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const uint2 *bin_range_table = bin_range_base + P.bin_off;
    /* A workgroup owns a 32x8-pixel tile, wave w its w-th 8x8 block, lane = pixel. */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx0 = (int)tile_x * RZ_TILE_W, ty0 = (int)tile_y * RZ_TILE_H;
    const int bx0 = tx0 + wave * 8, by0 = ty0;
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool valid = px < P.width && py < P.height;
    const uint32_t pixel = (uint32_t)py * (uint32_t)P.width + (uint32_t)px;
    const int bin = (ty0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN;
    const uint2 range = bin_range_table[bin];          /* {first entry, entries} */
    const uint32_t count = range.y;
    if (count == 0 && !P.clear_color_pending && !P.clear_depth_pending) return;     /* uniform over the workgroup */

    float zcur = P.clear_depth;
    uint32_t ccur = P.clear_color;
    if (valid) {
        if (!P.clear_depth_pending) zcur = P.depth[pixel];
        if (!P.clear_color_pending) ccur = P.color[pixel];
    }
    const BinEntry *list = list_table + range.x;
    const int bx1 = bx0 + 7, by1 = by0 + 7;

    /* per-lane stack of unshaded fragments (see above) */
    uint32_t ref0 = RZ_REF_NONE, ref1 = RZ_REF_NONE;
    int depth_of_stack = 0;

    /* iterator over this block's triangles, in list order.  Every lane holds one WHOLE entry of the current group
     * of 64 (five 16-byte loads of consecutive addresses: one memory round trip per group); the entries that survive
     * the block test are then taken one at a time, their record passed round from the lane that holds it with
     * readlane.  Fetched through scalar loads when their turn came, every survivor cost a memory round trip of its
     * own -- in a bin where many do survive (layers of hair over each other) that was what the kernel's time went to. */
    uint32_t base = 0;
    unsigned long long todo = 0;
    float4 c1 = make_float4(0, 0, 0, 0), c2 = c1, c3 = c1, c4 = c1;
    uint32_t cref = 0;
    bool have = false, exhausted = false;
    TriRec r;
    uint32_t ref = 0;
    /* block corners at pixel centres, for the conservative triangle/block test */
    const float cx0 = (float)bx0 + 0.5f, cx1 = (float)bx0 + 7.5f, cy0 = (float)by0 + 0.5f, cy1 = (float)by0 + 7.5f;

    while (true) {
        /* ---- phase A: coverage + depth, until the list ends or a stack is full */
        bool stack_full = false;
        while (true) {
            if (!have) {
                if (todo) {
                    const int L = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    ref = (uint32_t)RZ_BCAST_I(cref, L);
                    r.ax = RZ_BCAST_F(c1.x, L); r.ay = RZ_BCAST_F(c1.y, L); r.abx = RZ_BCAST_F(c1.z, L); r.aby = RZ_BCAST_F(c1.w, L);
                    r.acx = RZ_BCAST_F(c2.x, L); r.acy = RZ_BCAST_F(c2.y, L); r.factor = RZ_BCAST_F(c2.z, L); r.z[0] = RZ_BCAST_F(c2.w, L);
                    r.z[1] = RZ_BCAST_F(c3.x, L); r.z[2] = RZ_BCAST_F(c3.y, L); r.rw[0] = RZ_BCAST_F(c3.z, L); r.rw[1] = RZ_BCAST_F(c3.w, L);
                    r.rw[2] = RZ_BCAST_F(c4.x, L); r.meta = (uint32_t)RZ_BCAST_I(__float_as_uint(c4.y), L);
                    r.bx = (uint32_t)RZ_BCAST_I(__float_as_uint(c4.z), L); r.by = (uint32_t)RZ_BCAST_I(__float_as_uint(c4.w), L);
                    have = true;
                } else {
                    if (base >= count) { exhausted = true; break; }
                    const bool present = base + lane < count;
                    if (present) {
                        const float4 *e = (const float4*)&list[base + lane];          /* reference + record, 80 bytes */
                        cref = __float_as_uint(e[0].x);
                        c1 = e[1]; c2 = e[2]; c3 = e[3]; c4 = e[4];
                    }
                    base += 64u;
                    bool hit = false;
                    if (present) {
                        hit = box_overlaps(make_uint2(__float_as_uint(c4.z), __float_as_uint(c4.w)), bx0, by0, bx1, by1);
                        if (hit) hit = !block_outside(c1, c2, cx0, cx1, cy0, cy1);
                    }
                    todo = __ballot(hit);
                    continue;
                }
            }
            /* the triangle in r against this block */
            int x0 = (int)(r.bx & 0xffffu), x1 = (int)(r.bx >> 16);
            int y0 = (int)(r.by & 0xffffu), y1 = (int)(r.by >> 16);
            float w0, w1, w2;
            pixel_weights(r, px, py, w0, w1, w2);
            bool inside = valid && px >= x0 && px <= x1 && py >= y0 && py <= y1 &&
                          w0 > -RZ_EPS && w1 > -RZ_EPS && w2 > -RZ_EPS;                  /* :516-519 */
            float depth = r.z[0] * w0 + r.z[1] * w1 + r.z[2] * w2;                        /* :403-408 */
            bool pass = inside && depth <= zcur;                                          /* :523 */
            if (!__any(pass)) { have = false; continue; }
            const bool blend = RZ_META_BLEND(r.meta) != 0;
            if (blend && __any(pass && depth_of_stack == 2)) { stack_full = true; break; }
            if (P.counters) {       /* fragments the reference would shade: [2] all, [6] skybox, [7] blended */
                unsigned long long n = (unsigned long long)__popcll(__ballot(pass));
                if (lane == 0) {
                    atomicAdd(P.counters + 2, n);
                    if (RZ_META_FAMILY(r.meta) == 2) atomicAdd(P.counters + 6, n);
                    if (blend) atomicAdd(P.counters + 7, n);
                }
            }
            if (RZ_META_CUTOFF(r.meta)) {
                /* the shaders' discard test (blinn_shader.c:104-113, 220; pbr_shader.c:116-134, 452):
                 * alpha = base alpha * first map's alpha, from the interpolated texcoord */
                const DrawRec &D = draw_table[RZ_META_DRAW(r.meta)];
                const int tc = RZ_META_FAMILY(r.meta) == 0 ? RZ_BV_TEXCOORD : RZ_PV_TEXCOORD;   /* (from the record: the varyings do not wait for the draw) */
                const float *v0, *v1, *v2;
                ref_varyings(ref, vary_table, fan_vary_table, v0, v1, v2);
                float q0 = r.rw[0] * w0, q1 = r.rw[1] * w1, q2 = r.rw[2] * w2;
                float norm = 1 / (q0 + q1 + q2);
                float tu = (v0[tc] * q0 + v1[tc] * q1 + v2[tc] * q2) * norm;
                float tv = (v0[tc + 1] * q0 + v1[tc + 1] * q1 + v2[tc + 1] * q2) * norm;
                if (pass) {
                    float a = D.base[3];
                    if (D.maps[0].data) a *= sample_repeat(D.maps[0], tu, tv).w;
                    if (a < D.alpha_cutoff) pass = false;
                }
            }
            if (pass) {
                zcur = depth;
                if (blend && depth_of_stack == 1) { ref1 = ref; depth_of_stack = 2; }
                else if (blend && depth_of_stack == 0) { ref0 = ref; depth_of_stack = 1; }
                else if (!blend) { ref0 = ref; depth_of_stack = 1; }
            }
            have = false;
        }
        /* ---- phase B: resolve the stacks, bottom level first */
        for (int level = 0; level < 2; level++) {
            bool want = depth_of_stack > level;
            if (!__any(want)) break;
            TriRec m;
            const float *m0 = vary_table, *m1 = vary_table, *m2 = vary_table;
            if (want) fetch_ref(level == 0 ? ref0 : ref1, tri_table, fan_table, vary_table, fan_vary_table, m, m0, m1, m2);
            shade_lanes<FAMILY, NARROW>(P, draw_table, want, m, m0, m1, m2, px, py, ccur);
        }
        ref0 = ref1 = RZ_REF_NONE;
        depth_of_stack = 0;
        if (exhausted || !stack_full) break;
    }

    if (valid) {
        P.depth[pixel] = zcur;
        P.color[pixel] = ccur;
    }
}

/* The order in which the workgroups of rz_strip_kernel start (that of their index) over the strips of a
 * batch of passes.  The pass is the fastest part of the index, and columns and rows count outwards from
 * the middle of the frame: what takes long -- the strips a model lies in, usually in the middle -- starts
 * first in every pass of the batch, and what is still running when the launch ends are the cheap strips
 * at the frames' borders.  (With the pass as the slowest part the last frame's long strips were the
 * launch's tail: 12.8 -> 11.0 us per frame for the helmet at 1080p, 26.4 -> 19.6 for
 * azura at 800x600, 80.7 -> 62.2 for dieselpunk at 4K.)  The unit of that order is the bin, not
 * the strip: the strips of a bin walk the same list and start next to each other.
 * grid: x = pass, y = column (whole bins: an even number of 32-pixel columns), z = row (rows_per_bin per bin row)
 * -- three dimensions, so that no index has to be divided by a number of passes only known at run time: a
 * wave that finds nothing to do in its bin must get there in a handful of scalar instructions. */
struct TileOrder { uint32_t pass, column, row; };
RZ_DEV uint32_t outwards(uint32_t turn, uint32_t count) {      /* 0, 1, 2, .. -> middle, middle + 1, middle - 1, .. */
    return (turn & 1u) ? (count - 1u) / 2u + (turn + 1u) / 2u : (count - 1u) / 2u - (turn + 1u) / 2u;
}
RZ_DEV TileOrder tile_order(uint32_t rows_per_bin) {
    TileOrder o;
    o.pass = blockIdx.x;
    o.column = outwards(blockIdx.y / 2u, gridDim.y / 2u) * 2u + (blockIdx.y & 1u);
    o.row = outwards(blockIdx.z / rows_per_bin, gridDim.z / rows_per_bin) * rows_per_bin + blockIdx.z % rows_per_bin;
    return o;
}

/* over the batch's list of tiles that need ordered shading (RZ_ORDERED_LIST: every tile of the passes' bins that
 * are not empty, listed by rz_bin_scan_kernel, or -- large framebuffers -- the tiles rz_blend_visibility_kernel
 * found blended fragments piling up in): a fixed number of workgroups strides over it.  Most of a frame's bins
 * hold nothing of a blended pass; a wave per tile of the whole frame spent the launch on starting waves that
 * read their pass's parameters and left (a million of them for 32 frames at 1080p). */
template <int FAMILY, bool NARROW>
__global__ void __launch_bounds__(256) rz_raster_list_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const BinEntry *__restrict__ list_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ queue_base, const uint32_t *__restrict__ queue_counts, uint32_t queue_capacity) {
    const int lane = threadIdx.x & 63;
    const uint32_t my_count = min(queue_counts[(RZ_ORDERED_LIST * RZ_QUEUE_WAYS + lane) * RZ_QUEUE_PAD], queue_capacity);
    uint32_t longest = my_count;
    longest = (uint32_t)wave_max_i((int)longest);
    for (uint32_t k = blockIdx.x; k < longest * RZ_QUEUE_WAYS; k += gridDim.x) {
        const uint32_t part = k & (RZ_QUEUE_WAYS - 1u), index = k / RZ_QUEUE_WAYS;
        if (index >= (uint32_t)__builtin_amdgcn_readlane((int)my_count, (int)part)) continue;
        const uint32_t item = queue_base[(size_t)(RZ_ORDERED_LIST * RZ_QUEUE_WAYS + part) * queue_capacity + index];
        const PassParams P = passes[item >> 20];
        raster_tile<FAMILY, NARROW>(P, item >> 20, item & 511u, (item >> 9) & 2047u, tri_base, fan_base, list_base, vary_base,
                                    fan_vary_base, upload_base, bin_range_base);
    }
}

/* K4b  visibility first, for a pass that has blended draws in a large framebuffer: a wave owns a block of
 * 16x8 pixels (lane = the pixel at (lane & 7, lane >> 3) and the one 8 to its right), a workgroup of two waves
 * a 32x8 tile.  A blended fragment over nothing else of this pass blends with the colour the framebuffer holds,
 * which the shade kernels can do; one over another pending fragment of the pass needs them shaded in order:
 * such a tile is left untouched and handed to rz_raster_list_kernel.  Everywhere else this leaves depth, the
 * tagged reference of the visible fragment and the tile on its family's work list. */
RZ_DEV void blend_visibility_tile(const PassParams P, const uint32_t pass_index, const uint32_t tile_x, const uint32_t tile_y,
        const BinEntry *__restrict__ list_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base) {
    constexpr int PPL = 2;
    if ((int)tile_x >= P.tiles_x || (int)tile_y >= P.tiles_y) return;
    if (P.control[2] != 0u) return;                             /* out of scratch, or left out: the pass is run again later */
    const BinEntry *list_table = list_base + P.list_off;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx0 = (int)tile_x * RZ_TILE_W, ty0 = (int)tile_y * RZ_TILE_H;
    const int bx0 = tx0 + wave * 8 * PPL, by0 = ty0, bx1 = bx0 + 8 * PPL - 1, by1 = by0 + 7;
    const int py = by0 + (lane >> 3);
    int px[PPL];
    bool valid[PPL];
    uint32_t pixel[PPL];
    for (int q = 0; q < PPL; q++) {
        px[q] = bx0 + 8 * q + (lane & 7);
        valid[q] = px[q] < P.width && py < P.height;
        pixel[q] = (uint32_t)py * (uint32_t)P.width + (uint32_t)px[q];
    }
    const int bin = (ty0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN;
    const uint2 range = (bin_range_base + P.bin_off)[bin];          /* {first entry, entries} */
    const uint32_t count = range.y;
    if (count == 0 && !P.clear_color_pending && !P.clear_depth_pending) return;     /* uniform over the workgroup; nobody looks at the references of a tile that is on no work list */
    if (count == 0) {               /* nothing but the pending clears (uniform over the workgroup) */
        for (int q = 0; q < PPL; q++) {
            if (!valid[q]) continue;
            if (P.clear_depth_pending) P.depth[pixel[q]] = P.clear_depth;
            if (P.clear_color_pending) P.color[pixel[q]] = P.clear_color;
        }
        return;
    }
    float zcur[PPL];
    uint32_t ref_of[PPL];                                           /* tagged with the family; NONE = no fragment */
    bool pending[PPL];                                              /* a fragment of this pass is the pixel's visible one so far */
    for (int q = 0; q < PPL; q++) {
        ref_of[q] = RZ_REF_NONE;
        pending[q] = false;
        zcur[q] = P.clear_depth;
        if (valid[q] && !P.clear_depth_pending) zcur[q] = P.depth[pixel[q]];
    }
    const BinEntry *list = list_table + range.x;
    __shared__ int s_ordered;
    bool needs_order = false;
    __syncthreads();                                            /* the workgroup's tile before this one has been settled */
    if (threadIdx.x == 0) s_ordered = 0;
    __syncthreads();

    /* this lane's candidate of the NEXT group of 64 entries (box and edge data), loaded one group ahead */
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0;
    uint2 nbox = make_uint2(RZ_BBOX_EMPTY_X, 0u);
    if ((uint32_t)lane < count) {
        const float4 *e = (const float4*)&list[lane];              /* reference + record, 80 bytes */
        n0 = e[1]; n1 = e[2];
        nbox = entry_box(e[4]);
    }
    /* block corners at pixel centres, for the conservative triangle/block test */
    const float cx0 = (float)bx0 + 0.5f, cx1 = (float)bx1 + 0.5f, cy0 = (float)by0 + 0.5f, cy1 = (float)by0 + 7.5f;

    for (uint32_t base = 0; base < count;) {
        const uint2 cbox = nbox;
        const float4 c0 = n0, c1 = n1;
        const bool present = base + lane < count;
        const BinEntry *group = list + base;
        base += 64u;
        if (base + lane < count) {
            const float4 *e = (const float4*)&list[base + lane];
            n0 = e[1]; n1 = e[2];
            nbox = entry_box(e[4]);
        }
        bool hit = false;
        if (present) {
            hit = box_overlaps(make_uint2(cbox.x, cbox.y & 0x7fffffffu), bx0, by0, bx1, by1);
            if (hit) hit = !block_outside(c0, c1, cx0, cx1, cy0, cy1);
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            /* the wave-uniform entry index makes these scalar loads */
            const uint32_t at = (uint32_t)__ffsll((long long)todo) - 1u;
            const BinEntry *e = &group[at];
            todo &= todo - 1;
            const uint32_t ref = e->tri;
            const TriRec r = e->rec;
            const int x0 = (int)(r.bx & 0xffffu), x1 = (int)(r.bx >> 16);
            const int y0 = (int)(r.by & 0xffffu), y1 = (int)(r.by >> 16);
            const bool row_in = py >= y0 && py <= y1;
            float w[PPL][3], depth[PPL];
            bool pass[PPL];
            for (int q = 0; q < PPL; q++) {
                pixel_weights(r, px[q], py, w[q][0], w[q][1], w[q][2]);
                const bool inside = valid[q] && row_in && px[q] >= x0 && px[q] <= x1 &&
                                    w[q][0] > -RZ_EPS && w[q][1] > -RZ_EPS && w[q][2] > -RZ_EPS;     /* :516-519 */
                depth[q] = r.z[0] * w[q][0] + r.z[1] * w[q][1] + r.z[2] * w[q][2];              /* :403-408 */
                pass[q] = inside && depth[q] <= zcur[q];                                          /* :523 */
            }
            bool any_pass = false;
            for (int q = 0; q < PPL; q++) any_pass |= pass[q];
            if (!__any(any_pass)) continue;
            if (RZ_META_CUTOFF(r.meta)) {
                /* the shaders' discard test (blinn_shader.c:104-113, 220; pbr_shader.c:116-134, 452):
                 * alpha = base alpha * first map's alpha, from the interpolated texcoord */
                const DrawRec &D = draw_table[RZ_META_DRAW(r.meta)];
                const int tc = RZ_META_FAMILY(r.meta) == 0 ? RZ_BV_TEXCOORD : RZ_PV_TEXCOORD;   /* (from the record: the varyings do not wait for the draw) */
                const float *v0, *v1, *v2;
                ref_varyings(ref, vary_table, fan_vary_table, v0, v1, v2);
                for (int q = 0; q < PPL; q++) {
                    float q0 = r.rw[0] * w[q][0], q1 = r.rw[1] * w[q][1], q2 = r.rw[2] * w[q][2];
                    float norm = 1 / (q0 + q1 + q2);
                    float tu = (v0[tc] * q0 + v1[tc] * q1 + v2[tc] * q2) * norm;
                    float tv = (v0[tc + 1] * q0 + v1[tc + 1] * q1 + v2[tc + 1] * q2) * norm;
                    if (pass[q]) {
                        float a = D.base[3];
                        if (D.maps[0].data) a *= sample_repeat(D.maps[0], tu, tv).w;
                        if (a < D.alpha_cutoff) pass[q] = false;
                    }
                }
            }
            const uint32_t tagged = ref | (RZ_META_FAMILY(r.meta) << 28);
            if (RZ_META_BLEND(r.meta))
                for (int q = 0; q < PPL; q++) needs_order |= pass[q] && pending[q];
            for (int q = 0; q < PPL; q++) if (pass[q]) { zcur[q] = depth[q]; ref_of[q] = tagged; pending[q] = true; }
        }
    }

    if (__any(needs_order) && lane == 0) s_ordered = 1;
    __syncthreads();
    if (s_ordered) {
        for (int q = 0; q < PPL; q++) if (valid[q]) P.ref[pixel[q]] = RZ_REF_NONE;     /* nothing here for the shade kernels */
        if (threadIdx.x == 0) {
            const uint32_t tile = tile_y * (uint32_t)P.tiles_x + tile_x;
            const uint32_t part = RZ_ORDERED_LIST * RZ_QUEUE_WAYS + ((P.tile_off + tile) & (RZ_QUEUE_WAYS - 1u));
            const uint32_t at = atomicAdd(&P.queue_counts[part * RZ_QUEUE_PAD], 1u);
            if (at < P.queue_capacity)
                P.queue[(size_t)part * P.queue_capacity + at] = (pass_index << 20) | (tile_y << 9) | tile_x;
            else
                atomicOr(&P.control[2], (uint32_t)RZ_ERR_QUEUE_OVERFLOW);
        }
        return;
    }
    for (int q = 0; q < PPL; q++) {
        if (!valid[q]) continue;
        if (ref_of[q] != RZ_REF_NONE || P.clear_depth_pending) P.depth[pixel[q]] = zcur[q];     /* else: what was there */
        P.ref[pixel[q]] = ref_of[q];
        /* the shade kernels write the colour of every pixel that has a fragment */
        if (P.clear_color_pending && ref_of[q] == RZ_REF_NONE) P.color[pixel[q]] = P.clear_color;
    }
    /* every tile that holds fragments goes on their family's work list (all passes of the batch
     * together), appended by the first wave of the tile to find one (tile_listed keeps who was first).  A list is
     * kept in RZ_QUEUE_WAYS parts, chosen by the tile's index, with counters a cache line apart, so that the appends
     * of a whole batch do not pile up on one counter. */
    for (uint32_t f = 0; f < 3; f++) {
        bool mine = false;
        for (int q = 0; q < PPL; q++) mine |= ((ref_of[q] >> 28) & 3u) == f;
        if (!__any(mine)) continue;
        if (lane == 0) {
            const uint32_t tile = tile_y * (uint32_t)P.tiles_x + tile_x;
            const uint32_t before = atomicOr(&P.tile_listed[tile], 1u << f);
            if (!(before >> f & 1u)) {
                /* by the tile's index in the batch: no part can get more than its even share */
                const uint32_t part = (f == 2u ? RZ_SKY_LIST : f) * RZ_QUEUE_WAYS + ((P.tile_off + tile) & (RZ_QUEUE_WAYS - 1u));
                const uint32_t at = atomicAdd(&P.queue_counts[part * RZ_QUEUE_PAD], 1u);
                if (at < P.queue_capacity)
                    P.queue[(size_t)part * P.queue_capacity + at] = (pass_index << 20) | (tile_y << 9) | tile_x;
                else
                    atomicOr(&P.control[2], (uint32_t)RZ_ERR_QUEUE_OVERFLOW);
            }
        }
    }
}

/* ... over the batch's list of the tiles of bins that are not empty */
__global__ void __launch_bounds__(128) rz_blend_visibility_kernel(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ queue_base, const uint32_t *__restrict__ queue_counts, uint32_t queue_capacity) {
    /* lane q holds the length of part q of the list of tiles to look at (RZ_VISIBILITY_LIST, rz_bin_scan_kernel) */
    const uint32_t my_count = min(queue_counts[(RZ_VISIBILITY_LIST * RZ_QUEUE_WAYS + (threadIdx.x & 63u)) * RZ_QUEUE_PAD], queue_capacity);
    const uint32_t longest = (uint32_t)wave_max_i((int)my_count);
    for (uint32_t k = blockIdx.x; k < longest * RZ_QUEUE_WAYS; k += gridDim.x) {
        const uint32_t part = k & (RZ_QUEUE_WAYS - 1u), index = k / RZ_QUEUE_WAYS;
        if (index >= (uint32_t)__builtin_amdgcn_readlane((int)my_count, (int)part)) continue;
        const uint32_t item = queue_base[(size_t)(RZ_VISIBILITY_LIST * RZ_QUEUE_WAYS + part) * queue_capacity + index];
        blend_visibility_tile(passes[item >> 20], item >> 20, item & 511u, (item >> 9) & 2047u, list_base, vary_base, fan_vary_base,
                              upload_base, bin_range_base);
    }
}

/* The sky box.  A frame's sky is twelve triangles that cover every pixel the models leave free: as list entries
 * they would sit in every bin of the frame (written there by the binning kernels, walked there by every strip).  A
 * sky box that is drawn last in its pass -- the frame driver's order: opaque models, sky box, then the blended
 * models, which start a pass of their own (test_helper.c:480-506) -- is kept out of the bins instead: the vertex
 * and clip kernels put its triangles (clipped: their fans) on the pass's TABLE of sky triangles, in submission
 * order, and rz_sky_kernel resolves them per pixel against what the strips left in the depth buffer.  Drawn last
 * means: a sky fragment wins iff its depth <= the depth the models left (graphics.c:523), and among sky fragments
 * the later one wins on equal depth -- the table's order.  Depth, colour and weights make no round trip between
 * two kernels.  (A skybox program that draws more than 12 triangles, or not last, or while fragments are counted,
 * goes through the bins like any other draw, and its fragments through rz_shade_list_kernel.)
 *
 * rz_bin_class_kernel, one wave per bin, writes a SkyBin record for every bin of such a pass: the (up to
 * RZ_SKY_BIN_MAX) table entries whose box meets the bin, and for each of the 32 REGIONS of 32x4 pixels a wave of
 * rz_sky_kernel takes the set of those that can touch it (box and conservative edge test, as rz_strip_kernel culls
 * its blocks, lane = region).  The two triangles of a sky box face share their box: testing both in every pixel of
 * the face cost the sky kernel a third of its time again.  A wave learns all of this in one memory round trip. */
#define RZ_SKY_BIN_MAX 8u
#define RZ_SKY_BIN_OVERFLOW 0xffffffffu
struct SkyBin {
    uint8_t entries_of_region[32];  /* bit e: entry e can touch the region */
    uint32_t ref[RZ_SKY_BIN_MAX];   /* the entries, in table order: what the table holds for them (a triangle's index or its fan record's) */
    uint32_t count;                 /* entries; RZ_SKY_BIN_OVERFLOW: more than fit, the whole table is looked at */
    uint32_t fresh;                 /* the bin's list is empty: nothing has consumed the pass's pending clears for its pixels */
    uint32_t pad[14];
};
static_assert(sizeof(SkyBin) == 128, "one record = one memory line");
RZ_DEV uint32_t sky_region_of(int tx0, int ry0) { return (uint32_t)(((ry0 % RZ_BIN) / 4) * 2 + (tx0 % RZ_BIN) / RZ_TILE_W); }
/* the record a table entry (or any fragment reference) stands for */
RZ_DEV const TriRec *ref_record(uint32_t ref, const TriRec *tri_table, const TriRec *fan_table) {
    return (ref & RZ_REF_FAN) ? &fan_table[ref & ~RZ_REF_FAN] : &tri_table[ref];
}

/* What rz_strip_kernel has to do with a bin, worked out once per bin (one wave each; grid x = bin, y = pass of the
 * batch) instead of by every one of its strips: RZ_BIN_STRIP = walk its list, RZ_BIN_DONE = nothing, the bin is
 * EMPTY.  An empty bin is finished right here, unless the pass has a table of sky triangles (then rz_sky_kernel
 * writes every pixel of it anyway): pending clears are written a row of four pixels per lane, 256 contiguous bytes
 * per bin row and plane -- a frame without a sky box is mostly such bins, and as 32-byte pieces out of strips whose
 * waves had each read their pass's parameters and the bin's range first, their clears ran at a quarter of the
 * memory's rate.  A strip whose batch has passes of one size finds its bin's class with two kernel arguments and
 * its own index, before it has read anything else.  For a pass with a table of sky triangles the bin's SkyBin
 * record is written here too. */
enum { RZ_BIN_STRIP = 0, RZ_BIN_DONE = 1 };
__global__ void __launch_bounds__(64) rz_bin_class_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const uint2 *__restrict__ bin_range_base, SkyBin *__restrict__ sky_base, uint32_t *__restrict__ class_base) {
    const PassParams &P = passes[blockIdx.y];
    if ((int)blockIdx.x >= P.bins_x * P.bins_y) return;
    const int lane = threadIdx.x & 63;
    uint32_t *bin_class = class_base + P.bin_off + blockIdx.x;
    if (P.control[2] != 0u) {                                   /* out of scratch, or left out: the pass is run again later */
        if (lane == 0) *bin_class = RZ_BIN_DONE;
        return;
    }
    const uint2 range = (bin_range_base + P.bin_off)[blockIdx.x];
    const int bin_x0 = (int)(blockIdx.x % (uint32_t)P.bins_x) * RZ_BIN, bin_y0 = (int)(blockIdx.x / (uint32_t)P.bins_x) * RZ_BIN;
    if (lane == 0) *bin_class = range.y == 0u ? (uint32_t)RZ_BIN_DONE : (uint32_t)RZ_BIN_STRIP;
    if (P.sky_draws == 0) {
        if (range.y != 0u || !(P.clear_color_pending || P.clear_depth_pending)) return;
        const float d = P.clear_depth;
        const uint32_t c = P.clear_color;
        if ((P.width & 3) == 0) {
            const int x = bin_x0 + (lane & 15) * 4;
            if (x < P.width)
                for (int row = lane >> 4; row < RZ_BIN; row += 4) {
                    const int py = bin_y0 + row;
                    if (py >= P.height) break;
                    const size_t pixel = (size_t)py * (size_t)P.width + (size_t)x;
                    if (P.clear_depth_pending) *(float4 RZ_GLOBAL*)(P.depth + pixel) = make_float4(d, d, d, d);
                    if (P.clear_color_pending) *(uint4 RZ_GLOBAL*)(P.color + pixel) = make_uint4(c, c, c, c);
                }
        } else {
            for (int i = lane; i < RZ_BIN * RZ_BIN; i += 64) {
                const int px = bin_x0 + (i & (RZ_BIN - 1)), py = bin_y0 + i / RZ_BIN;
                if (px >= P.width || py >= P.height) continue;
                const size_t pixel = (size_t)py * (size_t)P.width + (size_t)px;
                if (P.clear_depth_pending) P.depth[pixel] = d;
                if (P.clear_color_pending) P.color[pixel] = c;
            }
        }
        return;
    }
    /* ---- the bin's record for rz_sky_kernel: the table entries whose box meets the bin, in table order ... */
    const TriRec *tri_table = tri_base + P.tri_off;
    const TriRec *fan_table = fan_base + (size_t)P.fan_off * RZ_MAX_FAN;
    __shared__ uint32_t s_sel[RZ_SKY_BIN_MAX];
    uint32_t count = 0;                                         /* wave-uniform */
    const int table_entries = P.sky_draws * RZ_SKY_FACES * RZ_MAX_FAN;
    for (int first = 0; first < table_entries && count <= RZ_SKY_BIN_MAX; first += 64) {
        const int at = first + lane;
        const uint32_t ref = at < table_entries ? P.sky_table[at] : RZ_REF_NONE;
        bool meets = false;
        if (ref != RZ_REF_NONE) {
            const float4 q3 = ((const float4*)ref_record(ref, tri_table, fan_table))[3];
            meets = box_overlaps(make_uint2(__float_as_uint(q3.z), __float_as_uint(q3.w)), bin_x0, bin_y0, bin_x0 + RZ_BIN - 1, bin_y0 + RZ_BIN - 1);
        }
        const unsigned long long these = __ballot(meets);
        const uint32_t place = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(these >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)these, 0u));
        if (meets && place < RZ_SKY_BIN_MAX) s_sel[place] = ref;
        count += (uint32_t)__popcll(these);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool overflow = count > RZ_SKY_BIN_MAX;
    /* ... and which of them can touch which of the bin's 32 regions (lane = region) */
    const int rx0 = bin_x0 + RZ_TILE_W * (lane & 1), ry0 = bin_y0 + 4 * ((lane >> 1) & 15);
    uint32_t mine = 0;
    if (!overflow)
        for (uint32_t e = 0; e < count; e++) {
            const float4 *q = (const float4*)ref_record(s_sel[e], tri_table, fan_table);
            const float4 c0 = q[0], c1 = q[1], q3 = q[3];
            const bool hit = box_overlaps(make_uint2(__float_as_uint(q3.z), __float_as_uint(q3.w)), rx0, ry0, rx0 + RZ_TILE_W - 1, ry0 + 3) &&
                             !block_outside(c0, c1, (float)rx0 + 0.5f, (float)(rx0 + RZ_TILE_W - 1) + 0.5f, (float)ry0 + 0.5f, (float)ry0 + 3.5f);
            mine |= (hit ? 1u : 0u) << e;
        }
    SkyBin *rec = sky_base + P.bin_off + blockIdx.x;
    if (lane < 32) rec->entries_of_region[lane] = (uint8_t)mine;
    if (lane >= 32 && lane < 32 + (int)RZ_SKY_BIN_MAX) rec->ref[lane - 32] = (uint32_t)(lane - 32) < count ? s_sel[lane - 32] : RZ_REF_NONE;
    if (lane == 40) rec->count = overflow ? RZ_SKY_BIN_OVERFLOW : count;
    if (lane == 41) rec->fresh = range.y == 0u ? 1u : 0u;
}

/* the tag rz_strip_kernel puts on a fragment's reference: the family whose kernel shades it, or 3 for a fragment that
 * is black whatever its inputs (the strip's epilogue writes those itself; RZ_REF_NONE reads as 3 too) */
RZ_DEV uint32_t ref_tag(uint32_t meta) { return RZ_META_BLACK(meta) ? 3u : RZ_META_FAMILY(meta); }

/* What a wave of rz_strip_kernel does with a finished block of 16x8 pixels: depth, the clear colour where no
 * fragment is, black (draw_fragment keeps the buffer's alpha byte, graphics.c:466-468) where a shadow pass's
 * fragment is, and -- for blocks that hold fragments to shade -- the references. */
#define RZ_BLOCK_ITEM(pass, by, bx) (((uint32_t)(pass) << 21) | ((uint32_t)(by) << 10) | (uint32_t)(bx))
RZ_DEV void finish_block(const PassParams &P, uint32_t pass_index, int bx0, int by0, int lane,
                         const bool valid[2], const uint32_t pixel[2], const float zcur[2], const uint32_t ref_of[2]) {
    constexpr int PPL = 2;
    bool listed = false;
    for (int q = 0; q < PPL; q++) {
        if (valid[q] && (ref_of[q] != RZ_REF_NONE || P.clear_depth_pending)) P.depth[pixel[q]] = zcur[q];  /* else: what was there */
        listed |= ((ref_of[q] >> 28) & 3u) < 3u;                /* RZ_REF_NONE reads as family 3 */
        /* a pixel without a fragment shows the clear colour */
        if (valid[q] && P.clear_color_pending && ref_of[q] == RZ_REF_NONE) P.color[pixel[q]] = P.clear_color;
        if (valid[q] && ref_of[q] != RZ_REF_NONE && ((ref_of[q] >> 28) & 3u) == 3u)
            P.color[pixel[q]] = (P.clear_color_pending ? P.clear_color : P.color[pixel[q]]) & 0xff000000u;
    }
    if (__any(listed)) {
        /* the block holds fragments to shade: its references are written -- all of them, a
         * family's shade kernel looks at every pixel of a listed block -- and the block goes on the
         * work list of each family it holds (parts chosen by the block's index in its pass: a part
         * gets at most its even share of every pass, rounded up) */
        for (int q = 0; q < PPL; q++) if (valid[q]) P.ref[pixel[q]] = ref_of[q];
        const uint32_t block_x = (uint32_t)bx0 / (8u * PPL), block_y = (uint32_t)by0 / 8u;
        /* the block's index in its own pass: a block of the screen is on the same part, and with it on the same
         * XCD's share of the shade kernel's workgroups (workgroup b runs on XCD b mod 8 and takes parts 2b and
         * 2b + 1 mod 64), in every pass of the batch -- the views of a batch fetch much the same texels for it */
        const uint32_t index = block_y * 2u * (uint32_t)P.tiles_x + block_x;
        for (uint32_t f = 0; f < 3; f++) {
            bool mine_f = false;
            for (int q = 0; q < PPL; q++) mine_f |= ((ref_of[q] >> 28) & 3u) == f;
            if (!__any(mine_f)) continue;
            if (lane == 0) {
                const uint32_t part = (f == 2u ? RZ_SKY_LIST : f) * RZ_QUEUE_WAYS + (index & (RZ_QUEUE_WAYS - 1u));
                const uint32_t at = atomicAdd(&P.queue_counts[part * RZ_QUEUE_PAD], 1u);
                if (at < P.queue_capacity)
                    P.queue[(size_t)part * P.queue_capacity + at] = RZ_BLOCK_ITEM(pass_index, block_y, block_x);
                else
                    atomicOr(&P.control[2], (uint32_t)RZ_ERR_QUEUE_OVERFLOW);
            }
        }
    }
}

/* K4c  visibility for passes without blending: coverage + depth, leaves depth and one fragment reference per
 * pixel.  A wave owns blocks of 16x8 pixels (lane = the pixel at (lane & 7, lane >> 3) and the one 8 to its right:
 * half the waves walk the lists, cull candidates and fetch survivors compared with one pixel per lane), a STRIP of
 * them per wave.
 *
 * With one block per wave a wave spends its time waiting: it lives for some 14 000 cycles and issues 700
 * instructions, the rest is a chain of dependent memory round trips (the pass's parameters, the bin's
 * range, the list's first entries, the surviving entries, the stores), paid by every wave for 128
 * pixels.  Here a wave keeps its 16-pixel column and takes STRIP blocks of 16x8 pixels one above the
 * other inside one bin -- same list, same parameters -- so the chain is paid once per 128 * STRIP
 * pixels; a list of up to 64 entries (any bin that holds only sky) stays in the lanes' registers for
 * the whole strip.  The first group of the next block's candidates is requested before the current
 * block's epilogue, which hides its latency behind the epilogue's arithmetic.
 *
 * How tall a strip should be depends on the lists.  Where they are short the fixed costs of a wave and of
 * starting its workgroup count; with the sky box in the bins (every bin of the frame a list of a dozen entries)
 * tall strips won (helmet at 1080p, longest list 700 entries: 12.9 / 12.5 / 11.7 us per frame with 1 / 2 / 4
 * blocks).  With the sky box on its own table only the bins a model touches have strips at all, a scene of a few
 * thousand triangles has fewer waves than the device holds, and shorter strips finish sooner (helmet 6.4 / 6.6 /
 * 7.0 us; kgirl at 1080p, 32 passes: 43.0 / 43.9 / 40.6 thousand frames per second): two blocks is the default.  A long list makes
 * the blocks themselves expensive: then four times as many waves, each a quarter as long, fill the
 * device better and leave a shorter tail (azura at 800x600, a few bins of 10 000 entries and more: 19.2 /
 * 30.6 / 46.8 us).  The host picks STRIP per batch from the longest list its framebuffers had in their
 * last frame (control word 4, rz_bin_scan_kernel).  Deciding per bin inside one launch -- a workgroup
 * per block, all but the first of a strip leaving at once where the list is short -- gave the long lists
 * their waves but cost the short ones the workgroup starts again (helmet 11.9 -> 14.0 us).
 *
 * The epilogue writes depth and, for blocks that hold blinn or pbr fragments, the references; such a block is
 * appended to its family's work list (16x8 blocks, not 32x8 tiles: the wave that owns the block is the only one
 * to list it).  Skybox fragments are rz_sky_kernel's: it looks at the references of every pixel outside the bins
 * it resolves itself, so in a batch with a sky box the references of all blocks are written. */
#ifdef RZ_PROBE
#define RZ_PROBE_WORDS 12
#define RZ_PROBE_MAX (1 << 19)
__device__ unsigned long long rz_probe_data[RZ_PROBE_WORDS * RZ_PROBE_MAX];
__device__ unsigned int rz_probe_count;
#endif
template <int STRIP>
__global__ void __launch_bounds__(128) rz_strip_kernel(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base, const uint2 *__restrict__ box_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ class_base, int uniform_width, int uniform_height) {
    constexpr int PPL = 2;
    static_assert(RZ_BIN % (8 * STRIP) == 0, "a strip lies inside one bin");
    const TileOrder o = tile_order(RZ_BIN / (8 * STRIP));
    const uint32_t pass_index = o.pass, tile_column = o.column, strip_row = o.row;
    const int sy0 = (int)strip_row * 8 * STRIP, tx0 = (int)tile_column * RZ_TILE_W;
    /* the passes of the batch have one size (uniform_width != 0): what the bin's class says is known before the pass's
     * parameters are asked for, and most strips of most frames end here */
    if (uniform_width) {
        if (tx0 >= uniform_width || sy0 >= uniform_height) return;
        const uint32_t bins_x = ((uint32_t)uniform_width + RZ_BIN - 1u) / RZ_BIN, bins_y = ((uint32_t)uniform_height + RZ_BIN - 1u) / RZ_BIN;
        if (class_base[pass_index * (bins_x * bins_y) + (uint32_t)(sy0 / RZ_BIN) * bins_x + (uint32_t)(tx0 / RZ_BIN)] != (uint32_t)RZ_BIN_STRIP) return;
    }
    const PassParams P = passes[pass_index];
    if ((int)tile_column >= P.tiles_x || sy0 >= P.height) return;
    const BinEntry *list_table = list_base + P.list_off;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx0 = tx0 + wave * 8 * PPL, bx1 = bx0 + 8 * PPL - 1;
    int px[PPL];
    for (int q = 0; q < PPL; q++) px[q] = bx0 + 8 * q + (lane & 7);
    const int bin = (sy0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN;
    const uint2 range = (bin_range_base + P.bin_off)[bin];          /* {first entry, entries} */
    if (!uniform_width && class_base[P.bin_off + (uint32_t)bin] != (uint32_t)RZ_BIN_STRIP) return;
    const uint32_t count = range.y;                                 /* (a bin of this class is not empty) */
#ifdef RZ_PROBE
    const unsigned long long probe_t0 = wall_clock64();
    unsigned long long probe_ordered = 0, probe_ring = 0, probe_groups = 0, probe_hits = 0, probe_survivors = 0, probe_cutoffs = 0, probe_batches = 0, probe_small = 0;
#endif
    const BinEntry *list = list_table + range.x;
    const uint2 *boxes = box_base + P.list_off + range.x;

    /* Small triangles.  Which fragment a pixel ends up with does not depend on the order the
     * triangles are taken in: it is the one of least depth, the latest in submission order among
     * equals (graphics.c:523 keeps a fragment when depth <= the stored depth).  Candidates whose
     * box covers at most micro_area pixels of this wave's block are therefore set aside in a ring,
     * and each time 64 have come together every lane rasterises one of them, pixel by pixel,
     * into a per-pixel key in LDS: depth (as an order-preserving integer) above the complement of
     * the list position, so that an atomic minimum keeps exactly that fragment.  The keys meet the
     * registers of the one-triangle-per-step path when the list ends.  Triangles with an alpha
     * cutoff and passes that count fragments stay on the ordered path. */
    __shared__ unsigned long long micro_keys[256];
    __shared__ uint32_t micro_ring[2][128];
    unsigned long long *keys = micro_keys + wave * 64 * PPL;
    uint32_t *ring = micro_ring[wave];
    const int micro_area = (P.counters || count < (uint32_t)P.micro_min_list) ? 0 : P.micro_area;
    const int sx1 = min(bx1, P.width - 1);

    /* This lane's candidate of the NEXT group of 64 entries, loaded one group ahead.  A short list
     * is read from the entries themselves, box and edge data together (one round trip per group).
     * A long list is walked through the 8-byte boxes -- 512 contiguous bytes per load, where the
     * same fields out of the 80-byte entries are 64 cache lines and the loads, not the arithmetic,
     * set the pace -- and the edge data of the few candidates whose box meets the block is fetched
     * afterwards.  A list of at most 64 entries is loaded once and serves every block of the strip. */
    const bool long_list = count >= (uint32_t)P.micro_min_list;    /* wave-uniform */
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0;
    uint2 nbox = make_uint2(RZ_BBOX_EMPTY_X, 0u);
    auto request_first_group = [&]() {
        if ((uint32_t)lane < count) {
            if (long_list) {
                nbox = boxes[lane];
            } else {
                const float4 *e = (const float4*)&list[lane];          /* reference + record, 80 bytes */
                n0 = e[1]; n1 = e[2];
                nbox = entry_box(e[4]);
            }
        }
    };
    request_first_group();

    for (int k = 0; k < STRIP; k++) {
        const int by0 = sy0 + 8 * k, by1 = by0 + 7;
        if (by0 >= P.height) break;                                 /* uniform */
        const int py = by0 + (lane >> 3);
        const int sy1 = min(by1, P.height - 1);
        bool valid[PPL];
        uint32_t pixel[PPL];
        float zcur[PPL];
        uint32_t ref_of[PPL];                                       /* tagged with the family; NONE = no fragment */
        int ord_of[PPL];                                            /* its position in the bin's list; -1 = what the buffer held */
        for (int q = 0; q < PPL; q++) {
            valid[q] = px[q] < P.width && py < P.height;
            pixel[q] = (uint32_t)py * (uint32_t)P.width + (uint32_t)px[q];
            ref_of[q] = RZ_REF_NONE;
            ord_of[q] = -1;
            zcur[q] = P.clear_depth;
            if (valid[q] && !P.clear_depth_pending) zcur[q] = P.depth[pixel[q]];
        }
        uint32_t ring_head = 0, ring_tail = 0;                      /* wave-uniform */
        if (micro_area)
            for (int q = 0; q < PPL; q++) keys[q * 64 + lane] = ~0ull;
        auto rasterise_ring = [&](uint32_t n) {                     /* one triangle per lane, the ring's oldest n <= 64 */
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool on = (uint32_t)lane < n;
            const uint32_t ord = on ? ring[(ring_head + (uint32_t)lane) & 127u] : 0u;
            ring_head += n;
            TriRec r;
            int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
            if (on) {
                const float4 *e = (const float4*)&list[ord];
                const float4 a = e[1], b = e[2], c = e[3], d = e[4];
                r.ax = a.x; r.ay = a.y; r.abx = a.z; r.aby = a.w;
                r.acx = b.x; r.acy = b.y; r.factor = b.z; r.z[0] = b.w;
                r.z[1] = c.x; r.z[2] = c.y;
                const uint32_t bx = __float_as_uint(d.z), by = __float_as_uint(d.w);
                x0 = max((int)(bx & 0xffffu), bx0); x1 = min((int)(bx >> 16), sx1);
                y0 = max((int)(by & 0xffffu), by0); y1 = min((int)(by >> 16), sy1);
            }
            int x = x0, y = y0;
            const uint32_t low = ~ord;
            while (__any(y <= y1)) {
                if (y <= y1) {
                    float w0, w1, w2;
                    pixel_weights(r, x, y, w0, w1, w2);
                    const float depth = r.z[0] * w0 + r.z[1] * w1 + r.z[2] * w2;                       /* :403-408 */
                    if (w0 > -RZ_EPS && w1 > -RZ_EPS && w2 > -RZ_EPS && depth == depth) {             /* :516-519 */
                        const uint32_t bits = __float_as_uint(depth + 0.0f);
                        const uint32_t up = bits ^ ((uint32_t)((int32_t)bits >> 31) | 0x80000000u);
                        const int lx = x - bx0, ly = y - by0;
                        atomicMin(&keys[(lx >> 3) * 64 + (ly << 3) + (lx & 7)], ((unsigned long long)up << 32) | low);
                    }
                    if (++x > x1) { x = x0; y++; }
                }
            }
        };
        /* block corners at pixel centres, for the conservative triangle/block test */
        const float cx0 = (float)bx0 + 0.5f, cx1 = (float)bx1 + 0.5f, cy0 = (float)by0 + 0.5f, cy1 = (float)by0 + 7.5f;

        for (uint32_t base = 0; base < count;) {
            const uint2 cbox = nbox;
            float4 c0 = n0, c1 = n1;
            const bool present = base + lane < count;
            const BinEntry *group = list + base;
            const uint32_t group_at = base;
            base += 64u;
            if (base + lane < count) {
                if (long_list) {
                    nbox = boxes[base + lane];
                } else {
                    const float4 *e = (const float4*)&list[base + lane];
                    n0 = e[1]; n1 = e[2];
                    nbox = entry_box(e[4]);
                }
            }
            const bool cutoff = (cbox.y >> 31) != 0;
            bool hit = false;
            if (present) {
                hit = box_overlaps(make_uint2(cbox.x, cbox.y & 0x7fffffffu), bx0, by0, bx1, by1);
                if (hit && long_list) {
                    const float4 *e = (const float4*)&group[lane];
                    c0 = e[1]; c1 = e[2];
                }
                if (hit) hit = !block_outside(c0, c1, cx0, cx1, cy0, cy1);
            }
            if (micro_area) {
                bool small = false;
                if (hit) {
                    const uint32_t bx = cbox.x, by = cbox.y & 0x7fffffffu;
                    const int w = min((int)(bx >> 16), sx1) - max((int)(bx & 0xffffu), bx0) + 1;
                    const int h = min((int)(by >> 16), sy1) - max((int)(by & 0xffffu), by0) + 1;
                    small = w * h <= micro_area && !cutoff;
                }
                const unsigned long long smalls = __ballot(small);
                if (smalls) {
                    if (small) ring[(ring_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(smalls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)smalls, 0u))) & 127u] = group_at + (uint32_t)lane;
                    ring_tail += (uint32_t)__popcll(smalls);
#ifdef RZ_PROBE
                    probe_small += __popcll(smalls);
#endif
                    hit = hit && !small;
                    if (ring_tail - ring_head >= 64u) {
#ifdef RZ_PROBE
                        const unsigned long long t = clock64(); probe_batches++;
#endif
                        rasterise_ring(64u);
#ifdef RZ_PROBE
                        probe_ring += clock64() - t;
#endif
                    }
                }
            }
            unsigned long long todo = __ballot(hit);
#ifdef RZ_PROBE
            probe_groups++; probe_survivors += __popcll(todo); probe_cutoffs += __popcll(__ballot(hit && cutoff));
            const unsigned long long to = clock64();
#endif
            while (todo) {
                /* the wave-uniform entry index makes these scalar loads */
                const uint32_t at = (uint32_t)__ffsll((long long)todo) - 1u;
                const BinEntry *e = &group[at];
                todo &= todo - 1;
                const uint32_t ref = e->tri;
                const TriRec r = e->rec;
                const int x0 = (int)(r.bx & 0xffffu), x1 = (int)(r.bx >> 16);
                const int y0 = (int)(r.by & 0xffffu), y1 = (int)(r.by >> 16);
                const bool row_in = py >= y0 && py <= y1;
                float w[PPL][3], depth[PPL];
                bool pass[PPL];
                for (int q = 0; q < PPL; q++) {
                    pixel_weights(r, px[q], py, w[q][0], w[q][1], w[q][2]);
                    const bool inside = valid[q] && row_in && px[q] >= x0 && px[q] <= x1 &&
                                        w[q][0] > -RZ_EPS && w[q][1] > -RZ_EPS && w[q][2] > -RZ_EPS;     /* :516-519 */
                    depth[q] = r.z[0] * w[q][0] + r.z[1] * w[q][1] + r.z[2] * w[q][2];              /* :403-408 */
                    pass[q] = inside && depth[q] <= zcur[q];                                          /* :523 */
                }
                bool any_pass = false;
                for (int q = 0; q < PPL; q++) any_pass |= pass[q];
                if (!__any(any_pass)) continue;
                if (P.counters) {       /* fragments the reference would shade: [2] all, [6] skybox */
                    unsigned long long n = 0;
                    for (int q = 0; q < PPL; q++) n += (unsigned long long)__popcll(__ballot(pass[q]));
                    if (lane == 0) {
                        atomicAdd(P.counters + 2, n);
                        if (RZ_META_FAMILY(r.meta) == 2) atomicAdd(P.counters + 6, n);
                    }
                }
                if (RZ_META_CUTOFF(r.meta)) {
                    /* the shaders' discard test (blinn_shader.c:104-113, 220; pbr_shader.c:116-134, 452):
                     * alpha = base alpha * first map's alpha, from the interpolated texcoord */
                    const DrawRec &D = draw_table[RZ_META_DRAW(r.meta)];
                    const int tc = RZ_META_FAMILY(r.meta) == 0 ? RZ_BV_TEXCOORD : RZ_PV_TEXCOORD;   /* (from the record: the varyings do not wait for the draw) */
                    const float *v0, *v1, *v2;
                    ref_varyings(ref, vary_table, fan_vary_table, v0, v1, v2);
                    for (int q = 0; q < PPL; q++) {
                        float q0 = r.rw[0] * w[q][0], q1 = r.rw[1] * w[q][1], q2 = r.rw[2] * w[q][2];
                        float norm = 1 / (q0 + q1 + q2);
                        float tu = (v0[tc] * q0 + v1[tc] * q1 + v2[tc] * q2) * norm;
                        float tv = (v0[tc + 1] * q0 + v1[tc + 1] * q1 + v2[tc + 1] * q2) * norm;
                        if (pass[q]) {
                            float a = D.base[3];
                            if (D.maps[0].data) a *= sample_repeat(D.maps[0], tu, tv).w;
                            if (a < D.alpha_cutoff) pass[q] = false;
                        }
                    }
                }
                const uint32_t tagged = ref | (ref_tag(r.meta) << 28);
                for (int q = 0; q < PPL; q++) if (pass[q]) { zcur[q] = depth[q]; ref_of[q] = tagged; ord_of[q] = (int)(group_at + at); }
            }
#ifdef RZ_PROBE
            probe_ordered += clock64() - to;
#endif
        }
        if (micro_area) {
#ifdef RZ_PROBE
            const unsigned long long tr = clock64(); if (ring_tail != ring_head) probe_batches++;
#endif
            if (ring_tail != ring_head) rasterise_ring(ring_tail - ring_head);
#ifdef RZ_PROBE
            probe_ring += clock64() - tr;
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int q = 0; q < PPL; q++) {
                const unsigned long long key = keys[q * 64 + lane];
                if (key == ~0ull) continue;
                const uint32_t up = (uint32_t)(key >> 32);
                const float depth = __uint_as_float(up ^ ((up >> 31) ? 0x80000000u : 0xffffffffu));
                const int ord = (int)~(uint32_t)key;
                if (depth < zcur[q] || (depth == zcur[q] && ord > ord_of[q])) {
                    const BinEntry *e = &list[ord];
                    zcur[q] = depth;
                    ref_of[q] = e->tri | (ref_tag(e->rec.meta) << 28);
                    ord_of[q] = ord;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      /* the keys are set up again for the next block */
        }
        /* the next block's first candidates: on their way while this block's epilogue runs */
        if (count > 64u && k + 1 < STRIP) request_first_group();

        finish_block(P, pass_index, bx0, by0, lane, valid, pixel, zcur, ref_of);
    }
#ifdef RZ_PROBE
    if (lane == 0) {
        const unsigned int at = atomicAdd(&rz_probe_count, 1u);
        if (at < RZ_PROBE_MAX) {
            unsigned long long *o = rz_probe_data + (size_t)at * RZ_PROBE_WORDS;
            o[0] = pass_index; o[1] = (unsigned long long)bin; o[2] = count; o[3] = probe_groups; o[4] = probe_survivors; o[5] = probe_cutoffs;
            o[6] = probe_batches; o[7] = wall_clock64() - probe_t0; o[8] = probe_ordered; o[9] = probe_ring; o[10] = probe_small; o[11] = probe_t0;
        }
    }
#endif
}

/* K5  per-family fragment kernels behind rz_strip_kernel / rz_blend_visibility_kernel:
 * every lane shades the one fragment its pixel refers to. */
template <int FAMILY, bool NARROW = false, class TABLES = GlobalTables>
RZ_DEV void shade_pixel(const PassParams &P, const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
                        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
                        const char *__restrict__ upload_base, const uint32_t *__restrict__ ref_base,
                        int px, int py) {
    const TriRec *tri_table = tri_base + P.tri_off;
    const TriRec *fan_table = fan_base + (size_t)P.fan_off * RZ_MAX_FAN;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const uint32_t *ref_table = ref_base + P.ref_off;
    const bool valid = px < P.width && py < P.height;
    const uint32_t pixel = (uint32_t)py * (uint32_t)P.width + (uint32_t)px;
    uint32_t myref = valid ? ref_table[pixel] : RZ_REF_NONE;
    /* the rasteriser tagged the reference with the fragment's family (RZ_REF_NONE reads as family 3) */
    const bool want = ((myref >> 28) & 3u) == (uint32_t)FAMILY;
    if (!__any(want)) return;
    myref &= ~RZ_REF_FAMILY_BITS;
    /* the colour below only lends its alpha byte; under a pending clear it is the clear colour */
    uint32_t ccur = P.clear_color;
    if (want && !P.clear_color_pending) ccur = P.color[pixel];
    TriRec m;
    const float *m0 = vary_table, *m1 = vary_table, *m2 = vary_table;
    if (want) fetch_ref(myref, tri_table, fan_table, vary_table, fan_vary_table, m, m0, m1, m2);
    shade_lanes<FAMILY, NARROW, TABLES>(P, draw_table, want, m, m0, m1, m2, px, py, ccur);
    if (want) P.color[pixel] = ccur;
}

/* The sky boxes on the passes' tables, over every tile of every pass (grid x, y = tile, z = pass), two pixels
 * per lane: a wave takes a region of 32x4 pixels (lane = the pixel at (lane & 31, lane >> 5) of its rows and
 * the one two rows below), so the per-wave work -- the pass's parameters, the bin's record, a triangle's record
 * and varyings, the draw's face descriptors, all through scalar loads -- is paid once per 128 pixels.
 * Visibility among the table's triangles as in rz_strip_kernel (graphics.c:512-532), in table order, starting from
 * the depth the pass's other triangles left -- or, in a bin whose list is empty, from the pending clear, which
 * this kernel then carries out for the pixels no sky triangle covers. */
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    static_assert(PPL == 2, "a wave's region is 32x4 pixels");
    const PassParams P = passes[blockIdx.z];
    if ((int)blockIdx.x >= P.tiles_x || (int)blockIdx.y >= P.tiles_y || P.sky_draws == 0) return;
    const uint32_t failed = P.control[2];             /* looked at below, in the same round trip as the bin's record */
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const TriRec *tri_table = tri_base + P.tri_off;
    const TriRec *fan_table = fan_base + (size_t)P.fan_off * RZ_MAX_FAN;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const int tx0 = (int)blockIdx.x * RZ_TILE_W, ry0 = (int)blockIdx.y * RZ_TILE_H + wave * 2 * PPL;
    const int px = tx0 + (lane & 31);
    int py[PPL];
    uint32_t pixel[PPL];
    for (int q = 0; q < PPL; q++) {
        py[q] = ry0 + 2 * q + (lane >> 5);
        pixel[q] = (uint32_t)py[q] * (uint32_t)P.width + (uint32_t)px;
    }
    const uint32_t region = sky_region_of(tx0, ry0);
    const SkyBin &bin = sky_base[P.bin_off + (uint32_t)((ry0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN)];
    const uint32_t sky_count = bin.count, fresh = bin.fresh;
    const uint32_t sky_word = ((const uint32_t*)bin.entries_of_region)[region >> 2];
    uint32_t refs[RZ_SKY_BIN_MAX];
    for (uint32_t e = 0; e < RZ_SKY_BIN_MAX; e++) refs[e] = bin.ref[e];
    if (failed != 0u) return;                         /* the pass did not run */
    const bool overflow = sky_count == RZ_SKY_BIN_OVERFLOW;
    const uint32_t entries = overflow ? 0u : (sky_word >> ((region & 3u) * 8u)) & 0xffu;
    /* in a bin whose list is empty nothing has consumed the pass's clears; elsewhere rz_strip_kernel has */
    const bool clear_depth = fresh && P.clear_depth_pending, clear_color = fresh && P.clear_color_pending;
    if (!overflow && entries == 0u && !clear_depth && !clear_color) return;     /* uniform */
    bool valid[PPL];
    float zcur[PPL], w[PPL][3];
    uint32_t win[PPL];                                /* what this pixel's sky fragment comes from (a table entry); RZ_REF_NONE: none */
    for (int q = 0; q < PPL; q++) {
        valid[q] = px < P.width && py[q] < P.height;
        zcur[q] = P.clear_depth;
        if (valid[q] && !clear_depth) zcur[q] = P.depth[pixel[q]];
        win[q] = RZ_REF_NONE;
        w[q][0] = w[q][1] = w[q][2] = 0;
    }
    TriRec m;                                         /* the entry taken last */
    uint32_t m_ref = RZ_REF_NONE;
    auto consider = [&](uint32_t ref) {               /* uniform */
        m_ref = ref;
        m = load_tri(ref_record(ref, tri_table, fan_table));
        const int x0 = (int)(m.bx & 0xffffu), x1 = (int)(m.bx >> 16);
        const int y0 = (int)(m.by & 0xffffu), y1 = (int)(m.by >> 16);
        for (int q = 0; q < PPL; q++) {
            float w0, w1, w2;
            pixel_weights(m, px, py[q], w0, w1, w2);
            const bool inside = valid[q] && py[q] >= y0 && py[q] <= y1 && px >= x0 && px <= x1 &&
                                w0 > -RZ_EPS && w1 > -RZ_EPS && w2 > -RZ_EPS;                 /* :516-519 */
            const float depth = m.z[0] * w0 + m.z[1] * w1 + m.z[2] * w2;                  /* :403-408 */
            if (inside && depth <= zcur[q]) {                                             /* :523 */
                zcur[q] = depth; win[q] = ref;
                w[q][0] = w0; w[q][1] = w1; w[q][2] = w2;
            }
        }
    };
    if (!overflow) {
#pragma unroll
        for (uint32_t e = 0; e < RZ_SKY_BIN_MAX; e++) if (entries >> e & 1u) consider(refs[e]);
    } else {
        /* more table entries meet the bin than its record holds: all of the table, entry by entry */
        for (int at = 0; at < P.sky_draws * RZ_SKY_FACES * RZ_MAX_FAN; at++) {
            const uint32_t ref = P.sky_table[at];
            if (ref != RZ_REF_NONE) consider(ref);
        }
    }
    /* what the colour buffer lends is its alpha byte only (graphics.c:466-468); under a pending clear that is the
     * clear colour's, whoever carries the clear out */
    uint32_t below[PPL];
    for (int q = 0; q < PPL; q++) {
        below[q] = P.clear_color;
        if (valid[q] && !P.clear_color_pending) below[q] = P.color[pixel[q]];
    }
    bool every = true, any_want = false;
    for (int q = 0; q < PPL; q++) { every = every && win[q] != RZ_REF_NONE; any_want |= win[q] != RZ_REF_NONE; }
    if (!overflow && (entries & (entries - 1u)) == 0u && __all(every)) {
        /* The usual region: one entry can touch it and every pixel shows it.  Said once for the wave, what follows
         * needs no lane masks around its loads and stores (they were a quarter of the kernel's scalar instructions,
         * and the scalar unit was busy 73 % of the time). */
        const DrawRec &D = draw_table[RZ_META_DRAW(m.meta)];
        const float *m0, *m1, *m2;
        ref_varyings(m_ref, vary_table, fan_vary_table, m0, m1, m2);
        uint32_t at[PPL];
        float v[PPL][4];
        for (int q = 0; q < PPL; q++) {
            P.depth[pixel[q]] = zcur[q];
            if (!fresh) P.ref[pixel[q]] = RZ_REF_NONE;          /* a fragment of the pass's other triangles may have been here */
            const float q0 = m.rw[0] * w[q][0], q1 = m.rw[1] * w[q][1], q2 = m.rw[2] * w[q][2];    /* graphics.c:426-429 */
            const float norm = 1 / (q0 + q1 + q2);
            interpolate_uniform3(m0, m1, m2, q0, q1, q2, norm, v[q]);
        }
        if (D.flags & RZ_DRAW_SKY_BYTES) {      /* both texels asked for before either is waited for */
            const uint32_t RZ_GLOBAL *bytes = D.maps[0].data.as<const uint32_t>();
            for (int q = 0; q < PPL; q++) at[q] = sky_bytes_index(D, v[q]);
            uint32_t word[PPL];
            for (int q = 0; q < PPL; q++) word[q] = bytes[at[q]];
            for (int q = 0; q < PPL; q++) P.color[pixel[q]] = (word[q] & 0x00ffffffu) | (below[q] & 0xff000000u);
        } else {
            for (int q = 0; q < PPL; q++) P.color[pixel[q]] = sky_fragment(D, v[q], below[q]);
        }
        return;
    }
    bool want[PPL];
    for (int q = 0; q < PPL; q++) {
        want[q] = win[q] != RZ_REF_NONE;
        if (valid[q] && (want[q] || clear_depth)) P.depth[pixel[q]] = zcur[q];         /* else: what was there */
        if (valid[q] && clear_color && !want[q]) P.color[pixel[q]] = P.clear_color;
        if (want[q] && !fresh) P.ref[pixel[q]] = RZ_REF_NONE;   /* a fragment of the pass's other triangles may have been here */
    }
    if (!__any(any_want)) return;
    /* The lanes drew from one entry, or -- an edge or the diagonal of a sky box face crosses the region (one region in
     * five) -- from two or three.  Each lane takes the reciprocal w and the corner varyings of ITS entry, entry by entry
     * through scalar loads, kept by the lanes that drew from it (per-lane gathers of records and varyings made the
     * texture data path this kernel's bound); the entry's draw shades its lanes. */
    for (int q = 0; q < PPL; q++) {
        if (!__any(want[q])) continue;
        unsigned long long rest = __ballot(want[q]);
        while (rest) {
            const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)win[q], __ffsll((long long)rest) - 1);
            const bool mine = want[q] && win[q] == ref;
            rest &= ~__ballot(mine);
            const TriRec *rec = ref_record(ref, tri_table, fan_table);
            const float r0 = rec->rw[0], r1 = rec->rw[1], r2 = rec->rw[2];
            const DrawRec &D = draw_table[RZ_META_DRAW(rec->meta)];
            const float *e0, *e1, *e2;
            ref_varyings(ref, vary_table, fan_vary_table, e0, e1, e2);
            const rz_float4 a = *(const rz_float4 RZ_CONSTANT*)e0, b = *(const rz_float4 RZ_CONSTANT*)e1, c = *(const rz_float4 RZ_CONSTANT*)e2;
            if (mine) {
                const float q0 = r0 * w[q][0], q1 = r1 * w[q][1], q2 = r2 * w[q][2];                /* graphics.c:426-429 */
                const float norm = 1 / (q0 + q1 + q2);
                float v[4];
                v[0] = (a.x * q0 + b.x * q1 + c.x * q2) * norm;                                   /* :431-434 */
                v[1] = (a.y * q0 + b.y * q1 + c.y * q2) * norm;
                v[2] = (a.z * q0 + b.z * q1 + c.z * q2) * norm;
                P.color[pixel[q]] = sky_fragment(D, v, below[q]);
            }
        }
    }
}

/* over the family's work list: a fixed number of workgroups strides over it, so no wave is spent on
 * pixels without work.  The list holds 32x8 tiles (one per workgroup and turn, a wave takes two rows
 * of 32 pixels) or, with BLOCKS, the 16x8 blocks rz_strip_kernel lists (two per workgroup and turn,
 * a wave takes four rows of 16 pixels). */
template <int FAMILY, bool NARROW, bool BLOCKS>
__global__ void __launch_bounds__(256) rz_shade_list_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint32_t *__restrict__ ref_base,
        const uint32_t *__restrict__ queue_base, const uint32_t *__restrict__ queue_counts, uint32_t queue_capacity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LdsTables::fill();              /* a workgroup shades many blocks: the tables' copy is paid once */
    /* block lists come from rz_strip_kernel, which takes passes without blending only; tile lists may hold
     * blended fragments (passes cut up by rz_blend_visibility_kernel), which need the colour itself */
    typedef typename std::conditional<BLOCKS, LdsTablesBytes, LdsTables>::type Tables;
    /* lane q holds the length of part q of this family's list */
    constexpr uint32_t LIST = FAMILY == 2 ? RZ_SKY_LIST : (uint32_t)FAMILY;     /* (skybox fragments that came through the bins) */
    const uint32_t my_count = min(queue_counts[(LIST * RZ_QUEUE_WAYS + lane) * RZ_QUEUE_PAD], queue_capacity);
    uint32_t longest = my_count;
    longest = (uint32_t)wave_max_i((int)longest);
    constexpr uint32_t per_turn = BLOCKS ? 2u : 1u;
    /* (wave-uniform, and said so: everything read through `k` -- the item, the pass's parameters, the
     * draw record -- must come through scalar loads) */
    const uint32_t half = BLOCKS ? (uint32_t)__builtin_amdgcn_readfirstlane(wave >> 1) : 0u;
    for (uint32_t k = blockIdx.x * per_turn + half; k < longest * RZ_QUEUE_WAYS; k += gridDim.x * per_turn) {
        const uint32_t part = k & (RZ_QUEUE_WAYS - 1u), index = k / RZ_QUEUE_WAYS;
        if (index >= (uint32_t)__builtin_amdgcn_readlane((int)my_count, (int)part)) continue;
        const uint32_t item = queue_base[(size_t)(LIST * RZ_QUEUE_WAYS + part) * queue_capacity + index];
        if constexpr (BLOCKS) {
            const PassParams P = passes[item >> 21];
            const int t = (int)(threadIdx.x & 127u);
            shade_pixel<FAMILY, NARROW, Tables>(P, tri_base, fan_base, vary_base, fan_vary_base, upload_base, ref_base,
                                                (int)(item & 1023u) * 16 + (t & 15), (int)((item >> 10) & 2047u) * 8 + (t >> 4));
        } else {
            const PassParams P = passes[item >> 20];
            shade_pixel<FAMILY, NARROW, Tables>(P, tri_base, fan_base, vary_base, fan_vary_base, upload_base, ref_base,
                                                (int)(item & 511u) * RZ_TILE_W + (lane & 31), (int)((item >> 9) & 2047u) * RZ_TILE_H + wave * 2 + (lane >> 5));
        }
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

/* texture_from_depthbuffer (texture.c:147-158): the texel is (d, d, d, 1), stored as one float and
 * expanded on fetch; texture_from_colorbuffer (texture.c:130-145): byte / 255.0f per channel. */

/* one launch for several such copies (grid z = copy): e.g. the recorded copies behind one batch of passes */
#define RZ_COPY_GROUP 32
struct CopyJob { const void *from; void *to; int32_t w, h, flags, is_depth; const uint32_t *producer; };
struct CopyJobs { CopyJob job[RZ_COPY_GROUP]; };
__global__ void rz_buffers_to_textures_kernel(const CopyJobs jobs) {
    const CopyJob &j = jobs.job[blockIdx.z];
    /* a workgroup's 16x16 texels are four tiles of the image's 8x8 layout: a wave takes one of them, so that it writes 256
     * contiguous bytes (a row of 16 texels per 16 lanes wrote them as 32-byte pieces of two tiles: a 2048 x 2048 shadow map
     * copy ran at 3.2 TB/s of read + written bytes) */
    const int tile = threadIdx.x >> 6, within = threadIdx.x & 63;
    int x = blockIdx.x * 16 + (tile & 1) * 8 + (within & 7), y = blockIdx.y * 16 + (tile >> 1) * 8 + (within >> 3);
    if (x >= j.w || y >= j.h) return;
    /* the control block of the pass that rendered the source: when that pass did not run (it is run
     * again later, and this copy after it) the texture keeps what it holds */
    if (j.producer && j.producer[2] != 0u) return;
    if (j.is_depth) {
        ((float*)j.to)[texel_index(j.w, j.flags, x, y)] = ((const float*)j.from)[(size_t)y * j.w + x];
    } else {
        uint32_t c = ((const uint32_t*)j.from)[(size_t)y * j.w + x];
        ((float4*)j.to)[texel_index(j.w, j.flags, x, y)] =
            make_float4((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f,
                        (float)((c >> 16) & 0xffu) / 255.0f, (float)(c >> 24) / 255.0f);
    }
}

/* The same for depth buffers whose sides are multiples of 16 into tiled images, four texels per thread: one wave per
 * 16x16 texels (grid as above, 64 threads), a lane reads 16 bytes of a row and writes them as 16 bytes of a tile.  With
 * a thread per texel the copy of 32 shadow maps of 2048 x 2048 was two million waves of two instructions' work each and
 * ran at the rate waves can be started, 3.2 TB/s of read + written bytes. */
__global__ void __launch_bounds__(64) rz_depth_to_textures_kernel(const CopyJobs jobs) {
    const CopyJob &j = jobs.job[blockIdx.z];
    const int lane = threadIdx.x, row = lane >> 2, quad = lane & 3;
    const int x = blockIdx.x * 16 + quad * 4, y = blockIdx.y * 16 + row;
    if (x >= j.w || y >= j.h) return;
    if (j.producer && j.producer[2] != 0u) return;
    const float4 d = *(const float4*)((const float*)j.from + (size_t)y * j.w + x);
    *(float4*)((float*)j.to + texel_index(j.w, j.flags, x, y)) = d;       /* x is a multiple of 4: the four texels are neighbours in their tile */
}

/* a material's maps side by side (DrawRec::bundle): word k of texel i = map k's texel i, 0 where there is no map */
struct BundleMaps { const uint32_t *map[8]; };
__global__ void rz_bundle_kernel(uint32_t *__restrict__ bundle, const BundleMaps maps, size_t texels, int words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= texels) return;
    for (int k = 0; k < words; k++) bundle[i * (size_t)words + k] = maps.map[k] ? maps.map[k][i] : 0u;
}

/* RGBA8 -> RGB8.  A thread takes sixteen pixels, 64 bytes in as four 16-byte loads and 48 out as three 16-byte stores,
 * when the destination is 16-byte aligned (a frame of a gather buffer is); otherwise, and for the last pixels of a frame
 * that is no multiple of sixteen, four pixels with 4-byte stores or pixel by pixel.  (Four pixels per thread everywhere:
 * 14.5 MB of a 1080p frame moved in 5.2 us.) */
RZ_DEV void strip_alpha_four(uint4 c, uint32_t &o0, uint32_t &o1, uint32_t &o2) {
    const uint32_t a = c.x & 0xffffffu, b = c.y & 0xffffffu, d = c.z & 0xffffffu, e = c.w & 0xffffffu;
    o0 = a | (b << 24);
    o1 = (b >> 8) | (d << 16);
    o2 = (d >> 16) | (e << 8);
}
__global__ void rz_strip_alpha_kernel(const uint32_t *__restrict__ color, uint32_t *__restrict__ rgb, size_t pixels) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;         /* group of sixteen pixels */
    if (t * 16 >= pixels) return;
    const bool aligned = ((uintptr_t)rgb & 15u) == 0;                       /* uniform */
    if (aligned && t * 16 + 16 <= pixels) {
        const uint4 *in = (const uint4*)color + t * 4;
        const uint4 c0 = in[0], c1 = in[1], c2 = in[2], c3 = in[3];
        uint4 o0, o1, o2;
        strip_alpha_four(c0, o0.x, o0.y, o0.z);
        strip_alpha_four(c1, o0.w, o1.x, o1.y);
        strip_alpha_four(c2, o1.z, o1.w, o2.x);
        strip_alpha_four(c3, o2.y, o2.z, o2.w);
        uint4 *out = (uint4*)rgb + t * 3;
        out[0] = o0; out[1] = o1; out[2] = o2;
        return;
    }
    for (size_t q = t * 4; q < t * 4 + 4 && q * 4 < pixels; q++) {         /* groups of four pixels */
        if (q * 4 + 4 <= pixels) {
            uint32_t o0, o1, o2;
            strip_alpha_four(((const uint4*)color)[q], o0, o1, o2);
            rgb[q * 3 + 0] = o0; rgb[q * 3 + 1] = o1; rgb[q * 3 + 2] = o2;
        } else {
            unsigned char *out = (unsigned char*)rgb;
            for (size_t p = q * 4; p < pixels; p++) {
                const uint32_t c = color[p];
                out[p * 3] = (unsigned char)c; out[p * 3 + 1] = (unsigned char)(c >> 8); out[p * 3 + 2] = (unsigned char)(c >> 16);
            }
        }
    }
}

/* private_blit_bgr / private_blit_rgb (private.c:8-50): vertical flip, optional red/blue swap; the
 * fourth byte of the staged pixel is zero (the host keeps the destination's own) */
__global__ void rz_blit_kernel(const uint32_t *__restrict__ color, uint32_t *__restrict__ staged, int w, int h, int swap) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t c = color[(size_t)y * w + x];
    const uint32_t r = c & 0xffu, g = (c >> 8) & 0xffu, b = (c >> 16) & 0xffu;
    staged[(size_t)(h - 1 - y) * w + x] = swap ? (b | (g << 8) | (r << 16)) : (r | (g << 8) | (b << 16));
}

/* 2-D overlays (draw2d.c).  A line segment walks its longer axis pixel by pixel -- one thread per
 * step -- and takes the other coordinate as (int)(a + (b - a) * t), t = step / distance in single
 * precision (draw2d.c:43-73: rows when the row distance is the larger one, columns otherwise, a
 * single pixel when both are zero).  Up to three segments of one colour per launch (a triangle's
 * outline, draw2d.c:114-126): blockIdx.y = segment; where they meet they write the same word. */
struct OverlaySegments { int32_t r0[3], c0[3], r1[3], c1[3]; uint32_t pixel; int32_t width; };
__global__ void __launch_bounds__(64) rz_overlay_segments_kernel(uint32_t *__restrict__ color, const OverlaySegments s) {
    const int k = blockIdx.y;
    const int r0 = s.r0[k], c0 = s.c0[k], r1 = s.r1[k], c1 = s.c1[k];
    const int dr = abs(r0 - r1), dc = abs(c0 - c1);
    const bool along_rows = dr > dc;
    int a0 = along_rows ? r0 : c0, a1 = along_rows ? r1 : c1;
    int b0 = along_rows ? c0 : r0, b1 = along_rows ? c1 : r1;
    const int dist = along_rows ? dr : dc;
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    const int a = a0 + (int)(blockIdx.x * 64 + threadIdx.x);
    if (a > a1) return;
    int b = b0;
    if (dist != 0) {
        const float t = (float)(a - a0) / (float)dist;
        b = (int)((float)b0 + (float)(b1 - b0) * t);
    }
    const int row = along_rows ? a : b, col = along_rows ? b : a;
    color[(size_t)row * s.width + col] = s.pixel;
}

/* draw2d.c:75-93: rows [0, h) x columns [0, w) of the image, from its row 0 on, are added byte-wise
 * (wrapping) to red, green and blue of the pixels from (row, col) on; alpha stays */
__global__ void __launch_bounds__(64) rz_overlay_texture_kernel(uint32_t *__restrict__ color, int fb_width, const void *__restrict__ texels,
                                                                int tex_width, int flags, int row, int col, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t i = texel_index(tex_width, flags, x, y);
    f4 t;
    if (flags & RZ_TEX_NARROW) t = decode_word(flags, ((const float*)texels)[i]);
    else { const float4 q = ((const float4*)texels)[i]; t = mk4(q.x, q.y, q.z, q.w); }
    auto u8 = [](float v) { return (uint32_t)(int)(v * 255) & 0xffu; };        /* maths.c:33 */
    uint32_t *p = color + (size_t)(row + y) * fb_width + (col + x);
    const uint32_t d = *p;
    *p = (d & 0xff000000u) | ((d + u8(t.x)) & 0xffu) | (((d >> 8) + u8(t.y)) & 0xffu) << 8 | (((d >> 16) + u8(t.z)) & 0xffu) << 16;
}

/* row-major float4 image -> device layout (8x8 tiles where the size allows) */
__global__ void rz_texture_tile_kernel(const float4 *linear, float4 *texels, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, x, y)] = linear[(size_t)y * w + x];
}

/* the same for the 4-byte packed formats */
__global__ void rz_texture_tile32_kernel(const uint32_t *linear, uint32_t *texels, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, x, y)] = linear[(size_t)y * w + x];
}

/* device layout -> row-major float4 (texture_sync) */
/* the RZ_TEX_DISPLAY8 plane of an image, texel by texel in the image's own layout */
__global__ void rz_texture_display_kernel(const void *texels, uint32_t *bytes, size_t count, int flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    f4 c;
    if (flags & RZ_TEX_NARROW) c = decode_word(flags, __uint_as_float(((const uint32_t*)texels)[i]));
    else { const float4 q = ((const float4*)texels)[i]; c = mk4(q.x, q.y, q.z, q.w); }
    c = mk4(d_sat(c.x), d_sat(c.y), d_sat(c.z), d_sat(c.w));
    bytes[i] = pack_rgb(c, 0u);
}

__global__ void rz_texture_untile_kernel(const void *texels, float4 *linear, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    uint32_t i = texel_index(w, flags, x, y);
    float4 v;
    if (flags & RZ_TEX_NARROW) {            /* one word per texel: decoded to the floats the reference's texture holds */
        const f4 t = decode_word(flags, ((const float*)texels)[i]);
        v = make_float4(t.x, t.y, t.z, t.w);
    }
    else v = ((const float4*)texels)[i];
    linear[(size_t)y * w + x] = v;
}

#endif
