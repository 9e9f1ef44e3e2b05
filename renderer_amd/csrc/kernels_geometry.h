/* Part of kernels.h (included there, in this order; not a header of its own): vertex shaders and triangle set-up (K1), skeletal poses (K0), clipping (K2). */

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

/* What the passes' kernels count in or add into starts at zero.  zero_lane_scratch: the part nothing before the bin
 * count looks at (block `block` of RZ_BEGIN_BLOCKS of pass `pass`); rz_begin_kernel: that and what the vertex kernel
 * itself adds into -- the pass's control block and its table of sky triangles. */
#define RZ_BEGIN_BLOCKS 64u
RZ_DEV void zero_lane_scratch(const PassParams &P, uint32_t pass, uint32_t block) {
    const uint32_t gv = block * 256u + threadIdx.x, stride = RZ_BEGIN_BLOCKS * 256u;
    if (pass == 0 && gv <= RZ_BATCH_CONTESTED) P.queue_counts[gv * RZ_QUEUE_PAD] = 0;    /* the batch's work-list lengths and its flag word */
    for (uint32_t t = gv; t < (uint32_t)(P.tiles_x * P.tiles_y); t += stride) P.tile_listed[t] = 0;
    {   /* ... and the pass's (chunk, bin) matrix, which rz_bin_count_kernel adds into (four words per store: the
         * matrix of a batch's pass starts at a multiple of four words when every pass before it has a multiple of four
         * cells, else word by word) */
        const uint32_t cells = (((uint32_t)P.num_tris + 63u) / 64u) * (uint32_t)(P.bins_x * P.bins_y);
        uint32_t RZ_GLOBAL *counts = P.bin_counts;
        if ((((size_t)counts) & 15u) == 0u) {
            const uint32_t quads = cells / 4u;
            for (uint32_t t = gv; t < quads; t += stride) ((uint4 RZ_GLOBAL*)counts)[t] = make_uint4(0u, 0u, 0u, 0u);
            if (gv < (cells & 3u)) counts[quads * 4u + gv] = 0;
        } else {
            for (uint32_t t = gv; t < cells; t += stride) counts[t] = 0;
        }
    }
}
__global__ void __launch_bounds__(256) rz_begin_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    const uint32_t gv = blockIdx.x * 256u + threadIdx.x;
    /* the pass's device-written scalars start at zero -- unless another pass of the flush ran out of
     * scratch and this one may depend on it: then it is marked to be left out and run again later */
    if (gv < 8u) P.control[gv] = (gv == 2u && P.level > 0 && *P.flush_flag != 0u) ? (uint32_t)RZ_ERR_SKIPPED : 0u;
    if (P.sky_draws && gv < (uint32_t)RZ_SKY_TABLE) P.sky_table[gv] = 0xffffffffu;      /* RZ_REF_NONE: no sky triangle yet */
    zero_lane_scratch(P, blockIdx.y, blockIdx.x);
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
template <bool ZEROES>
RZ_DEV void vertices_of_workgroup(const PassParams *__restrict__ passes, uint32_t vertex_blocks) {
    const PassParams &P = passes[blockIdx.y];
    if constexpr (ZEROES) {
        if (blockIdx.x >= vertex_blocks) {          /* (a batch without rz_begin_kernel: the workgroups behind the vertices') */
            zero_lane_scratch(P, blockIdx.y, blockIdx.x - vertex_blocks);
            return;
        }
    }
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
            P.clip_list.as<uint2>()[atomicAdd(&P.control[3], 1u)] = make_uint2(t, (uint32_t)dt);    /* (with its draw: the clipper need not search for it) */
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
__global__ void __launch_bounds__(256) rz_vertex_kernel(const PassParams *__restrict__ passes) {
    vertices_of_workgroup<false>(passes, 0u);
}
/* ... with workgroups behind the vertices' that zero the lane's scratch (zero_lane_scratch): the first kernel of a batch
 * that starts without rz_begin_kernel.  A kernel of its own: the branch alone cost the plain one a fifth of its time. */
__global__ void __launch_bounds__(256) rz_vertex_zero_kernel(const PassParams *__restrict__ passes, uint32_t vertex_blocks) {
    vertices_of_workgroup<true>(passes, vertex_blocks);
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
        const uint2 queued_item = P.clip_list.as<const uint2>()[item];
        const uint32_t t = queued_item.x;
        const int d = (int)queued_item.y;
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
            if (cross == 0u && inside == (1u << n) - 1u) continue;     /* every corner inside: the plane leaves the polygon as it is (uniform) */
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
