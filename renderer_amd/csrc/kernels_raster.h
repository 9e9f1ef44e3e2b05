/* Part of kernels.h (included there, in this order; not a header of its own): rasterisers: the ordered kernel of passes with blending, blend visibility, bin classes and sky records, strips (K4). */

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
/* BYTES_ONLY: every sky draw the kernel can meet has its cube's plane of display bytes (the host has looked); the
 * float texels' path is left out of the kernel -- it is half of the sky kernel's code, and that kernel shares the
 * instruction cache with the shade kernel that runs beside it */
template <bool BYTES_ONLY = false>
RZ_DEV uint32_t sky_fragment(const DrawRec &D, const float *v, uint32_t below) {
    if (BYTES_ONLY || (D.flags & RZ_DRAW_SKY_BYTES)) {
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
enum { RZ_BIN_STRIP = 0, RZ_BIN_DONE = 1, RZ_BIN_CROWD = 2 };
/* a 16x8 block on a work list */
#define RZ_BLOCK_ITEM(pass, by, bx) (((uint32_t)(pass) << 21) | ((uint32_t)(by) << 10) | (uint32_t)(bx))
static_assert(RZ_SKY_BIN_MAX == RZ_CLASS_SELECTED, "a wave's list of selected table entries");
/* (a wave's part of rz_bin_fill_kernel's launch: the workgroups behind those that fill -- neither reads what the other
 * writes, and in a launch of a frame or two each is as long as its chain of round trips, which now run side by side) */
RZ_DEV void bin_class_of(const PassParams *__restrict__ passes, uint32_t pass_index, uint32_t bin_index, int lane, uint32_t *s_sel,
                         const ClassTables &tables) {
    const TriRec *__restrict__ tri_base = tables.tri_base, *__restrict__ fan_base = tables.fan_base;
    const uint2 *__restrict__ bin_range_base = tables.bin_range_base;
    SkyBin *__restrict__ sky_base = tables.sky_base;
    uint32_t *__restrict__ class_base = tables.class_base;
    const PassParams &P = passes[pass_index];
    if ((int)bin_index >= P.bins_x * P.bins_y) return;
    uint32_t *bin_class = class_base + P.bin_off + bin_index;
    if (P.control[2] != 0u) {                                   /* out of scratch, or left out: the pass is run again later */
        if (lane == 0) *bin_class = RZ_BIN_DONE;
        return;
    }
    const uint2 range = (bin_range_base + P.bin_off)[bin_index];
    const int bin_x0 = (int)(bin_index % (uint32_t)P.bins_x) * RZ_BIN, bin_y0 = (int)(bin_index / (uint32_t)P.bins_x) * RZ_BIN;
    /* a crowded bin is left to rz_crowd_kernel: its 16x8 blocks go on that kernel's work list, block by lane */
    const bool crowd = P.crowd_min > 0 && range.y >= (uint32_t)P.crowd_min;
    if (lane == 0) *bin_class = range.y == 0u ? (uint32_t)RZ_BIN_DONE : crowd ? (uint32_t)RZ_BIN_CROWD : (uint32_t)RZ_BIN_STRIP;
    if (crowd && lane < (RZ_BIN / 16) * (RZ_BIN / 8)) {
        const uint32_t block_x = (uint32_t)bin_x0 / 16u + (uint32_t)lane % (RZ_BIN / 16u), block_y = (uint32_t)bin_y0 / 8u + (uint32_t)lane / (RZ_BIN / 16u);
        if ((int)(block_x * 16u) < P.width && (int)(block_y * 8u) < P.height) {
            const uint32_t way = RZ_CROWD_LIST * RZ_QUEUE_WAYS + ((block_y * 2u * (uint32_t)P.tiles_x + block_x) & (RZ_QUEUE_WAYS - 1u));
            const uint32_t at = atomicAdd(&P.queue_counts[way * RZ_QUEUE_PAD], 1u);
            if (at < P.queue_capacity)
                P.queue[(size_t)way * P.queue_capacity + at] = RZ_BLOCK_ITEM(pass_index, block_y, block_x);
            else
                atomicOr(&P.control[2], (uint32_t)RZ_ERR_QUEUE_OVERFLOW);
        }
    }
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
    SkyBin *rec = sky_base + P.bin_off + bin_index;
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
RZ_DEV void finish_block(const PassParams &P, uint32_t pass_index, int bx0, int by0, int lane,
                         const bool valid[2], const uint32_t pixel[2], const float zcur[2], const uint32_t ref_of[2]) {
    constexpr int PPL = 2;
    bool listed = false;
    if (P.sky_draws != 0) {
        /* a fragment the pass's sky box may hide: said for the pass and for the batch (rz_sky_side_kernel, rz_sky_late_kernel) */
        bool far_one = false;
        for (int q = 0; q < PPL; q++) far_one |= ref_of[q] != RZ_REF_NONE && !(zcur[q] < RZ_SKY_NEAR);
        if (__any(far_one) && lane == 0) {
            atomicOr(&P.control[5], 1u);
            atomicOr(&P.queue_counts[RZ_BATCH_CONTESTED * RZ_QUEUE_PAD], 1u);
        }
    }
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

#ifdef RZ_PROBE
#define RZ_PROBE_WORDS 12
#define RZ_PROBE_MAX (1 << 19)
__device__ unsigned long long rz_probe_data[RZ_PROBE_WORDS * RZ_PROBE_MAX];
__device__ unsigned int rz_probe_count;
#endif
/* The walk of rz_strip_kernel (see there): STRIP blocks of 16x8 pixels one above the other from (bx0, sy0) down, all in
 * the bin whose list is given.  SPLIT > 1: the workgroup's waves share ONE block (STRIP == 1) -- wave `part` takes
 * the list's groups of 64 entries part, part + SPLIT, ... -- and their results meet in `merged`, a key per pixel.  Which
 * fragment a pixel ends up with does not depend on the order the triangles are taken in (least depth, the latest in
 * submission order among equals), so any cut of the list gives the same picture. */
template <int STRIP, int SPLIT>
RZ_DEV void strip_walk(const PassParams &P, uint32_t pass_index, int probe_bin, const BinEntry *__restrict__ list, const uint2 *__restrict__ boxes, uint32_t count,
                       const float *__restrict__ vary_table, const float *__restrict__ fan_vary_table, const DrawRec *__restrict__ draw_table,
                       int bx0, int sy0, int lane, int part, unsigned long long *keys, uint32_t *ring, unsigned long long *merged) {
    constexpr int PPL = 2;
    static_assert(SPLIT == 1 || STRIP == 1, "waves share one block");
    const int bx1 = bx0 + 8 * PPL - 1;
    int px[PPL];
    for (int q = 0; q < PPL; q++) px[q] = bx0 + 8 * q + (lane & 7);
#ifdef RZ_PROBE
    const unsigned long long probe_t0 = wall_clock64();
    unsigned long long probe_ordered = 0, probe_ring = 0, probe_groups = 0, probe_hits = 0, probe_survivors = 0, probe_cutoffs = 0, probe_batches = 0, probe_small = 0;
#endif

    /* Small triangles.  Which fragment a pixel ends up with does not depend on the order the
     * triangles are taken in: it is the one of least depth, the latest in submission order among
     * equals (graphics.c:523 keeps a fragment when depth <= the stored depth).  Candidates whose
     * box covers at most micro_area pixels of this wave's block are therefore set aside in a ring,
     * and each time 64 have come together every lane rasterises one of them, pixel by pixel,
     * into a per-pixel key in LDS: depth (as an order-preserving integer) above the complement of
     * the list position, so that an atomic minimum keeps exactly that fragment.  The keys meet the
     * registers of the one-triangle-per-step path when the list ends.  Triangles with an alpha
     * cutoff and passes that count fragments stay on the ordered path. */
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
        const uint32_t first = (uint32_t)part * 64u + (uint32_t)lane;
        if (first < count) {
            if (long_list) {
                nbox = boxes[first];
            } else {
                const float4 *e = (const float4*)&list[first];         /* reference + record, 80 bytes */
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

        for (uint32_t base = (uint32_t)part * 64u; base < count;) {
            const uint2 cbox = nbox;
            float4 c0 = n0, c1 = n1;
            const bool present = base + lane < count;
            const BinEntry *group = list + base;
            const uint32_t group_at = base;
            base += 64u * SPLIT;
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

        if constexpr (SPLIT > 1) {
            /* the waves' fragments meet: depth (as an order-preserving integer) above the complement of the list position,
             * as the small triangles' keys above; the first wave takes what the minimum kept */
            if (part == 0) for (int q = 0; q < PPL; q++) merged[q * 64 + lane] = ~0ull;
            __syncthreads();
            for (int q = 0; q < PPL; q++) {
                if (ord_of[q] < 0) continue;
                const uint32_t bits = __float_as_uint(zcur[q] + 0.0f);
                const uint32_t up = bits ^ ((uint32_t)((int32_t)bits >> 31) | 0x80000000u);
                atomicMin(&merged[q * 64 + lane], ((unsigned long long)up << 32) | (uint32_t)~(uint32_t)ord_of[q]);
            }
            __syncthreads();
            if (part == 0) {
                for (int q = 0; q < PPL; q++) {
                    const unsigned long long key = merged[q * 64 + lane];
                    if (key == ~0ull) continue;                     /* no fragment: zcur[q] is what the buffer held */
                    const uint32_t up = (uint32_t)(key >> 32);
                    const int ord = (int)~(uint32_t)key;
                    if (ord == ord_of[q]) continue;                 /* this wave's own */
                    const BinEntry *e = &list[ord];
                    zcur[q] = __uint_as_float(up ^ ((up >> 31) ? 0x80000000u : 0xffffffffu));
                    ref_of[q] = e->tri | (ref_tag(e->rec.meta) << 28);
                }
                finish_block(P, pass_index, bx0, by0, lane, valid, pixel, zcur, ref_of);
            }
        } else {
            finish_block(P, pass_index, bx0, by0, lane, valid, pixel, zcur, ref_of);
        }
    }
#ifdef RZ_PROBE
    if (lane == 0) {
        const unsigned int at = atomicAdd(&rz_probe_count, 1u);
        if (at < RZ_PROBE_MAX) {
            unsigned long long *o = rz_probe_data + (size_t)at * RZ_PROBE_WORDS;
            o[0] = pass_index; o[1] = (unsigned long long)probe_bin; o[2] = count; o[3] = probe_groups; o[4] = probe_survivors; o[5] = probe_cutoffs;
            o[6] = probe_batches; o[7] = wall_clock64() - probe_t0; o[8] = probe_ordered; o[9] = probe_ring; o[10] = probe_small; o[11] = probe_t0;
        }
    }
#endif
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
/* (for a workgroup of rz_strip_team_kernel: what the sky of its tile is read from, see sky_tile) */
struct SkyTables { const TriRec *tri_base, *fan_base; const SkyBin *sky_base; };
template <int PPL, bool DEFERS, bool BYTES_ONLY = false>
RZ_DEV void sky_tile(const PassParams &P, int tile_x, int tile_y, int wave, int lane, int WHICH,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base);

/* a workgroup of the strip kernels: two strips side by side (a 32-pixel tile column), SPLIT waves to each.  SKY: the
 * tile's sky too (rz_strip_team_kernel) -- at once where the bin is empty, behind the strips where it has a list; both
 * ways lead to ONE copy of the sky's code at the end (inlined twice it tripled the kernel and spilled its scalar
 * registers; as a function the kernel calls it took 118 vector registers by the calling convention's numbering). */
template <int STRIP, int SPLIT, bool SKY>
RZ_DEV void strips_of_workgroup(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base, const uint2 *__restrict__ box_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ class_base, int uniform_width, int uniform_height, SkyTables sky) {
    static_assert(RZ_BIN % (8 * STRIP) == 0, "a strip lies inside one bin");
    static_assert(!SKY || STRIP == 1, "a workgroup's strips cover one tile");
    const TileOrder o = tile_order(RZ_BIN / (8 * STRIP));
    const uint32_t pass_index = o.pass, tile_column = o.column, strip_row = o.row;
    const int sy0 = (int)strip_row * 8 * STRIP, tx0 = (int)tile_column * RZ_TILE_W;
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool with_sky = SKY && sky.sky_base != nullptr;
    bool strips = true;                     /* uniform over the workgroup */
    int sky_which = -1, sky_half = 0;       /* uniform over the wave: what this wave does of the tile's sky, and for which four rows */
    /* the passes of the batch have one size (uniform_width != 0): what the bin's class says is known before the pass's
     * parameters are asked for, and most strips of most frames end here */
    if (uniform_width) {
        if (tx0 >= uniform_width || sy0 >= uniform_height) return;
        const uint32_t bins_x = ((uint32_t)uniform_width + RZ_BIN - 1u) / RZ_BIN, bins_y = ((uint32_t)uniform_height + RZ_BIN - 1u) / RZ_BIN;
        const uint32_t bin_class = class_base[pass_index * (bins_x * bins_y) + (uint32_t)(sy0 / RZ_BIN) * bins_x + (uint32_t)(tx0 / RZ_BIN)];
        if (bin_class != (uint32_t)RZ_BIN_STRIP) {
            if (!(with_sky && bin_class == (uint32_t)RZ_BIN_DONE && wave < 2)) return;
            strips = false;
        }
    }
    const PassParams P = passes[pass_index];
    if ((int)tile_column >= P.tiles_x || sy0 >= P.height) return;
    if (strips && !uniform_width) {
        const uint32_t bin_class = class_base[P.bin_off + (uint32_t)((sy0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN)];
        if (bin_class != (uint32_t)RZ_BIN_STRIP) {
            if (!(with_sky && bin_class == (uint32_t)RZ_BIN_DONE && wave < 2)) return;
            strips = false;
        }
    }
    if (strips) {
        const BinEntry *list_table = list_base + P.list_off;
        const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
        const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
        const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
        const int side = wave & 1, part = wave >> 1;
        const int bin = (sy0 / RZ_BIN) * P.bins_x + tx0 / RZ_BIN;
        const uint2 range = (bin_range_base + P.bin_off)[bin];          /* {first entry, entries} */
        __shared__ unsigned long long micro_keys[2 * SPLIT][128];
        __shared__ uint32_t micro_ring[2 * SPLIT][128];
        __shared__ unsigned long long merged[SPLIT > 1 ? 2 : 1][128];
        strip_walk<STRIP, SPLIT>(P, pass_index, bin, list_table + range.x, box_base + P.list_off + range.x, range.y /* (a bin of this class is not empty) */,
                                 vary_table, fan_vary_table, draw_table, tx0 + side * 16, sy0, lane, part, micro_keys[wave], micro_ring[wave],
                                 SPLIT > 1 ? merged[side] : nullptr);
        if constexpr (SKY && SPLIT > 1) {
            /* the tile's own sky, once its two blocks are finished: by the waves that did not finish them (the tile's
             * upper four rows one, the lower four the other), against the depth the others have just written */
            if (with_sky && P.sky_draws != 0) {     /* uniform over the workgroup */
                __syncthreads();
                if (part == 1) { sky_which = 2 /* RZ_SKY_STALE_ONLY */; sky_half = side; }
            }
        }
    } else {
        /* an empty bin's tile: its sky, by the workgroup's first two waves -- nothing there waits for the strips */
        sky_which = 1 /* RZ_SKY_FRESH_ONLY */; sky_half = wave;
    }
    if constexpr (SKY) {
        if (sky_which >= 0 && P.sky_draws != 0)
            sky_tile<2, false, false>(P, (int)tile_column, (int)strip_row, sky_half, lane, sky_which,
                               sky.tri_base, sky.fan_base, vary_base, fan_vary_base, upload_base, sky.sky_base);
    }
}
template <int STRIP>
__global__ void __launch_bounds__(128) rz_strip_kernel(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base, const uint2 *__restrict__ box_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ class_base, int uniform_width, int uniform_height) {
    strips_of_workgroup<STRIP, 1, false>(passes, list_base, box_base, vary_base, fan_vary_base, upload_base, bin_range_base, class_base,
                                         uniform_width, uniform_height, SkyTables());
}
/* A launch of a frame or two (a viewer's frame-at-a-time loop) is as long as its longest wave, and a wave alone on its
 * SIMD issues an instruction every few cycles at best: some 900 cycles for each candidate that survives culling, whether
 * its record comes through scalar loads or out of the lanes' registers (helmet at 1080p, one frame: 2 500 waves on 1 024
 * SIMDs, the longest with 70 survivors and 37 us; whole entries loaded a group ahead and handed round with readlane: 33.9
 * -> 32.9 us, not kept).  So the waves are made shorter: a TEAM of SPLIT waves per block of one strip each, sharing the
 * list's groups as in rz_crowd_kernel (33.4 -> 25.6 us with two; four and more waves per block no longer fit the device
 * at once and take longer again, 31 us and up; sharing a group's survivors instead of the groups: the same 25.9).
 * Such a launch has wave slots to spare, and a workgroup whose bin is empty spends its first two waves on its tile's
 * sky (`sky` with tables; sky_tile): nothing there waits for the strips.  The sky of a tile whose bin has a list follows
 * when the tile's two blocks are finished, by the waves that did not finish them: no sky kernel behind such a launch
 * (strips and sky of a helmet frame 33.4 + 17.7 -> 34 us). */
template <int SPLIT>
__global__ void __launch_bounds__(128 * SPLIT) rz_strip_team_kernel(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base, const uint2 *__restrict__ box_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ class_base, int uniform_width, int uniform_height, SkyTables sky) {
    strips_of_workgroup<1, SPLIT, true>(passes, list_base, box_base, vary_base, fan_vary_base, upload_base, bin_range_base, class_base,
                                        uniform_width, uniform_height, sky);
}

/* Crowded bins.  A launch of rz_strip_kernel is as long as its longest wave, and a bin that holds the better part of a
 * mesh -- a head of hair in a few bins of 10 000 entries and more -- keeps 32 waves busy for the whole launch while the
 * rest of the device drains.  rz_bin_class_kernel puts the 16x8 blocks of such bins on a work list of their own, and here
 * the SPLIT waves of a workgroup share a block's list between them; the launch runs beside the strips, on the lane's side
 * stream. */
template <int SPLIT>
__global__ void __launch_bounds__(64 * SPLIT) rz_crowd_kernel(const PassParams *__restrict__ passes,
        const BinEntry *__restrict__ list_base, const uint2 *__restrict__ box_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const uint2 *__restrict__ bin_range_base,
        const uint32_t *__restrict__ queue_base, const uint32_t *__restrict__ queue_counts, uint32_t queue_capacity) {
    __shared__ unsigned long long micro_keys[SPLIT][128];
    __shared__ uint32_t micro_ring[SPLIT][128];
    __shared__ unsigned long long merged[128];
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t my_count = min(queue_counts[(RZ_CROWD_LIST * RZ_QUEUE_WAYS + (uint32_t)lane) * RZ_QUEUE_PAD], queue_capacity);
    /* The list's parts are dealt by a block's place on the screen, and crowded bins are a few places: most parts are
     * empty, a few long.  The workgroups therefore stride over the parts laid end to end (lane = part: a prefix over the
     * lanes says where each starts), not over every part in step -- striding part by part, one workgroup in eight did all
     * the work. */
    uint32_t total;
    const uint32_t before = wave_exclusive_scan(my_count, &total);
    for (uint32_t j = blockIdx.x; j < total; j += gridDim.x) {
        const int way = 63 - __clzll((long long)__ballot(before <= j));      /* the last part that starts at or before j (uniform) */
        const uint32_t index = j - (uint32_t)__builtin_amdgcn_readlane((int)before, way);
        const uint32_t item = queue_base[(size_t)(RZ_CROWD_LIST * RZ_QUEUE_WAYS + (uint32_t)way) * queue_capacity + index];
        const uint32_t pass_index = item >> 21;
        const PassParams P = passes[pass_index];
        const int bx0 = (int)(item & 1023u) * 16, by0 = (int)((item >> 10) & 2047u) * 8;
        const int bin = (by0 / RZ_BIN) * P.bins_x + bx0 / RZ_BIN;
        const uint2 range = (bin_range_base + P.bin_off)[bin];
        strip_walk<1, SPLIT>(P, pass_index, bin, list_base + P.list_off + range.x, box_base + P.list_off + range.x, range.y,
                             vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE, fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE,
                             (const DrawRec*)(upload_base + P.draws_off), bx0, by0, lane, wave, micro_keys[wave], micro_ring[wave], merged);
        __syncthreads();            /* (the first wave has read `merged` before any wave sets it up again) */
    }
}
