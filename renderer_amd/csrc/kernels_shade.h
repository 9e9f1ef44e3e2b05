/* Part of kernels.h (included there, in this order; not a header of its own): the sky kernel and the per-family shade kernels (K5). */

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
 * this kernel then carries out for the pixels no sky triangle covers.
 *
 * sky_tile is a wave's part of it: tile (tile_x, tile_y) of pass P, wave 0 or 1 of the tile.  WHICH leaves bins out
 * (a constant wherever the function is inlined): RZ_SKY_FRESH_ONLY those with lists, RZ_SKY_STALE_ONLY those without --
 * nothing in a bin without a list waits for the strips, and in a launch of a frame or two the strips' kernel takes the
 * sky along in those two parts (rz_strip_team_kernel) instead of leaving it for a launch behind it.
 * DEFERS: beside the shade kernels (rz_sky_side_kernel).  There the sky must not take a pixel that holds a fragment
 * for them -- colour and reference would go to whoever wrote last -- and it can only where a fragment lies behind the
 * sky's depth (RZ_SKY_NEAR; the strips flag such a pass, control word 5): the bins with lists of a flagged pass are left
 * to rz_sky_late_kernel, which runs when the shade kernels are through. */
enum { RZ_SKY_ALL = 0, RZ_SKY_FRESH_ONLY = 1, RZ_SKY_STALE_ONLY = 2 };
template <int PPL, bool DEFERS, bool BYTES_ONLY>
RZ_DEV void sky_tile(const PassParams &P, int tile_x, int tile_y, int wave, int lane, int WHICH,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    static_assert(PPL == 2, "a wave's region is 32x4 pixels");
    const uint32_t failed = P.control[2];             /* looked at below, in the same round trip as the bin's record */
    const uint32_t contested = DEFERS ? P.control[5] : 0u;
    const TriRec *tri_table = tri_base + P.tri_off;
    const TriRec *fan_table = fan_base + (size_t)P.fan_off * RZ_MAX_FAN;
    const float *vary_table = vary_base + (size_t)P.tri_off * 3 * RZ_VARY_STRIDE;
    const float *fan_vary_table = fan_vary_base + (size_t)P.fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
    const DrawRec *draw_table = (const DrawRec*)(upload_base + P.draws_off);
    const int tx0 = tile_x * RZ_TILE_W, ry0 = tile_y * RZ_TILE_H + wave * 2 * PPL;
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
    if ((WHICH == RZ_SKY_FRESH_ONLY && !fresh) || (WHICH == RZ_SKY_STALE_ONLY && fresh)) return;
    const bool left = DEFERS && contested != 0u && !fresh;      /* rz_sky_late_kernel's: nothing to do here, as if no entry touched the region */
    const bool overflow = !left && sky_count == RZ_SKY_BIN_OVERFLOW;
    const uint32_t entries = overflow || left ? 0u : (sky_word >> ((region & 3u) * 8u)) & 0xffu;
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
        /* (the first two entries in line -- the usual region has one or two --, the rest in a loop: eight copies of the
         * test were a third of the kernel, which shares the instruction cache with the shade kernel beside it) */
        if (entries & 1u) consider(refs[0]);
        if (entries & 2u) consider(refs[1]);
        for (uint32_t left = entries >> 2; left != 0u; left &= left - 1u) {
            const uint32_t e = (uint32_t)__builtin_ctz(left);
            uint32_t ref = refs[2];
#pragma unroll
            for (uint32_t k = 3; k < RZ_SKY_BIN_MAX; k++) if (e + 2u == k) ref = refs[k];
            consider(ref);
        }
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
        if (BYTES_ONLY || (D.flags & RZ_DRAW_SKY_BYTES)) {      /* both texels asked for before either is waited for */
            const uint32_t RZ_GLOBAL *bytes = D.maps[0].data.as<const uint32_t>();
            for (int q = 0; q < PPL; q++) at[q] = sky_bytes_index(D, v[q]);
            uint32_t word[PPL];
            for (int q = 0; q < PPL; q++) word[q] = bytes[at[q]];
            for (int q = 0; q < PPL; q++) P.color[pixel[q]] = (word[q] & 0x00ffffffu) | (below[q] & 0xff000000u);
        } else {
            for (int q = 0; q < PPL; q++) P.color[pixel[q]] = sky_fragment<BYTES_ONLY>(D, v[q], below[q]);
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
                P.color[pixel[q]] = sky_fragment<BYTES_ONLY>(D, v, below[q]);
            }
        }
    }
}

/* (every sky draw of the batch with its cube's display bytes: the usual case) */
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    const PassParams P = passes[blockIdx.z];
    if ((int)blockIdx.x >= P.tiles_x || (int)blockIdx.y >= P.tiles_y || P.sky_draws == 0) return;
    sky_tile<PPL, false, true>(P, (int)blockIdx.x, (int)blockIdx.y, (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63u), RZ_SKY_ALL,
                              tri_base, fan_base, vary_base, fan_vary_base, upload_base, sky_base);
}
/* ... a batch with a sky draw whose cube has no plane of display bytes (faces of different sizes, a blended sky) */
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_float_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    const PassParams P = passes[blockIdx.z];
    if ((int)blockIdx.x >= P.tiles_x || (int)blockIdx.y >= P.tiles_y || P.sky_draws == 0) return;
    sky_tile<PPL, false, false>(P, (int)blockIdx.x, (int)blockIdx.y, (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63u), RZ_SKY_ALL,
                              tri_base, fan_base, vary_base, fan_vary_base, upload_base, sky_base);
}
/* ... beside the shade kernels, on the lane's side stream */
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_side_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    const PassParams P = passes[blockIdx.z];
    if ((int)blockIdx.x >= P.tiles_x || (int)blockIdx.y >= P.tiles_y || P.sky_draws == 0) return;
    sky_tile<PPL, true, true>(P, (int)blockIdx.x, (int)blockIdx.y, (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63u), RZ_SKY_ALL,
                              tri_base, fan_base, vary_base, fan_vary_base, upload_base, sky_base);
}
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_side_float_kernel(const PassParams *__restrict__ passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base) {
    const PassParams P = passes[blockIdx.z];
    if ((int)blockIdx.x >= P.tiles_x || (int)blockIdx.y >= P.tiles_y || P.sky_draws == 0) return;
    sky_tile<PPL, true, false>(P, (int)blockIdx.x, (int)blockIdx.y, (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63u), RZ_SKY_ALL,
                              tri_base, fan_base, vary_base, fan_vary_base, upload_base, sky_base);
}
/* ... and what that one left: the bins with lists of the passes flagged as holding fragments the sky may hide.  Hardly
 * ever any (geometry between the sky's depth and the far plane): the batch's flag word says so to every workgroup in one
 * load; otherwise the workgroups stride over the flagged passes' tiles. */
template <int PPL>
__global__ void __launch_bounds__(256 / PPL) rz_sky_late_kernel(const PassParams *__restrict__ passes, uint32_t num_passes,
        const TriRec *__restrict__ tri_base, const TriRec *__restrict__ fan_base,
        const float *__restrict__ vary_base, const float *__restrict__ fan_vary_base,
        const char *__restrict__ upload_base, const SkyBin *__restrict__ sky_base, const uint32_t *__restrict__ queue_counts) {
    if (queue_counts[RZ_BATCH_CONTESTED * RZ_QUEUE_PAD] == 0u) return;
    const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    for (uint32_t p = 0; p < num_passes; p++) {
        const PassParams P = passes[p];
        if (P.sky_draws == 0 || P.control[5] == 0u) continue;
        for (uint32_t tile = blockIdx.x; tile < (uint32_t)(P.tiles_x * P.tiles_y); tile += gridDim.x)
            sky_tile<PPL, false, false>(P, (int)(tile % (uint32_t)P.tiles_x), (int)(tile / (uint32_t)P.tiles_x), wave, lane, RZ_SKY_STALE_ONLY,
                                                    tri_base, fan_base, vary_base, fan_vary_base, upload_base, sky_base);
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
