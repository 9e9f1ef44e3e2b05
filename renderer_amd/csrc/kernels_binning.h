/* Part of kernels.h (included there, in this order; not a header of its own): ordered per-bin triangle lists (K3). */

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

/* 256 threads = 16 bins x 16 segments of the chunk range.  (Not 64 bins: a workgroup of 16 waves finds room only on a
 * compute unit with 16 free wave slots, and next to a kernel of another lane that keeps every slot taken by
 * workgroups of a wave or two it waited for that kernel's end -- 20 us alone, 800 us beside a shade kernel, with
 * its lane's strips and shading behind it.) */
#define RZ_SCAN_SEGMENTS 16
#define RZ_SCAN_BINS 16
__global__ void __launch_bounds__(RZ_SCAN_BINS * RZ_SCAN_SEGMENTS) rz_bin_scan_kernel(const PassParams *__restrict__ passes) {
    const PassParams &P = passes[blockIdx.y];
    __shared__ uint32_t s_part[RZ_SCAN_SEGMENTS][RZ_SCAN_BINS];
    const int bins = P.bins_x * P.bins_y;
    if ((int)(blockIdx.x * RZ_SCAN_BINS) >= bins) return;
    const int column = threadIdx.x & (RZ_SCAN_BINS - 1);
    const int bin = (int)(blockIdx.x * RZ_SCAN_BINS) + column, segment = threadIdx.x / RZ_SCAN_BINS;
    const uint32_t chunks = ((uint32_t)P.num_tris + 63u) / 64u;
    const uint32_t per = (chunks + RZ_SCAN_SEGMENTS - 1u) / RZ_SCAN_SEGMENTS;
    const uint32_t c0 = min((uint32_t)segment * per, chunks), c1 = min(c0 + per, chunks);
    uint32_t *col = P.bin_counts + bin;
    uint32_t sum = 0;
    /* a segment of at most 16 chunks (a frame or two per launch) is kept in registers between the two passes over it:
     * read again it was two more memory round trips in a kernel that is nothing but round trips */
    const bool few = per <= 16u;                    /* uniform */
    uint32_t kept[16];
    if (few) {
#pragma unroll
        for (int u = 0; u < 16; u++) kept[u] = (bin < bins && c0 + u < c1) ? col[(size_t)(c0 + u) * bins] : 0u;
#pragma unroll
        for (int u = 0; u < 16; u++) sum += kept[u];
    } else if (bin < bins) {
        for (uint32_t c = c0; c < c1; c += 8u) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (c + u) < c1 ? col[(size_t)(c + u) * bins] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) sum += v[u];
        }
    }
    s_part[segment][column] = sum;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int q = 0; q < RZ_SCAN_SEGMENTS; q++) {
        uint32_t v = s_part[q][column];
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
        s_part[0][column] = total ? start : 0xffffffffu;
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
    if (threadIdx.x < 64u) {        /* the pass's longest list, for the host's choice of strip height next time (segment 0 is in wave 0) */
        uint32_t longest = (segment == 0 && bin < bins) ? total : 0u;
        longest = (uint32_t)wave_max_i((int)longest);           /* entry counts stay far below 2^31 */
        if (threadIdx.x == 0 && longest) atomicMax(&P.control[4], longest);
    }
    __syncthreads();
    const uint32_t start = s_part[0][column];
    if (bin >= bins || start == 0xffffffffu) return;    /* empty or overflowed bin: its entries are never written */
    uint32_t run = start + before;
    if (few) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (c0 + u < c1) col[(size_t)(c0 + u) * bins] = run;    /* count -> first entry of the chunk */
            run += kept[u];
        }
        return;
    }
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
/* what the bins' classes are worked out from and written to (bin_class_of, kernels_raster.h): the workgroups of
 * rz_bin_fill_kernel's launch from number fill_blocks on take the batch's bins, a wave each */
struct SkyBin;
struct ClassTables {
    const TriRec *tri_base, *fan_base;
    const uint2 *bin_range_base;
    SkyBin *sky_base;
    uint32_t *class_base;
    uint32_t max_bins, fill_blocks;
};
#define RZ_CLASS_SELECTED 8u
RZ_DEV void bin_class_of(const PassParams *__restrict__ passes, uint32_t pass_index, uint32_t bin_index, int lane, uint32_t *s_sel,
                         const ClassTables &tables);

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) rz_bin_fill_kernel(const PassParams *__restrict__ passes, uint32_t num_passes, const ClassTables classes) {
    if (blockIdx.x >= classes.fill_blocks) {                    /* uniform over the workgroup */
        __shared__ uint32_t s_selected[WAVES][RZ_CLASS_SELECTED];
        const uint32_t wave_of_group = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const uint32_t item = (blockIdx.x - classes.fill_blocks) * (uint32_t)WAVES + wave_of_group;
        const uint32_t pass_index = item / classes.max_bins;
        if (pass_index < num_passes)
            bin_class_of(passes, pass_index, item - pass_index * classes.max_bins, (int)(threadIdx.x & 63u), s_selected[wave_of_group], classes);
        return;
    }
    /* Workgroups start in index order: the pass is the fastest part of the index and a pass's chunks are taken
     * from its last one down.  The chunks that run longest are those of triangles over many bins, i.e. the sky
     * box, which a frame draws last (test_helper.c:497-503): they start first in every pass instead of making
     * up the launch's tail. */
    const PassParams &P = passes[blockIdx.x % num_passes];
    const uint32_t chunks = ((uint32_t)P.num_tris + 63u) / 64u, turn = blockIdx.x / num_passes;
    if (turn >= chunks) return;                                 /* uniform over the workgroup */
    const uint32_t chunk = chunks - 1u - turn;
    if (P.control[2] != 0u) return;                             /* out of scratch, or left out: the pass is run again later */
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
         * (ascending lane = submission order, as below) and write their entries side by side.  Every wave of the
         * workgroup holds the chunk: wave w takes bins w, w + WAVES, ... (a lone wave issues an instruction every few
         * cycles; in a launch of a frame or two the chunk's turns, not the device, set the kernel's length). */
        /* lane i asks for bin i's start and whether its list is kept, all bins in one memory round trip
         * (asked for bin after bin inside the loop they were one dependent round trip each) */
        uint32_t my_start = 0u, my_kept = 0u;
        if (lane < ubins) {
            const int b = (uy0 + lane / uw) * P.bins_x + ux0 + lane % uw;
            my_start = row[b];
            my_kept = P.bin_range[b].y;         /* zero: the list is empty or overflowed, nothing is kept */
        }
        for (int i = wave; i < ubins; i += WAVES) {
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
         * in one memory round trip.  Of the workgroup's waves, each of which holds the chunk, wave w takes the bins whose
         * index is w modulo WAVES: a lane's walk leaves out the bins of the other waves, and those whose lists are not
         * kept (empty or overflowed). */
        uint32_t start[9];
        uint32_t slots = 0;             /* bit q: place q of this lane's box is a bin for this wave to fill */
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const bool in_box = r.entries != 0 && q % 3 < w && q / 3 < h;
            const int b = in_box ? (r.by0 + q / 3) * P.bins_x + r.bx0 + q % 3 : 0;     /* (every lane loads: nothing waits on a branch) */
            const uint32_t first = row[b], kept = P.bin_range[b].y;
            start[q] = first;
            if (in_box && kept != 0u && (b & (WAVES - 1)) == wave) slots |= 1u << q;
        }
        for (;;) {
            /* places ascend with the bin's index (row by row): the lowest place left is the lane's next bin */
            const int place = slots ? __builtin_ctz(slots) : 0, row_of = (place * 11) >> 5, col_of = place - 3 * row_of;
            const int mine = slots ? (r.by0 + row_of) * P.bins_x + r.bx0 + col_of : 0x7fffffff;
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
                uint32_t at = start[0];
#pragma unroll
                for (int q = 1; q < 9; q++) if (place == q) at = start[q];
                fill_entries(P, at + before, ref, r.entries, r0, r1, r2, r3, slot, s_fan, fan_tri);
                slots &= slots - 1u;
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
