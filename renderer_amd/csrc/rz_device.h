/*
 * Device-side data layout of the pipeline (shared by the host-side recorder
 * in api.hip and the kernels).  Everything here lives in HBM for the duration
 * of one pass; a "pass" is a maximal run of draws into one framebuffer.
 *
 *   draws[D]        one DrawRec per draw call: family, state, uniforms in
 *                   device form (texture handles resolved to DevTex)
 *   first_tri[D+1]  exclusive prefix of the draws' face counts: triangle t of
 *                   the pass belongs to the draw d with first_tri[d] <= t
 *   clip[3T]        clip-space position per vertex            (vertex kernel)
 *   vary[3T][24]    varying floats per vertex, 96-byte stride  (vertex kernel)
 *   tri[T]          TriRec per input triangle                  (setup kernel)
 *   bbox[T]         packed screen bounding box per triangle    (setup kernel)
 *   fan_tri / fan_vary   records and varyings of clipped triangles' fans
 *   bin_*           per 64x64-pixel bin: ordered list of triangle indices
 *
 * Triangle index == submission order (draw order, then face order); every
 * later stage walks triangles in ascending index, which is what preserves the
 * reference's last-writer-wins / blend order per pixel.
 */
#ifndef RZ_DEVICE_H
#define RZ_DEVICE_H

#include <stdint.h>

/* Pointers the kernels read out of these structures point into device memory, all of them.  Typed that way
 * (address space 1) in the device pass, the compiler addresses through them with global_load / global_store
 * -- scalar base plus a 32-bit lane offset -- instead of building a 64-bit flat address per lane and
 * waiting on both memory counters.  On the host the same eight bytes are an ordinary pointer. */
#if defined(__HIP_DEVICE_COMPILE__)
#define RZ_GLOBAL __attribute__((address_space(1)))
#else
#define RZ_GLOBAL
#endif
template <class T> struct rz_gptr {
    typedef T RZ_GLOBAL *type;
    type p;
    __host__ __device__ rz_gptr &operator=(T *q) { p = (type)q; return *this; }
    __host__ __device__ operator type() const { return p; }
    __host__ __device__ type operator->() const { return p; }
    template <class U> __host__ __device__ U RZ_GLOBAL *as() const { return (U RZ_GLOBAL*)p; }     /* the same address, other element type */
};

#define RZ_VARY_STRIDE 24           /* floats per vertex in vary[] (max used: 21) */
/* Where the pbr varyings (pbr_shader.h:29-37) sit in a vertex's block: what every fragment reads in the
 * first four groups of four floats, what only shadow mapping and the layer inspector read behind them,
 * so that the other fragments fetch four groups per corner instead of six. */
enum { RZ_PV_WORLD = 0, RZ_PV_NORMAL = 3, RZ_PV_TANGENT = 6, RZ_PV_BITANGENT = 9, RZ_PV_TEXCOORD = 12,
       RZ_PV_DEPTH = 14, RZ_PV_CLIP = 17, RZ_PV_ALWAYS = 14, RZ_PV_COUNT = 21 };
#define RZ_BV_TEXCOORD 6            /* blinn_shader.h:16-21: world position, depth position, texcoord, normal */
#define RZ_MAX_POLY 10              /* corners of a triangle clipped by 7 planes */
#define RZ_MAX_FAN 8                /* RZ_MAX_POLY - 2 */
#define RZ_BIN 64                   /* bin edge in pixels */
#define RZ_QUEUE_WAYS 64u           /* parts of a shade work list (one wave reads all the lengths) */
#define RZ_ORDERED_LIST 2u          /* the third list, after the blinn and pbr families': tiles of a pass with blended draws
                                     * where a blended fragment lies over another fragment of the pass */
#define RZ_QUEUE_PAD 32u            /* counters sit one 128-byte line apart: atomics on one line serialise */
#define RZ_VISIBILITY_LIST 3u       /* the fourth: tiles of such a pass that rz_blend_visibility_kernel has to look at */
#define RZ_SKY_LIST 4u              /* the fifth: blocks with skybox fragments that came through the bins (a sky box that is not on the
                                     * pass's table: more than 12 triangles, not drawn last, fragments being counted) */
#define RZ_CROWD_LIST 5u            /* the sixth: the blocks of crowded bins, rz_crowd_kernel's */
#define RZ_LISTS 6u
/* behind the lists' lengths in queue_counts, one word per batch: set when a pass of the batch has a fragment that a sky
 * box on its table may hide (see RZ_SKY_NEAR) */
#define RZ_BATCH_CONTESTED (RZ_LISTS * RZ_QUEUE_WAYS)
/* A sky box lies at z = w * (1 - EPSILON) (skybox_shader.c:15), short of the far plane: depth 0.999995 give or take
 * rounding, wherever it is seen from.  Drawn last it hides what lies behind it (graphics.c:523), and a fragment nearer
 * than this can not be hidden by it. */
#define RZ_SKY_NEAR 0.9999f
/* the table of a pass's sky boxes: up to RZ_SKY_DRAWS draws of up to RZ_SKY_FACES triangles, RZ_MAX_FAN entries each -- the
 * triangle's index, RZ_REF_FAN | its fan records', or RZ_REF_NONE --, in submission order */
#define RZ_SKY_DRAWS 4
#define RZ_SKY_FACES 12
#define RZ_SKY_TABLE (RZ_SKY_DRAWS * RZ_SKY_FACES * RZ_MAX_FAN)
#define RZ_TILE_W 32                /* raster / shade workgroup tile: 4 waves of 8x8 side by side, */
#define RZ_TILE_H 8                 /* so a tile row is one 128-byte line of a 4-byte plane */

enum { RZ_FMT_RGBA32F = 0, RZ_FMT_R32F = 1 };
/* texel formats: float4 unless one of the format bits is set.  The packed
 * formats are used only for images they hold without loss:
 *   R32F    one float, read as (d, d, d, 1)                    shadow maps
 *   UNORM8  4 bytes, channel = byte / 255.0f                   8-bit data maps
 *   SRGB8   4 bytes, rgb = (float)pow(byte / 255.0f, 2.2),     8-bit colour maps
 *           alpha = byte / 255.0f
 *   RGBE8   3 mantissa bytes + shared exponent byte e,         Radiance .hdr maps
 *           rgb = (float)mantissa * 2^(e - 136), alpha = 1 */
enum { RZ_TEX_TILED = 1, RZ_TEX_R32F = 2, RZ_TEX_UNORM8 = 4, RZ_TEX_SRGB8 = 8, RZ_TEX_RGBE8 = 16,
       /* not an image format but a plane derived from one: per texel the three bytes draw_fragment writes for
        * its colour (saturate, * 255, truncate: graphics.c:452-468, maths.c:33), for skybox faces */
       RZ_TEX_DISPLAY8 = 128,
       RZ_TEX_LAYOUT = RZ_TEX_TILED };

/* a sampled image; data == nullptr means "no map bound" */
struct DevTex {
    rz_gptr<const void> data;
    int32_t w, h;
    int32_t flags;                  /* RZ_TEX_* */
    int32_t reserved;
};

struct DevIbl {                     /* pbr_shader.h:13-18 in device form */
    int32_t mip_levels;
    int32_t pad;
    DevTex lut;
    DevTex diffuse[6];
    DevTex specular[15][6];
};

enum {
    RZ_DRAW_DOUBLE_SIDED = 1,
    RZ_DRAW_BLEND = 2,
    RZ_DRAW_SHADOW_PASS = 4,
    RZ_DRAW_SKINNED = 8,
    RZ_DRAW_UNIFORM_CUBE = 16,      /* skybox: the six faces have one size and format */
    RZ_DRAW_NARROW_TEXELS = 32,     /* blinn / pbr: every image the draw samples has a 4-byte format (or is absent) */
    RZ_DRAW_SKY_BYTES = 64          /* skybox: maps[] are the faces' RZ_TEX_DISPLAY8 planes */
};

struct DrawRec {
    int32_t family;                 /* rz_family */
    int32_t flags;                  /* RZ_DRAW_* */
    int32_t num_varyings;           /* floats: 11 blinn, 21 pbr, 3 skybox */
    int32_t num_faces;
    rz_gptr<const void> vertices;         /* rz_vertex[3 * num_faces] on the device */
    rz_gptr<const float> joints;          /* mat4[] for skinning or nullptr */
    rz_gptr<const float> joint_normals;    /* mat3[] or nullptr */
    /* uniforms common to blinn and pbr (blinn_shader.h:23-34, pbr_shader.h:39-50);
     * the skybox keeps its vp matrix in camera_vp */
    float light_dir[3];
    float camera_pos[3];
    float model[16];
    float normal[9];
    float light_vp[16];
    float camera_vp[16];
    float ambient, punctual;
    DevTex shadow_map;
    /* surface: blinn basecolor / pbr basecolor_factor or diffuse_factor */
    float base[4];
    float shininess;                /* blinn */
    float metalness, roughness;     /* pbr metal-rough factors */
    float specular[3];              /* pbr spec-gloss */
    float glossiness;
    /* maps: blinn {diffuse, specular, emission};
     * pbr {basecolor|diffuse, metalness|specular, roughness|glossiness,
     *      normal, occlusion, emission} */
    DevTex maps[6];
    /* the maps' texels side by side, texel i of map k at word i * W + k (W = 4 blinn, 8 pbr; an absent map's word is 0), when
     * they have one size, one layout and 4-byte formats: a fragment's material texels are then neighbours in memory, one
     * load of one cache line where the maps' own images are a line each; nullptr: sample the maps */
    rz_gptr<const uint32_t> bundle;
    rz_gptr<const DevIbl> ibl;
    rz_gptr<const DevTex> skybox;         /* 6 faces (the same six are in maps[] of a skybox draw) */
    int32_t workflow;               /* 0 metal-rough, 1 spec-gloss */
    float alpha_cutoff;
    int32_t layer_view;
    /* a sky box drawn last in its pass (at most 12 triangles, no blending): slot + 1 of its triangles in the pass's
     * table of sky triangles -- they are kept out of the bins --; 0: an ordinary draw */
    int32_t sky_slot;
};

/* One rasterisable triangle: what rasterize_triangle (graphics.c:472-535)
 * derives before its pixel loop.  For an input triangle that had to be
 * clipped the main record only points at its fan. */
struct TriRec {
    float ax, ay;                   /* screen position of corner a */
    float abx, aby, acx, acy;       /* edges b-a and c-a */
    float factor;                   /* 1 / (ab x ac) */
    float z[3];                     /* screen depth per corner */
    float rw[3];                    /* 1 / clip.w per corner */
    uint32_t meta;                  /* draw | backface << 16 | clipped << 17 | fan_count << 18
                                     * | blend << 22 | alpha-cutoff << 23 | family << 24 */
    uint32_t bx, by;                /* own bounding box: min | max << 16 */
};

/* One bin-list entry: the triangle's index and a copy of its record, so the
 * rasteriser needs a single memory round trip per 64 candidates. */
struct BinEntry {
    uint32_t tri;
    uint32_t pad[3];
    TriRec rec;
};

#define RZ_META_DRAW(m) ((m) & 0xffffu)
#define RZ_META_BACKFACE(m) (((m) >> 16) & 1u)
#define RZ_META_CLIPPED(m) (((m) >> 17) & 1u)
#define RZ_META_FAN(m) (((m) >> 18) & 15u)
#define RZ_META_BLEND(m) (((m) >> 22) & 1u)
#define RZ_META_CUTOFF(m) (((m) >> 23) & 1u)
#define RZ_META_FAMILY(m) (((m) >> 24) & 3u)
#define RZ_META_BLACK(m) (((m) >> 26) & 1u)     /* a shadow pass's fragment without blending: (0, 0, 0) whatever its inputs (blinn_shader.c:101-115, pbr_shader.c:113-136) */
#define RZ_BBOX_EMPTY_X 0x0000ffffu  /* min 65535 > max 0 */

struct PassParams {
    /* target */
    rz_gptr<uint32_t> color;                /* RGBA8, row 0 = bottom */
    rz_gptr<float> depth;
    int32_t width, height;
    int32_t clear_color_pending, clear_depth_pending;
    uint32_t clear_color;
    float clear_depth;
    /* geometry */
    rz_gptr<const DrawRec> draws;
    rz_gptr<const uint32_t> first_tri;      /* num_draws + 1 */
    int32_t num_draws, num_tris;
    rz_gptr<float> clip;                    /* float4 per vertex */
    rz_gptr<float> vary;
    rz_gptr<TriRec> tri;
    rz_gptr<uint2> bbox;
    rz_gptr<TriRec> fan_tri;                /* fan_capacity * RZ_MAX_FAN */
    rz_gptr<float> fan_vary;                /* fan_capacity * RZ_MAX_POLY * RZ_VARY_STRIDE */
    int32_t fan_capacity;
    rz_gptr<uint32_t> clip_list;            /* triangles queued for the clipper, each with its draw: pairs of words (count in control[3]) */
    /* binning */
    int32_t bins_x, bins_y;
    rz_gptr<uint32_t> bin_counts;           /* [chunks of 64 triangles][bins]: entries, then first entry */
    rz_gptr<uint2> bin_range;               /* per bin: {first entry, entries} */
    rz_gptr<BinEntry> bin_list;
    rz_gptr<uint2> bin_boxes;               /* per list entry: x0 | x1 << 16, y0 | y1 << 16 | alpha-cutoff << 31 -- all that walking a list reads */
    uint32_t bin_capacity;
    rz_gptr<uint32_t> ref;                  /* per pixel: reference of the visible fragment (split passes) */
    int32_t sky_draws;              /* draws of this pass whose triangles are on its table of sky triangles */
    /* scalars written by the device: [0] fan blocks used, [1] bin entries
     * used, [2] error bits, [3] triangles queued for clipping */
    rz_gptr<uint32_t> control;
    rz_gptr<unsigned long long> counters;   /* nullptr unless counting is enabled */
    int32_t crowd_min;              /* bins with at least this many list entries are rz_crowd_kernel's (0: none are) */
    int32_t micro_area;             /* triangles whose box covers at most this many pixels of a wave's block
                                     * are rasterised one per lane; 0 = none */
    int32_t micro_min_list;         /* ... in bins whose list has at least this many entries */
    /* A launch covers a batch of passes (blockIdx.y = pass).  The rasteriser
     * and the shade kernels address the batch-wide tables through restrict
     * kernel arguments plus these per-pass offsets. */
    int32_t tiles_x, tiles_y;       /* 32x8 tiles of this framebuffer */
    /* per batch, for the blinn and pbr families: a work list of tiles (pass << 20 | tile y << 9 |
     * tile x) in RZ_QUEUE_WAYS parts of queue_capacity entries; lengths in
     * queue_counts[(family * RZ_QUEUE_WAYS + part) * RZ_QUEUE_PAD]; tile_listed[tile of this pass]:
     * bit f set = the tile is on family f's list */
    rz_gptr<uint32_t> tile_listed;
    rz_gptr<uint32_t> queue;
    rz_gptr<uint32_t> queue_counts;
    uint32_t queue_capacity;
    uint32_t tile_off;              /* tiles before this pass in the batch */
    uint32_t tri_off;               /* triangles before this pass (vertices = 3x) */
    uint32_t fan_off;               /* fan blocks before this pass */
    uint32_t list_off;              /* bin-list entries before this pass */
    uint32_t bin_off;               /* bins before this pass */
    uint64_t ref_off;               /* pixels before this pass in the reference plane */
    uint64_t draws_off;             /* byte offset of draws[] in the flush's upload region */
    /* Scratch that turns out too small.  The clipper's blocks and the bin lists are sized by an
     * estimate; a pass that needs more sets its error bits and one word shared by the whole flush,
     * touches nothing of its framebuffer (every kernel behind the binning looks at control[2]
     * first), and goes on counting what it needs.  A pass that may depend on another pass of the
     * flush (level > 0) and finds the flush's word set does not run either (RZ_ERR_SKIPPED).  The
     * host runs those passes again, in order, with scratch of the sizes they reported. */
    rz_gptr<uint32_t> flush_flag;
    int32_t level;                  /* 0: depends on no other pass of its flush */
    /* a pass with blended draws: the list (RZ_ORDERED_LIST or RZ_VISIBILITY_LIST) rz_bin_scan_kernel puts the tiles of
     * its bins on that are not empty -- the kernel behind it then starts no wave for the rest of the frame; -1: none */
    int32_t scan_list;
    rz_gptr<uint32_t> sky_table;            /* RZ_SKY_TABLE entries (above); filled by the vertex and clip kernels */
};

enum { RZ_ERR_FAN_OVERFLOW = 1, RZ_ERR_BIN_OVERFLOW = 2, RZ_ERR_QUEUE_OVERFLOW = 4, RZ_ERR_SKIPPED = 8 };

#endif
