/*
 * Host side of the MI355X triangle pipeline: the C ABI of include/mi355r.h.
 *
 * The reference API is immediate-mode and synchronous (graphics.c); here draw
 * calls are RECORDED -- the program's uniform block is snapshotted and its
 * host pointers (textures, joint matrices, IBL data) are translated to device
 * handles -- and executed in submission order at flush points:
 *   framebuffer_sync / framebuffer_read*, texture_from_{depth,color}buffer,
 *   a clear of a framebuffer that has draws pending, graphics_flush, releases.
 * A flush turns every maximal run of draws into one framebuffer into a "pass"
 * and launches vertex -> setup -> bin -> raster for it on the library's HIP
 * stream; nothing waits for the GPU except the calls that hand pixels back to
 * the host.  There is no CPU rendering path: without a GPU the first call
 * aborts with the HIP error.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <chrono>
#include <vector>

#include "../../include/mi355r.h"
#include "kernels.h"

static_assert(sizeof(rz_vertex) == 80, "vertex stream layout (mesh.h:8-15)");
static_assert(sizeof(rz_blinn_uniforms) == 344, "blinn uniform block (blinn_shader.h:23-44)");
static_assert(sizeof(rz_pbr_uniforms) == 440, "pbr uniform block (pbr_shader.h:39-76)");
static_assert(sizeof(rz_skybox_uniforms) == 72, "skybox uniform block (skybox_shader.h:16-19)");
static_assert(sizeof(rz_blinn_attribs) == 64, "blinn attribs");
static_assert(sizeof(TriRec) == 64, "TriRec is four float4");
static_assert(sizeof(BinEntry) == 80, "BinEntry is five float4");
static_assert(sizeof(DevTex) == 24, "DevTex");

/* The message goes to stderr and, when MI355R_FATAL_LOG names a file, to that file as well: a test
 * runner that captures stderr loses what it captured when the process aborts. */
[[noreturn]] static void fatal(const char *fmt, ...) {
    char text[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof text, fmt, ap);
    va_end(ap);
    fprintf(stderr, "mi355r: fatal: %s\n", text);
    fflush(stderr);
    if (const char *path = getenv("MI355R_FATAL_LOG")) {
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "mi355r: fatal: %s\n", text);
            fclose(f);
        }
    }
    abort();
}

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    fatal("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

/* ---------------------------------------------------------------- objects -- */

struct Fb {
    rz_framebuffer pub;             /* must stay first: handed out as rz_framebuffer* */
    uint32_t *d_color;
    float *d_depth;
    int lane;                       /* execution lane of the last pass into it, -1 = none yet */
    uint32_t longest_list = 0;      /* entries of the longest bin list among its passes of the last settled flush */
    bool clear_color_pending, clear_depth_pending;
    uint32_t clear_color;
    float clear_depth;
};

struct Tex {
    rz_texture pub;                 /* must stay first */
    void *d_data;
    size_t d_bytes;
    int flags;                      /* RZ_TEX_* of the device image */
    bool device_stale;              /* host buffer is newer */
    bool host_stale;                /* device image is newer */
    bool host_pinned = false;       /* pub.buffer is page-locked (textures the GPU writes: their read-back is a plain DMA) */
    bool host_from_runtime = false; /* ... because it was allocated with hipHostMalloc (freed with hipHostFree) */
    /* a texture written on the GPU (texture_from_*buffer) orders the passes
     * around it: readers wait for the write, the next write for the readers */
    int write_lane;
    hipEvent_t write_event;
    std::vector<hipEvent_t> read_events;
    /* a copy from a framebuffer that is recorded but not launched yet will fill the image; and
     * the flush (Context::flush_epoch) during whose recording a draw last referred to it */
    unsigned long long image_version = 0;   /* counts the times the device image was (about to be) rewritten */
    bool pending_write = false;
    unsigned long long used_epoch = ~0ull;
    unsigned long long write_seq = 0;       /* order of the writes on a lane: waiting for a later one covers the earlier ones */
};

struct CubeDisplay {
    uint32_t *d = nullptr;                  /* six faces of face_texels words each */
    size_t face_texels = 0;
    unsigned long long version[6] = {};     /* Tex::image_version of the faces it was made from */
};

struct rz_program {
    int family;
    int attrib_bytes, varying_bytes, uniform_bytes;
    int double_sided, blend;
    void *attribs[3];
    void *uniforms;
};

struct Mesh {
    void *d_vertices;
    int num_faces;
    int num_joints;                 /* 1 + largest joint index referenced */
};

struct rz_pose_source {
    uint32_t *d_block;              /* the skeleton on the device (kernels.h, K0) */
    int num_joints;
    float max_time;
};

struct Pending {
    /* pose evaluated on the device (graphics_draw_mesh_posed): source, wrapped time, attached joint or -1 */
    rz_pose_source *pose = nullptr;
    float pose_time = 0;
    int pose_attached = -1;
    size_t pose_job = 0;            /* its job of the flush (graphics_flush) */
    Fb *fb;
    rz_program *program;
    DrawRec rec;                    /* joints / vertices patched at flush */
    int num_joints;
    size_t joints_at, joint_normals_at;     /* offsets into Context::side */
    std::vector<Tex*> gpu_textures; /* sampled textures that a GPU pass wrote */
    std::vector<Tex*> pending_textures;     /* sampled textures that a recorded copy of this flush will fill */
    bool transient;                 /* vertices recorded by graphics_draw_triangle ... */
    size_t verts_at = 0, verts_count = 0;   /* ... in Context::transient_verts, unless ... */
    struct Replay *replay = nullptr;        /* ... they repeat, so far, the stream remembered for the program */
    std::vector<char> uniform_bytes;        /* for merging per-triangle calls */
};

/* A host that submits face by face (the reference's draw_model loops) sends the same vertices frame after
 * frame.  The stream a program's draw was made of is remembered -- on the host to compare against, on the
 * device to draw from -- and a later draw of that program whose triangles repeat it, in full or a prefix of
 * it, is drawn from the device's copy: nothing is stored per triangle and nothing is uploaded per frame.  The
 * first triangle that differs turns the draw back into an ordinary recorded one. */
struct Replay {
    std::vector<rz_vertex> host;
    void *d_vertices = nullptr;
    size_t faces = 0;
    int misses = 0;                         /* streams remembered in a row that no later draw repeated */
    unsigned long long made_epoch = ~0ull;  /* the flush that (re)made it */
};

template <typename T> struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;
};

/* An execution lane: a HIP stream with its own per-pass scratch.  Passes that
 * do not depend on each other (different framebuffers, no texture hand-over)
 * run on different lanes and overlap on the GPU; passes on one lane are
 * serialised by the stream, which is what lets them share the scratch. */
struct Lane {
    hipStream_t stream = nullptr;
    DevBuf<float> clip, vary, fan_vary;
    DevBuf<TriRec> tri, fan_tri;
    DevBuf<uint2> bbox;
    DevBuf<uint32_t> clip_list, ref;
    DevBuf<uint2> bin_range;
    DevBuf<SkyBin> sky_bins;            /* per bin of the batch: rz_bin_class_kernel's record */
    DevBuf<uint32_t> sky_table;         /* per pass of the batch: its table of sky triangles */
    DevBuf<uint32_t> bin_class;         /* per bin of the batch: what rz_strip_kernel has to do with it */
    DevBuf<uint32_t> bin_counts, queue, queue_counts, tile_listed;
    DevBuf<BinEntry> bin_list;
    DevBuf<uint2> bin_boxes;
};

static const int RZ_TEX_NARROW_FORMATS = RZ_TEX_R32F | RZ_TEX_UNORM8 | RZ_TEX_SRGB8 | RZ_TEX_RGBE8;
static const size_t MAX_BATCH = 64;     /* passes per launch sequence, upper bound */
static const int MAX_LANES = 8;

struct Context {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;       /* lane 0's stream: uploads and utilities */
    hipStream_t copy_stream = nullptr;  /* table uploads / control read-back of a flush */
    Lane lanes[MAX_LANES];
    int num_lanes = 1, next_lane = 0;
    size_t max_batch = 32;                  /* passes per launch sequence */
    bool sky_bytes = true;                  /* MI355R_SKY_BYTES=0: skybox fragments fetch the face's texel and convert it themselves */
    bool pack_textures = true;              /* MI355R_PACK=0: keep every image as float4 */
    int micro_min_list = 256;               /* MI355R_MICRO_MIN_LIST: small triangles are rasterised one per lane only in bins with at least this many list entries */
    bool sky_table = true;                  /* MI355R_SKY_TABLE=0: a sky box goes through the bins like any other draw */
    int fill_waves = 0;                     /* MI355R_FILL_WAVES: waves that share a chunk of 64 triangles in rz_bin_fill_kernel (1, 2, 4; unset: by the launch's size) */
    unsigned strip_long = 2048;             /* MI355R_STRIP_LONG: a batch whose framebuffers had a bin list of this many entries in their last frame takes one block per wave,
                                             * of half as many two; 0, or MI355R_STRIP set = always `strip` */
    int strip = 2;                          /* MI355R_STRIP: 16x8 blocks a wave of rz_strip_kernel takes (8, 4, 2, 1) */
    /* passes with blended draws: 0 = whole through the ordered kernel, 1 = visibility first and the ordered kernel only for
     * the tiles that need it, 2 = the latter for large framebuffers (MI355R_SPLIT_BLEND=0 / 1; measured: 4K frames gain 8 %,
     * 1080p frames lose as much to the three extra launches) */
    int split_blend = 2;
    bool verbose = false;                   /* MI355R_VERBOSE */
    bool record_copies = true;              /* MI355R_RECORD_COPIES=0: texture_from_*buffer always flushes first */
    int micro_area = 32;                    /* MI355R_MICRO_AREA: largest box (pixels of a block) rasterised one triangle per lane; 0 = off */
    int shade_workgroups = 4096;            /* 256-thread workgroups per shade launch (compute units x 16) */
    std::vector<hipEvent_t> events;     /* recorded since the last full sync */
    std::vector<Tex*> current_gpu_textures, current_pending_textures;
    /* framebuffer -> texture copies recorded between draws: run behind the pass that ends at draw `after` */
    struct TexWrite { Tex *tex; Fb *fb; bool depth; size_t after; };
    std::vector<TexWrite> tex_writes;
    unsigned long long flush_epoch = 0;     /* counts the flushes that had something to submit */
    unsigned long long write_seq = 0;
    /* upload ring: pinned host memory mirrored 1:1 in device memory */
    char *ring_host = nullptr, *ring_dev = nullptr;
    size_t ring_size = 0, ring_at = 0;
    /* scratch estimates (scratch_entries): what passes that ran out turned out to need, and values forced for tests */
    size_t fan_highwater = 0, list_highwater = 0;
    size_t forced_fan_estimate = 0, forced_list_estimate = 0;
    unsigned long long passes_run_again = 0;
    unsigned long long replayed_draws = 0;  /* draws recorded triangle by triangle that were drawn from a remembered stream */
    /* recording */
    std::vector<Pending> pending;
    std::vector<char> side;                 /* joint matrices snapshotted at record time */
    std::vector<rz_vertex> transient_verts; /* vertices recorded triangle by triangle, draw after draw; emptied with the list */
    std::unordered_map<rz_program*, Replay> replays;
    bool replay_streams = true;             /* MI355R_REPLAY=0: every recorded triangle is stored and uploaded */
    Fb *checked_fb = nullptr;               /* the framebuffer graphics_draw_triangle looked up last */
    /* resources */
    std::unordered_set<Fb*> fbs;
    std::unordered_set<Tex*> texs;
    std::unordered_map<const void*, Mesh> meshes;
    std::unordered_map<const void*, DevTex*> cubes;     /* rz_cubemap* -> DevTex[6] */
    std::unordered_map<const void*, CubeDisplay> cube_displays;     /* rz_cubemap* -> its plane of display bytes */
    std::unordered_map<const void*, DevIbl*> ibls;      /* rz_ibldata* -> DevIbl */
    std::unordered_map<const void*, bool> ibl_narrow;   /* ... and whether all its images have 4-byte formats */
    std::unordered_set<rz_pose_source*> pose_sources;
    /* joint and normal matrices of the poses a flush evaluates on the device; two pools taken in turn, so
     * that the flush in flight keeps its poses while the next one is put together */
    DevBuf<char> pose_pool[2];
    /* framebuffer_blit: the flipped image on the device and its page-locked landing place on the host */
    DevBuf<uint32_t> blit_staging;
    uint32_t *blit_host = nullptr;
    size_t blit_host_cap = 0;
    struct Registered { void *vs, *fs; int family; };
    std::vector<Registered> families;
    /* measurement */
    bool timing = false;
    std::vector<hipEvent_t> timing_events;  /* 8 per launch sequence: before vertex, after each of the 7 timed kernels */
    std::vector<unsigned> timing_passes;    /* passes per timed launch sequence */
    double kernel_ms[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long timed_sequences = 0;
    unsigned long long kernel_launches = 0;
    bool counting = false;
    unsigned long long *d_counters = nullptr;
    unsigned long long host_counters[8] = {0};
    char last_raster_kernel[64] = "";       /* the rasterising kernel of the last batch launched (measurement: which symbol to look up) */
    /* the pass mi355r_inspect_* looks at: the last one submitted */
    PassParams last_pass;
    bool have_last_pass = false;
    int last_pass_sky_first = 0;            /* its first triangle that is kept on the table of sky triangles (= its triangle count: none) */
};

static Context g;

static const size_t RING_BYTES = 64u << 20;

static void ensure_ready() {
    if (!g.ready) mi355r_init(-1);
}

template <typename T> static void reserve(DevBuf<T> &b, size_t n, hipStream_t user) {
    if (n <= b.cap) return;
    if (b.ptr) { HIP_OK(hipStreamSynchronize(user)); HIP_OK(hipFree(b.ptr)); }
    size_t cap = std::max(n, b.cap + b.cap / 2);
    HIP_OK(hipMalloc((void**)&b.ptr, cap * sizeof(T)));
    b.cap = cap;
}

static void check_passes_in_flight();
static void init_packed_formats();
static float4 *scratch_on_device(int device);

static void sync_stream();

/* events live until the next full synchronisation; graphics_flush bounds their number before it creates its first */
static hipEvent_t new_event(hipStream_t on) {
    hipEvent_t e;
    HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e, on));
    g.events.push_back(e);
    return e;
}

/* full synchronisation: every lane and the copy stream */
static void sync_stream() {
    for (int i = 0; i < g.num_lanes; i++) HIP_OK(hipStreamSynchronize(g.lanes[i].stream));
    HIP_OK(hipStreamSynchronize(g.copy_stream));
    check_passes_in_flight();
    for (hipEvent_t e : g.events) HIP_OK(hipEventDestroy(e));
    g.events.clear();
    for (Tex *t : g.texs) { t->write_event = nullptr; t->read_events.clear(); }
}

extern "C" int mi355r_init(int device) {
    if (g.ready) return g.device;
    if (device >= 0) HIP_OK(hipSetDevice(device));
    HIP_OK(hipGetDevice(&g.device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, g.device));
    g.shade_workgroups = std::max(1, prop.multiProcessorCount) * 16;
    if (const char *env = getenv("MI355R_SHADE_WGS")) g.shade_workgroups = std::max(1, atoi(env));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fprintf(stderr, "mi355r: warning: built for gfx950, running on %s\n", prop.gcnArchName);
    {
        const char *env = getenv("MI355R_LANES");
        g.num_lanes = env ? atoi(env) : 4;
        if (g.num_lanes < 1) g.num_lanes = 1;
        if (g.num_lanes > MAX_LANES) g.num_lanes = MAX_LANES;
        env = getenv("MI355R_STRIP");
        if (env) { const int v = atoi(env); if (v == 8 || v == 4 || v == 2 || v == 1) { g.strip = v; g.strip_long = 0; } }
        env = getenv("MI355R_SKY_BYTES");
        if (env) g.sky_bytes = atoi(env) != 0;
        env = getenv("MI355R_SKY_TABLE");
        if (env) g.sky_table = atoi(env) != 0;
        env = getenv("MI355R_FILL_WAVES");
        if (env) { const int v = atoi(env); if (v == 1 || v == 2 || v == 4) g.fill_waves = v; }
        env = getenv("MI355R_STRIP_LONG");
        if (env) g.strip_long = (unsigned)atoi(env);
        env = getenv("MI355R_SPLIT_BLEND");
        if (env) g.split_blend = atoi(env) != 0 ? 1 : 0;
        g.verbose = getenv("MI355R_VERBOSE") != nullptr;
        env = getenv("MI355R_RECORD_COPIES");
        if (env) g.record_copies = atoi(env) != 0;
        env = getenv("MI355R_REPLAY");
        if (env) g.replay_streams = atoi(env) != 0;
        env = getenv("MI355R_MICRO_AREA");
        if (env) g.micro_area = std::max(0, std::min(atoi(env), 128));
        env = getenv("MI355R_MICRO_MIN_LIST");
        if (env) g.micro_min_list = std::max(0, atoi(env));
        env = getenv("MI355R_PACK");
        if (env) g.pack_textures = atoi(env) != 0;
        env = getenv("MI355R_BATCH");
        if (env && atoi(env) >= 1) g.max_batch = std::min<size_t>((size_t)atoi(env), MAX_BATCH);
    }
    for (int i = 0; i < g.num_lanes; i++) HIP_OK(hipStreamCreateWithFlags(&g.lanes[i].stream, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
    g.stream = g.lanes[0].stream;
    HIP_OK(hipHostMalloc((void**)&g.ring_host, RING_BYTES, hipHostMallocDefault));
    HIP_OK(hipMalloc((void**)&g.ring_dev, RING_BYTES));
    g.ring_size = RING_BYTES;
    {
        /* tables of d_srgb (shade.h), from the host's pow */
        static SrgbTables tables;
        for (int e = -149; e <= 0; e++) tables.pow2[e + 149] = pow(ldexp(1.0, e), 1 / 2.2);
        for (int i = 0; i < 32; i++) {
            tables.m0[i] = pow(1.0 + i / 32.0, 1 / 2.2);
            tables.inv[i] = 1.0 / (1.0 + i / 32.0);
        }
        /* ... and the steps of d_srgb_for_bytes: the byte of v as the reference computes it (maths.c:41, :33),
         * bisected over the positive floats, which are ordered like their bit patterns */
        auto byte_of_linear = [](float v) { return (int)(unsigned char)((float)pow((double)v, 1 / 2.2) * 255); };
        tables.step[0] = 0.0f;
        for (int k = 1; k < 256; k++) {
            uint32_t lo = 1u, hi = 0x3f800000u;             /* byte(lo) = 0 < k <= 255 = byte(1.0f) */
            while (hi - lo > 1u) {
                const uint32_t mid = lo + (hi - lo) / 2u;
                float v;
                memcpy(&v, &mid, sizeof v);
                if (byte_of_linear(v) >= k) hi = mid; else lo = mid;
            }
            memcpy(&tables.step[k], &hi, sizeof(float));
        }
        {   /* the step function is the reference's: checked on a spread of values and around every step */
            uint32_t state = 12345u;
            for (int i = 0; i < 20000; i++) {
                state = state * 1664525u + 1013904223u;
                uint32_t bits = i < 255 * 16 ? 0u : 0x30000000u + state % (0x3f800000u - 0x30000000u);
                if (i < 255 * 16) { memcpy(&bits, &tables.step[1 + i / 16], 4); bits += (uint32_t)(i % 16) - 8u; }
                float v;
                memcpy(&v, &bits, sizeof v);
                if (!(v > 0.0f && v <= 1.0f)) continue;
                const int k = (int)(std::upper_bound(tables.step + 1, tables.step + 256, v) - (tables.step + 1));
                if (k != byte_of_linear(v)) fatal("the display transfer is not the step function its table says (v = %a)", (double)v);
            }
        }
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(rz_srgb_tables), &tables, sizeof tables));
    }
    init_packed_formats();
    (void)scratch_on_device(g.device);      /* texture_sync_noalloc's staging array: its address on this device, while it is current */
    HIP_OK(hipMalloc((void**)&g.d_counters, 12 * sizeof(unsigned long long)));
    HIP_OK(hipMemset(g.d_counters, 0, 12 * sizeof(unsigned long long)));
    HIP_OK(hipDeviceSynchronize());
    g.ready = true;
    return g.device;
}

extern "C" const char *mi355r_backend(void) { return "hip:gfx950"; }
extern "C" void *mi355r_stream(void) { ensure_ready(); return (void*)g.stream; }

/* ------------------------------------------------------------ framebuffers -- */

static Fb *as_fb(rz_framebuffer *p) {
    Fb *fb = (Fb*)p;
    if (!p || !g.fbs.count(fb)) fatal("framebuffer %p was not created by this library", (void*)p);
    return fb;
}

static uint32_t pack_clear(rz_vec4 c) {                 /* float_to_uchar per channel, graphics.c:42-45 */
    auto u8 = [](float v) { return (uint32_t)(unsigned char)(v * 255); };
    return u8(c.x) | (u8(c.y) << 8) | (u8(c.z) << 16) | (u8(c.w) << 24);
}

static bool has_pending_for(Fb *fb) {
    for (auto &p : g.pending) if (p.fb == fb) return true;
    return false;
}

extern "C" rz_framebuffer *framebuffer_create(int width, int height) {
    ensure_ready();
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384) fatal("framebuffer size %dx%d (at most 16384x16384)", width, height);
    Fb *fb = new Fb();
    size_t n = (size_t)width * height;
    fb->pub.width = width;
    fb->pub.height = height;
    /* the host mirrors are page-locked: framebuffer_sync then copies at the bus rate (pageable memory is staged, 3 to 4 times slower) */
    HIP_OK(hipHostMalloc((void**)&fb->pub.color_buffer, n * 4, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&fb->pub.depth_buffer, n * sizeof(float), hipHostMallocDefault));
    HIP_OK(hipMalloc((void**)&fb->d_color, n * 4));
    HIP_OK(hipMalloc((void**)&fb->d_depth, n * sizeof(float)));
    fb->lane = -1;
    g.fbs.insert(fb);
    rz_vec4 black = {0, 0, 0, 1};
    framebuffer_clear_color(&fb->pub, black);           /* graphics.c:26-27 */
    framebuffer_clear_depth(&fb->pub, 1);
    return &fb->pub;
}

extern "C" void framebuffers_read_device_rgb_end(void);

extern "C" void framebuffer_release(rz_framebuffer *p) {
    Fb *fb = as_fb(p);
    framebuffers_read_device_rgb_end();
    graphics_flush();
    sync_stream();
    HIP_OK(hipFree(fb->d_color));
    HIP_OK(hipFree(fb->d_depth));
    HIP_OK(hipHostFree(fb->pub.color_buffer));
    HIP_OK(hipHostFree(fb->pub.depth_buffer));
    g.fbs.erase(fb);
    g.checked_fb = nullptr;
    delete fb;
}

extern "C" void framebuffer_clear_color(rz_framebuffer *p, rz_vec4 color) {
    Fb *fb = as_fb(p);
    if (has_pending_for(fb)) graphics_flush();
    fb->clear_color_pending = true;
    fb->clear_color = pack_clear(color);
}

extern "C" void framebuffer_clear_depth(rz_framebuffer *p, float depth) {
    Fb *fb = as_fb(p);
    if (has_pending_for(fb)) graphics_flush();
    fb->clear_depth_pending = true;
    fb->clear_depth = depth;
}

/* a clear that no pass consumed has to be executed on its own */
static void settle_before_touching(Fb *fb);

static void apply_pending_clear(Fb *fb) {
    if (!fb->clear_color_pending && !fb->clear_depth_pending) return;
    settle_before_touching(fb);     /* a pass of the flush in flight that has to run again must not come after this clear */
    size_t n = (size_t)fb->pub.width * fb->pub.height;
    if (fb->lane < 0) fb->lane = 0;
    rz_clear_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.lanes[fb->lane].stream>>>(
        fb->d_color, fb->d_depth, n, fb->clear_color_pending, fb->clear_color,
        fb->clear_depth_pending, fb->clear_depth);
    HIP_OK(hipGetLastError());
    fb->clear_color_pending = fb->clear_depth_pending = false;
}

extern "C" void framebuffer_sync(rz_framebuffer *p) {
    Fb *fb = as_fb(p);
    framebuffer_read(p, fb->pub.color_buffer, fb->pub.depth_buffer);
}

extern "C" void framebuffer_sync_color(rz_framebuffer *p) {
    Fb *fb = as_fb(p);
    framebuffer_read(p, fb->pub.color_buffer, nullptr);
}

static void settle_framebuffer(Fb *fb);

/* Only this framebuffer's lane has to drain: other frames keep rendering.  The control blocks of its
 * passes come back on that lane behind them; should one of them say that the pass did not run (out of
 * scratch), the flush is settled -- that pass and what depends on it run again -- before the pixels
 * are handed out. */
static void read_framebuffer(Fb *fb, void *color, void *depth, hipMemcpyKind kind) {
    graphics_flush();
    HIP_OK(hipStreamSynchronize(g.lanes[fb->lane < 0 ? 0 : fb->lane].stream));
    settle_framebuffer(fb);
    apply_pending_clear(fb);
    size_t n = (size_t)fb->pub.width * fb->pub.height;
    hipStream_t st = g.lanes[fb->lane < 0 ? 0 : fb->lane].stream;
    if (color) HIP_OK(hipMemcpyAsync(color, fb->d_color, n * 4, kind, st));
    if (depth) HIP_OK(hipMemcpyAsync(depth, fb->d_depth, n * sizeof(float), kind, st));
    HIP_OK(hipStreamSynchronize(st));
}

extern "C" void framebuffer_read_device_rgb(rz_framebuffer *p, void *rgb) {
    Fb *fb = as_fb(p);
    if (!rgb || ((uintptr_t)rgb & 3u)) fatal("framebuffer_read_device_rgb: the destination must be a 4-byte aligned device pointer");
    graphics_flush();
    HIP_OK(hipStreamSynchronize(g.lanes[fb->lane < 0 ? 0 : fb->lane].stream));
    settle_framebuffer(fb);
    apply_pending_clear(fb);
    const size_t n = (size_t)fb->pub.width * fb->pub.height;
    hipStream_t st = g.lanes[fb->lane < 0 ? 0 : fb->lane].stream;
    rz_strip_alpha_kernel<<<dim3((unsigned)((n / 4 + 1 + 255) / 256)), dim3(256), 0, st>>>(fb->d_color, (uint32_t*)rgb, n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
}

/* private.c:8-50 */
extern "C" void framebuffer_blit(rz_framebuffer *p, unsigned char *destination, int swap_red_blue) {
    Fb *fb = as_fb(p);
    if (!destination) fatal("framebuffer_blit: no destination");
    graphics_flush();
    HIP_OK(hipStreamSynchronize(g.lanes[fb->lane < 0 ? 0 : fb->lane].stream));
    settle_framebuffer(fb);
    apply_pending_clear(fb);
    const int w = fb->pub.width, h = fb->pub.height;
    const size_t n = (size_t)w * h;
    hipStream_t st = g.lanes[fb->lane < 0 ? 0 : fb->lane].stream;
    reserve(g.blit_staging, n, st);
    if (n > g.blit_host_cap) {
        if (g.blit_host) HIP_OK(hipHostFree(g.blit_host));
        HIP_OK(hipHostMalloc((void**)&g.blit_host, n * sizeof(uint32_t), hipHostMallocDefault));
        g.blit_host_cap = n;
    }
    rz_blit_kernel<<<dim3((unsigned)((w + 63) / 64), (unsigned)h), dim3(64), 0, st>>>(fb->d_color, g.blit_staging.ptr, w, h, swap_red_blue ? 1 : 0);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(g.blit_host, g.blit_staging.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    /* three bytes per pixel are the blit's, the fourth stays the destination's (private.c:23-25 writes [0..2] only) */
    for (size_t i = 0; i < n; i++) {
        uint32_t old;
        memcpy(&old, destination + 4 * i, 4);
        old = (old & 0xff000000u) | g.blit_host[i];
        memcpy(destination + 4 * i, &old, 4);
    }
}

/* ---- 2-D overlays: draw2d.c ------------------------------------------------ */

/* everything recorded for the framebuffer is submitted and settled; what follows on its lane comes after it */
static hipStream_t overlay_stream(Fb *fb) {
    graphics_flush();
    HIP_OK(hipStreamSynchronize(g.lanes[fb->lane < 0 ? 0 : fb->lane].stream));
    settle_framebuffer(fb);
    apply_pending_clear(fb);
    if (fb->lane < 0) fb->lane = 0;
    return g.lanes[fb->lane].stream;
}

static void overlay_pixel(const Fb *fb, rz_vec2 p, int *row, int *col) {       /* draw2d.c:24-28 */
    auto sat = [](float f) { return f < 0 ? 0.0f : (f > 1 ? 1.0f : f); };
    *row = (int)((float)(fb->pub.height - 1) * sat(p.y) + 0.5f);
    *col = (int)((float)(fb->pub.width - 1) * sat(p.x) + 0.5f);
}

static void overlay_segments(Fb *fb, rz_vec4 color, const rz_vec2 *points, int count) {
    OverlaySegments s;
    memset(&s, 0, sizeof s);
    int row[3], col[3], longest = 0;
    for (int k = 0; k < count; k++) overlay_pixel(fb, points[k], &row[k], &col[k]);
    const int segments = count == 3 ? 3 : 1;
    for (int k = 0; k < segments; k++) {
        const int n = count == 1 ? 0 : (k + 1) % count;
        s.r0[k] = row[k]; s.c0[k] = col[k]; s.r1[k] = row[n]; s.c1[k] = col[n];
        longest = std::max(longest, std::max(abs(row[k] - row[n]), abs(col[k] - col[n])));
    }
    s.pixel = pack_clear(color);            /* draw2d.c:17-22: float_to_uchar per channel, as a clear packs its colour */
    s.width = fb->pub.width;
    hipStream_t st = overlay_stream(fb);
    rz_overlay_segments_kernel<<<dim3((unsigned)(longest / 64 + 1), (unsigned)segments), dim3(64), 0, st>>>(fb->d_color, s);
    HIP_OK(hipGetLastError());
}

extern "C" void draw2d_draw_point(rz_framebuffer *p, rz_vec4 color, rz_vec2 point) {                   /* draw2d.c:95 */
    overlay_segments(as_fb(p), color, &point, 1);
}

extern "C" void draw2d_draw_line(rz_framebuffer *p, rz_vec4 color, rz_vec2 point0, rz_vec2 point1) {   /* draw2d.c:104 */
    const rz_vec2 points[2] = {point0, point1};
    overlay_segments(as_fb(p), color, points, 2);
}

extern "C" void draw2d_draw_triangle(rz_framebuffer *p, rz_vec4 color, rz_vec2 point0, rz_vec2 point1, rz_vec2 point2) {    /* draw2d.c:114 */
    const rz_vec2 points[3] = {point0, point1, point2};
    overlay_segments(as_fb(p), color, points, 3);
}

static Tex *as_tex(rz_texture *p);
static void upload_texture(Tex *t);

extern "C" void draw2d_draw_texture(rz_framebuffer *p, rz_texture *texture, rz_vec2 origin) {          /* draw2d.c:128 */
    Fb *fb = as_fb(p);
    Tex *t = as_tex(texture);
    if (t->device_stale) upload_texture(t);
    graphics_flush();
    sync_stream();                  /* the image may have been written on another lane (a shadow map shown as an overlay) */
    hipStream_t st = overlay_stream(fb);
    int row, col;
    overlay_pixel(fb, origin, &row, &col);
    const int w = std::min(fb->pub.width - col, t->pub.width), h = std::min(fb->pub.height - row, t->pub.height);
    if (w <= 0 || h <= 0) return;
    rz_overlay_texture_kernel<<<dim3((unsigned)((w + 63) / 64), (unsigned)h), dim3(64), 0, st>>>(
        fb->d_color, fb->pub.width, t->d_data, t->pub.width, t->flags, row, col, w, h);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));       /* the image must not be replaced under the kernel */
}

extern "C" void framebuffer_read(rz_framebuffer *p, void *color, void *depth) {
    read_framebuffer(as_fb(p), color, depth, hipMemcpyDeviceToHost);
}

extern "C" void framebuffer_read_device(rz_framebuffer *p, void *color, void *depth) {
    read_framebuffer(as_fb(p), color, depth, hipMemcpyDeviceToDevice);
}

/* ---------------------------------------------------------------- textures -- */

static Tex *as_tex(rz_texture *p) {
    Tex *t = (Tex*)p;
    if (!p || !g.texs.count(t)) fatal("texture %p was not created by this library", (void*)p);
    return t;
}

static int log2_exact(int v) {
    int b = 0;
    if (v <= 0 || (v & (v - 1))) return -1;
    while ((1 << b) < v) b++;
    return b;
}

/* images whose sides are multiples of 8 are stored in 8x8-texel tiles (texel_index, shade.h) */
static void choose_layout(int w, int h, int *flags) {
    *flags &= ~RZ_TEX_LAYOUT;
    if (w % 8 == 0 && h % 8 == 0 && w < (1 << 24) && h < (1 << 24)) *flags |= RZ_TEX_TILED;   /* texel_index multiplies tile counts with 24 bits */
}

static void drop_derived_handles();

extern "C" rz_texture *texture_create(int width, int height) {      /* texture.c:12 */
    ensure_ready();
    if (width <= 0 || height <= 0) fatal("texture size %dx%d", width, height);
    Tex *t = new Tex();
    t->pub.width = width;
    t->pub.height = height;
    t->pub.buffer = (rz_vec4*)calloc((size_t)width * height, sizeof(rz_vec4));
    t->d_data = nullptr;
    t->d_bytes = 0;
    t->flags = 0;
    t->device_stale = true;
    t->host_stale = false;
    t->write_lane = -1;
    t->write_event = nullptr;
    g.texs.insert(t);
    return &t->pub;
}

extern "C" void texture_release(rz_texture *p) {                    /* texture.c:27 */
    Tex *t = as_tex(p);
    graphics_flush();
    sync_stream();
    drop_derived_handles();
    if (t->d_data) HIP_OK(hipFree(t->d_data));
    if (t->host_from_runtime) HIP_OK(hipHostFree(t->pub.buffer));
    else {
        if (t->host_pinned) (void)hipHostUnregister(t->pub.buffer);
        free(t->pub.buffer);
    }
    g.texs.erase(t);
    delete t;
}

extern "C" void texture_invalidate(rz_texture *p) {
    Tex *t = as_tex(p);
    if (!t->pub.buffer) fatal("texture_invalidate: the texture's host image was never asked for (texture_sync), so it cannot have been modified");
    t->device_stale = true;
    t->host_stale = false;
}

static void device_image(Tex *t, int flags_wanted, size_t texel_bytes) {
    /* whole 16-byte units: the samplers read the aligned 16 bytes around a 4-byte texel */
    size_t bytes = ((size_t)t->pub.width * t->pub.height * texel_bytes + 15) & ~(size_t)15;
    int flags = flags_wanted;
    choose_layout(t->pub.width, t->pub.height, &flags);
    if (t->d_bytes != bytes) {
        /* the device address changes: cube / IBL handle tables that embed it are rebuilt */
        if (t->d_data) { graphics_flush(); sync_stream(); drop_derived_handles(); HIP_OK(hipFree(t->d_data)); }
        HIP_OK(hipMalloc(&t->d_data, bytes));
        t->d_bytes = bytes;
    }
    t->image_version++;             /* every caller is about to write the image */
    t->flags = flags;
}

/* ---- lossless 4-byte device formats (rz_device.h) --------------------------
 * texture->buffer is always float RGBA, as the reference defines it.  Most of
 * the reference's images come from 8-bit TGA or RGBE .hdr files, so their
 * floats take one of 256 values per channel (or mantissa * 2^e): such an image
 * is stored in 4 bytes per texel and decoded on fetch to the SAME floats.
 * Every texel is verified; an image with one texel that does not fit stays
 * float4. */
static float g_unorm8[256], g_srgb8[256];

static void fill_packed_tables() {
    static bool done = false;
    if (done) return;
    done = true;
    for (int k = 0; k < 256; k++) {
        g_unorm8[k] = (float)k / 255.0f;                        /* texture.c:40-61 */
        g_srgb8[k] = (float)pow((double)g_unorm8[k], 2.2);      /* maths.c:37-39 */
        /* the device computes k/255 as fma(fma(-q, 255, k), r, q), q = k * r, r = 1/255 */
        const float r = 1.0f / 255.0f, f = (float)k, q = f * r;
        if (fmaf(fmaf(-q, 255.0f, f), r, q) != g_unorm8[k]) fatal("byte %d: the division-free k/255 is off", k);
    }
}

static void init_packed_formats() {
    fill_packed_tables();
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(rz_srgb8_table), g_srgb8, sizeof g_srgb8));
}

static int byte_of(const float *table, float v) {               /* tables ascend strictly */
    const float *hit = std::lower_bound(table, table + 256, v);
    return (hit != table + 256 && *hit == v) ? (int)(hit - table) : -1;
}

/* returns the format flag and fills `out`, or 0 when the image needs float4 */
static int pack_texels(const rz_vec4 *texels, size_t n, std::vector<uint32_t> &out) {
    out.resize(n);
    for (int format : {RZ_TEX_UNORM8, RZ_TEX_SRGB8}) {
        const float *rgb = format == RZ_TEX_UNORM8 ? g_unorm8 : g_srgb8;
        size_t i = 0;
        for (; i < n; i++) {
            const int r = byte_of(rgb, texels[i].x), g_ = byte_of(rgb, texels[i].y), b = byte_of(rgb, texels[i].z);
            const int a = byte_of(g_unorm8, texels[i].w);
            if ((r | g_ | b | a) < 0) break;
            out[i] = (uint32_t)r | ((uint32_t)g_ << 8) | ((uint32_t)b << 16) | ((uint32_t)a << 24);
        }
        if (i == n) return format;
    }
    for (size_t i = 0; i < n; i++) {                            /* shared-exponent: rgb = mantissa * 2^(e - 136) */
        const float v[3] = {texels[i].x, texels[i].y, texels[i].z};
        if (texels[i].w != 1.0f) return 0;
        const float top = std::max(v[0], std::max(v[1], v[2]));
        if (!(v[0] >= 0 && v[1] >= 0 && v[2] >= 0) || std::isinf(top)) return 0;
        if (top == 0) { out[i] = 0; continue; }
        int e2;
        frexpf(top, &e2);                                       /* top = f * 2^e2, f in [0.5, 1): mantissa = f * 256 */
        const int e = std::max(e2 + 128, 1);                    /* a smaller exponent byte does not exist; mantissas shrink */
        if (e > 255) return 0;
        const float factor = ldexpf(1.0f, e - 136);
        uint32_t word = (uint32_t)e << 24;
        for (int c = 0; c < 3; c++) {
            const float m = v[c] / factor;                      /* exact: a power of two */
            if (!(m >= 0 && m <= 255 && m == (float)(int)m) || (float)(int)m * factor != v[c]) return 0;
            word |= (uint32_t)(int)m << (8 * c);
        }
        out[i] = word;
    }
    return RZ_TEX_RGBE8;
}

/* the format decision alone, for tests: host code only, no device needed */
extern "C" int mi355r_pack_texels(const rz_vec4 *texels, size_t count, unsigned int *packed) {
    fill_packed_tables();
    std::vector<uint32_t> out;
    const int format = pack_texels(texels, count, out);
    if (format && packed) memcpy(packed, out.data(), count * sizeof(uint32_t));
    return format;
}

/* host float4 image -> device (packed when lossless, 8x8 tiles where possible); load-time path */
static void upload_texture(Tex *t) {
    int w = t->pub.width, h = t->pub.height;
    size_t bytes = (size_t)w * h * sizeof(float4);
    graphics_flush();               /* recorded draws may still sample the old image */
    /* ... and so may passes that are still running, on any lane: an image that is replaced (after
     * texture_invalidate, or one the GPU wrote) waits for all of them.  Only the first upload of an
     * image has no reader yet. */
    if (t->d_data || t->write_lane >= 0 || !t->read_events.empty()) sync_stream();
    t->write_lane = -1;
    std::vector<uint32_t> packed;
    const int format = g.pack_textures ? pack_texels(t->pub.buffer, (size_t)w * h, packed) : 0;
    if (g.verbose) fprintf(stderr, "mi355r: texture %dx%d -> format %d\n", w, h, format);
    if (format) {
        device_image(t, format, sizeof(uint32_t));
        uint32_t *staging;
        HIP_OK(hipMalloc((void**)&staging, packed.size() * sizeof(uint32_t)));
        HIP_OK(hipMemcpyAsync(staging, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
        rz_texture_tile32_kernel<<<dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, g.stream>>>(
            staging, (uint32_t*)t->d_data, w, h, t->flags);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(g.stream));
        HIP_OK(hipFree(staging));
        t->device_stale = false;
        t->host_stale = false;
        return;
    }
    device_image(t, 0, sizeof(float4));
    float4 *staging;
    HIP_OK(hipMalloc((void**)&staging, bytes));
    HIP_OK(hipMemcpyAsync(staging, t->pub.buffer, bytes, hipMemcpyHostToDevice, g.stream));
    rz_texture_tile_kernel<<<dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, g.stream>>>(
        staging, (float4*)t->d_data, w, h, t->flags);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    HIP_OK(hipFree(staging));
    t->device_stale = false;
    t->host_stale = false;
}

static DevTex dev_tex(rz_texture *p) {
    DevTex d;
    memset(&d, 0, sizeof d);
    if (!p) return d;
    Tex *t = as_tex(p);
    if (t->device_stale) upload_texture(t);
    t->used_epoch = g.flush_epoch;
    if (t->pending_write) g.current_pending_textures.push_back(t);
    else if (t->write_lane >= 0) g.current_gpu_textures.push_back(t);
    d.data = t->d_data;
    d.w = t->pub.width;
    d.h = t->pub.height;
    d.flags = t->flags;
    d.reserved = 0;
    return d;
}

/* texture_create_packed: the image arrives in a 4-byte device format, straight from the file's bytes */
extern "C" rz_texture *texture_create_packed(int width, int height, int format, const unsigned int *words) {
    ensure_ready();
    if (!g.pack_textures) return nullptr;
    if (width <= 0 || height <= 0) fatal("texture size %dx%d", width, height);
    if (format != RZ_TEX_UNORM8 && format != RZ_TEX_SRGB8 && format != RZ_TEX_RGBE8) fatal("texture_create_packed: format %d", format);
    Tex *t = new Tex();
    t->pub.width = width;
    t->pub.height = height;
    t->pub.buffer = nullptr;            /* made by texture_sync when somebody asks for the floats */
    t->d_data = nullptr;
    t->d_bytes = 0;
    t->flags = 0;
    t->write_lane = -1;
    t->write_event = nullptr;
    g.texs.insert(t);
    const size_t n = (size_t)width * height;
    device_image(t, format, sizeof(uint32_t));
    uint32_t *staging;
    HIP_OK(hipMalloc((void**)&staging, n * sizeof(uint32_t)));
    HIP_OK(hipMemcpyAsync(staging, words, n * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
    rz_texture_tile32_kernel<<<dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, g.stream>>>(
        staging, (uint32_t*)t->d_data, width, height, t->flags);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    HIP_OK(hipFree(staging));
    t->device_stale = false;
    t->host_stale = true;
    return &t->pub;
}

extern "C" void texture_sync(rz_texture *p) {
    Tex *t = as_tex(p);
    if (!t->host_stale) return;
    int w = t->pub.width, h = t->pub.height;
    size_t bytes = (size_t)w * h * sizeof(float4);
    if (!t->pub.buffer) t->pub.buffer = (rz_vec4*)calloc((size_t)w * h, sizeof(rz_vec4));
    graphics_flush();
    sync_stream();
    float4 *staging;
    HIP_OK(hipMalloc((void**)&staging, bytes));
    rz_texture_untile_kernel<<<dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, g.stream>>>(
        t->d_data, staging, w, h, t->flags);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(t->pub.buffer, staging, bytes, hipMemcpyDeviceToHost, g.stream));
    sync_stream();
    HIP_OK(hipFree(staging));
    t->host_stale = false;
}

/* The framebuffer's lane writes the texture: it first waits for every pass
 * that still samples the previous contents; later readers wait for the write. */
static hipStream_t begin_texture_write(Tex *t, Fb *fb) {
    if (fb->lane < 0) fb->lane = 0;
    hipStream_t st = g.lanes[fb->lane].stream;
    for (hipEvent_t e : t->read_events) HIP_OK(hipStreamWaitEvent(st, e, 0));
    t->read_events.clear();
    if (t->write_event && t->write_lane != fb->lane) HIP_OK(hipStreamWaitEvent(st, t->write_event, 0));
    return st;
}
static void end_texture_write(Tex *t, Fb *fb, hipEvent_t shared = nullptr) {
    t->write_lane = fb->lane;
    t->write_event = shared ? shared : new_event(g.lanes[fb->lane].stream);
    t->write_seq = ++g.write_seq;
    t->device_stale = false;
    t->host_stale = true;
}

static void launch_texture_write(Tex *t, Fb *fb, bool depth);

/* texture.c:130-158.  When the framebuffer still has recorded draws the copy is recorded too, behind
 * them, and runs at the next flush between the passes on either side of it: a frame made of a
 * shadow pass, this copy and a main pass that samples the texture then needs no flush of its own,
 * and a caller that alternates between several shadow buffers gets several frames into one
 * flush.  A texture that recorded draws already refer to is never overwritten behind their back:
 * that case flushes first. */
static void texture_from_buffer(Tex *t, Fb *fb, bool depth, const char *who) {
    if (t->pub.width != fb->pub.width || t->pub.height != fb->pub.height) fatal("%s: size mismatch", who);
    if (!t->host_pinned) {
        /* its host image will be filled from the device (texture_sync): page-locked, that copy needs no staging.  Where
         * the process may not lock pages it allocated itself (an ordinary user's limit), the image moves into memory
         * the runtime hands out page-locked; its contents are about to be replaced anyway */
        const size_t bytes = (size_t)t->pub.width * t->pub.height * sizeof(rz_vec4);
        if (t->pub.buffer && hipHostRegister(t->pub.buffer, bytes, hipHostRegisterDefault) == hipSuccess) {
            t->host_pinned = true;
        } else {
            (void)hipGetLastError();
            rz_vec4 *locked = nullptr;
            if (hipHostMalloc((void**)&locked, bytes, hipHostMallocDefault) == hipSuccess) {
                if (t->pub.buffer) memcpy(locked, t->pub.buffer, bytes);
                else memset(locked, 0, bytes);          /* an image that came in packed: nobody has asked for its floats yet */
                free(t->pub.buffer);
                t->pub.buffer = locked;
                t->host_pinned = t->host_from_runtime = true;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    const bool in_use = t->pending_write || (!g.pending.empty() && t->used_epoch == g.flush_epoch);
    if (!has_pending_for(fb) || in_use || !g.record_copies) {
        graphics_flush();
        apply_pending_clear(fb);
        if (depth) device_image(t, RZ_TEX_R32F, sizeof(float));
        else device_image(t, 0, sizeof(float4));
        launch_texture_write(t, fb, depth);
        return;
    }
    if (depth) device_image(t, RZ_TEX_R32F, sizeof(float));
    else device_image(t, 0, sizeof(float4));
    g.tex_writes.push_back({t, fb, depth, g.pending.size()});
    t->pending_write = true;
    t->device_stale = false;
    t->host_stale = true;
}

extern "C" void texture_from_depthbuffer(rz_texture *tp, rz_framebuffer *fp) {   /* texture.c:147 */
    texture_from_buffer(as_tex(tp), as_fb(fp), true, "texture_from_depthbuffer");
}

extern "C" void texture_from_colorbuffer(rz_texture *tp, rz_framebuffer *fp) {   /* texture.c:130 */
    texture_from_buffer(as_tex(tp), as_fb(fp), false, "texture_from_colorbuffer");
}

/* Host-side accessors with the reference's addressing (texture.c:160-273).
 * They read the host image (synchronising it first if the GPU wrote it); the
 * pipeline itself samples on the device (shade.h). */
static rz_vec4 host_texel(rz_texture *p, float u, float v) {
    texture_sync(p);
    int c = (int)((p->width - 1) * u), r = (int)((p->height - 1) * v);
    return p->buffer[r * p->width + c];
}
extern "C" rz_vec4 texture_repeat_sample(rz_texture *p, rz_vec2 uv) {
    return host_texel(p, uv.x - (float)floor(uv.x), uv.y - (float)floor(uv.y));
}
extern "C" rz_vec4 texture_clamp_sample(rz_texture *p, rz_vec2 uv) {
    auto sat = [](float f) { return f < 0 ? 0.0f : (f > 1 ? 1.0f : f); };
    return host_texel(p, sat(uv.x), sat(uv.y));
}
extern "C" rz_vec4 texture_sample(rz_texture *p, rz_vec2 uv) { return texture_repeat_sample(p, uv); }

static int host_cube_face(rz_vec3 d, rz_vec2 *st) {
    float ax = (float)fabs(d.x), ay = (float)fabs(d.y), az = (float)fabs(d.z), ma, sc, tc;
    int face;
    if (ax > ay && ax > az) { ma = ax; if (d.x > 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay > az) { ma = ay; if (d.y > 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z > 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    st->x = (sc / ma + 1) / 2;
    st->y = 1 - (tc / ma + 1) / 2;
    return face;
}
extern "C" rz_vec4 cubemap_repeat_sample(rz_cubemap *c, rz_vec3 d) { rz_vec2 st; int f = host_cube_face(d, &st); return texture_repeat_sample(c->faces[f], st); }
extern "C" rz_vec4 cubemap_clamp_sample(rz_cubemap *c, rz_vec3 d) { rz_vec2 st; int f = host_cube_face(d, &st); return texture_clamp_sample(c->faces[f], st); }
extern "C" rz_vec4 cubemap_sample(rz_cubemap *c, rz_vec3 d) { return cubemap_repeat_sample(c, d); }

/* cube maps and IBL sets are handle tables in device memory, built on first use */
static const DevTex *dev_cube(rz_cubemap *cube) {
    if (!cube) return nullptr;
    auto it = g.cubes.find(cube);
    if (it != g.cubes.end()) return it->second;
    DevTex faces[6];
    for (int i = 0; i < 6; i++) faces[i] = dev_tex(cube->faces[i]);
    DevTex *d;
    HIP_OK(hipMalloc((void**)&d, sizeof faces));
    HIP_OK(hipMemcpy(d, faces, sizeof faces, hipMemcpyHostToDevice));
    g.cubes[cube] = d;
    return d;
}

/* A sky box's cube map as one plane of display bytes (RZ_TEX_DISPLAY8), its six faces one behind the other in
 * the faces' own layout: made when the cube is first drawn as a sky, made again when a face's image has changed.
 * false: the faces are sampled as they are (faces of different sizes, or one the GPU writes: the plane would
 * have to follow every write). */
static bool cube_display(rz_cubemap *cube, DevTex *plane) {
    if (!g.sky_bytes) return false;
    Tex *f[6];
    for (int i = 0; i < 6; i++) {
        f[i] = as_tex(cube->faces[i]);
        if (f[i]->write_lane >= 0 || f[i]->pending_write || !f[i]->d_data) return false;
        if (f[i]->pub.width != f[0]->pub.width || f[i]->pub.height != f[0]->pub.height ||
            (f[i]->flags & RZ_TEX_LAYOUT) != (f[0]->flags & RZ_TEX_LAYOUT)) return false;
    }
    const size_t n = (size_t)f[0]->pub.width * f[0]->pub.height;
    if (n > ((size_t)1 << 24)) return false;        /* face * texels + cell stays a 32-bit index */
    CubeDisplay &cd = g.cube_displays[cube];
    bool fresh = cd.d && cd.face_texels == n;
    for (int i = 0; i < 6; i++) fresh = fresh && cd.version[i] == f[i]->image_version;
    if (!fresh) {
        if (cd.d) {                 /* frames that show the old plane may still be on their way */
            graphics_flush();
            sync_stream();
            if (cd.face_texels != n) { HIP_OK(hipFree(cd.d)); cd.d = nullptr; }
        }
        if (!cd.d) HIP_OK(hipMalloc((void**)&cd.d, 6 * n * sizeof(uint32_t)));
        cd.face_texels = n;
        for (int i = 0; i < 6; i++) {
            rz_texture_display_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream>>>(f[i]->d_data, cd.d + (size_t)i * n, n, f[i]->flags);
            HIP_OK(hipGetLastError());
            cd.version[i] = f[i]->image_version;
        }
        HIP_OK(hipStreamSynchronize(g.stream));
    }
    plane->data = cd.d;
    plane->w = f[0]->pub.width;
    plane->h = f[0]->pub.height;
    plane->flags = (f[0]->flags & RZ_TEX_LAYOUT) | RZ_TEX_DISPLAY8;
    plane->reserved = 0;
    return true;
}

static const DevIbl *dev_ibl(rz_ibldata *ibl) {
    if (!ibl) return nullptr;
    auto it = g.ibls.find(ibl);
    if (it != g.ibls.end()) return it->second;
    if (ibl->mip_levels < 1 || ibl->mip_levels > 15) fatal("ibldata with %d levels", ibl->mip_levels);
    DevIbl *h = (DevIbl*)calloc(1, sizeof(DevIbl));
    h->mip_levels = ibl->mip_levels;
    h->lut = dev_tex(ibl->brdf_lut);
    for (int f = 0; f < 6; f++) h->diffuse[f] = dev_tex(ibl->diffuse_map->faces[f]);
    for (int l = 0; l < ibl->mip_levels; l++)
        for (int f = 0; f < 6; f++) h->specular[l][f] = dev_tex(ibl->specular_maps[l]->faces[f]);
    bool narrow = (h->lut.flags & RZ_TEX_NARROW_FORMATS) != 0;
    for (int f = 0; f < 6; f++) narrow = narrow && (h->diffuse[f].flags & RZ_TEX_NARROW_FORMATS);
    for (int l = 0; l < ibl->mip_levels; l++)
        for (int f = 0; f < 6; f++) narrow = narrow && (h->specular[l][f].flags & RZ_TEX_NARROW_FORMATS);
    g.ibl_narrow[ibl] = narrow;
    DevIbl *d;
    HIP_OK(hipMalloc((void**)&d, sizeof(DevIbl)));
    HIP_OK(hipMemcpy(d, h, sizeof(DevIbl), hipMemcpyHostToDevice));
    free(h);
    g.ibls[ibl] = d;
    return d;
}

/* a texture went away or moved: every table that may embed its address is
 * dropped and rebuilt on next use */
static void drop_derived_handles() {
    for (auto &kv : g.cubes) HIP_OK(hipFree(kv.second));
    for (auto &kv : g.ibls) HIP_OK(hipFree(kv.second));
    for (auto &kv : g.cube_displays) if (kv.second.d) HIP_OK(hipFree(kv.second.d));
    g.cube_displays.clear();
    g.cubes.clear();
    g.ibls.clear();
    g.ibl_narrow.clear();
}

/* ---------------------------------------------------------------- programs -- */

extern "C" rz_vec4 blinn_vertex_shader(void*, void*, void*);
extern "C" rz_vec4 blinn_fragment_shader(void*, void*, int*, int);
extern "C" rz_vec4 pbr_vertex_shader(void*, void*, void*);
extern "C" rz_vec4 pbr_fragment_shader(void*, void*, int*, int);
extern "C" rz_vec4 skybox_vertex_shader(void*, void*, void*);
extern "C" rz_vec4 skybox_fragment_shader(void*, void*, int*, int);

extern "C" void program_register_family(rz_vertex_shader *vs, rz_fragment_shader *fs, rz_family family) {
    g.families.push_back({(void*)vs, (void*)fs, (int)family});
}

static int identify_family(void *vs, void *fs, int sa, int sv, int su) {
    for (auto &r : g.families) if (r.vs == vs && r.fs == fs) return r.family;
    /* this library's own identity tags (host/tags.c) */
    if (vs == (void*)blinn_vertex_shader && fs == (void*)blinn_fragment_shader) return RZ_FAMILY_BLINN;
    if (vs == (void*)pbr_vertex_shader && fs == (void*)pbr_fragment_shader) return RZ_FAMILY_PBR;
    if (vs == (void*)skybox_vertex_shader && fs == (void*)skybox_fragment_shader) return RZ_FAMILY_SKYBOX;
    /* the host program's shader functions, when it exports them */
    static const char *names[3][2] = {{"blinn_vertex_shader", "blinn_fragment_shader"},
                                      {"pbr_vertex_shader", "pbr_fragment_shader"},
                                      {"skybox_vertex_shader", "skybox_fragment_shader"}};
    for (int f = 0; f < 3; f++) {
        void *v = dlsym(RTLD_DEFAULT, names[f][0]), *p = dlsym(RTLD_DEFAULT, names[f][1]);
        if (v && p && v == vs && p == fs) return f;
    }
    /* struct sizes of the three families (x86-64) */
    if (sa == 64 && sv == 44 && su == 344) return RZ_FAMILY_BLINN;
    if (sa == 80 && sv == 84 && su == 440) return RZ_FAMILY_PBR;
    if (sa == 12 && sv == 12 && su == 72) return RZ_FAMILY_SKYBOX;
    return -1;
}

extern "C" rz_program *program_create(rz_vertex_shader *vs, rz_fragment_shader *fs,
                                      int sizeof_attribs, int sizeof_varyings, int sizeof_uniforms,
                                      int double_sided, int enable_blend) {            /* graphics.c:82 */
    ensure_ready();
    if (sizeof_attribs <= 0 || sizeof_varyings <= 0 || sizeof_uniforms <= 0 || sizeof_varyings % 4)
        fatal("program_create: bad block sizes %d/%d/%d", sizeof_attribs, sizeof_varyings, sizeof_uniforms);
    int family = identify_family((void*)vs, (void*)fs, sizeof_attribs, sizeof_varyings, sizeof_uniforms);
    if (family < 0)
        fatal("program_create: shader pair %p/%p (blocks %d/%d/%d bytes) is not one of the blinn, pbr or "
              "skybox families; this library has no CPU shader path (see program_register_family)",
              (void*)vs, (void*)fs, sizeof_attribs, sizeof_varyings, sizeof_uniforms);
    static const int expect[3][3] = {{64, 44, 344}, {80, 84, 440}, {12, 12, 72}};
    if (sizeof_attribs != expect[family][0] || sizeof_varyings != expect[family][1] || sizeof_uniforms != expect[family][2])
        fatal("program_create: family %d expects blocks of %d/%d/%d bytes, got %d/%d/%d", family, expect[family][0],
              expect[family][1], expect[family][2], sizeof_attribs, sizeof_varyings, sizeof_uniforms);
    rz_program *p = new rz_program();
    p->family = family;
    p->attrib_bytes = sizeof_attribs;
    p->varying_bytes = sizeof_varyings;
    p->uniform_bytes = sizeof_uniforms;
    p->double_sided = double_sided;
    p->blend = enable_blend;
    p->attribs[0] = calloc(3, sizeof_attribs);          /* one allocation: a pbr program's three blocks are then a piece of vertex stream as they are */
    p->attribs[1] = (char*)p->attribs[0] + sizeof_attribs;
    p->attribs[2] = (char*)p->attribs[0] + 2 * (size_t)sizeof_attribs;
    p->uniforms = calloc(1, sizeof_uniforms);
    return p;
}

extern "C" void program_release(rz_program *p) {                    /* graphics.c:120 */
    graphics_flush();
    auto remembered = g.replays.find(p);
    if (remembered != g.replays.end()) {
        if (remembered->second.d_vertices) { sync_stream(); HIP_OK(hipFree(remembered->second.d_vertices)); }
        g.replays.erase(remembered);
    }
    free(p->attribs[0]);
    free(p->uniforms);
    delete p;
}

extern "C" void *program_get_attribs(rz_program *p, int nth) {     /* graphics.c:134 */
    if (nth < 0 || nth > 2) fatal("program_get_attribs: vertex %d", nth);
    return p->attribs[nth];
}

extern "C" void *program_get_uniforms(rz_program *p) { return p->uniforms; }   /* graphics.c:139 */

/* ------------------------------------------------------------------ meshes -- */

static Mesh &device_mesh(const rz_vertex *verts, int num_faces) {
    auto it = g.meshes.find(verts);
    if (it != g.meshes.end() && it->second.num_faces == num_faces) return it->second;
    if (it != g.meshes.end()) mesh_invalidate(verts);
    Mesh m;
    size_t bytes = (size_t)num_faces * 3 * sizeof(rz_vertex);
    HIP_OK(hipMalloc(&m.d_vertices, bytes));
    HIP_OK(hipMemcpy(m.d_vertices, verts, bytes, hipMemcpyHostToDevice));
    m.num_faces = num_faces;
    int top = 0;
    for (int i = 0; i < num_faces * 3; i++) {
        const rz_vertex &v = verts[i];
        top = std::max(top, std::max(std::max((int)v.joint.x, (int)v.joint.y), std::max((int)v.joint.z, (int)v.joint.w)));
    }
    m.num_joints = top + 1;
    return g.meshes[verts] = m;
}

extern "C" void mesh_invalidate(const rz_vertex *verts) {
    auto it = g.meshes.find(verts);
    if (it == g.meshes.end()) return;
    graphics_flush();
    sync_stream();
    HIP_OK(hipFree(it->second.d_vertices));
    g.meshes.erase(it);
}

/* --------------------------------------------------------------- recording -- */

static void copy_floats(float *dst, const void *src, int n) { memcpy(dst, src, sizeof(float) * n); }

/* uniform block (host layout, host pointers) -> DrawRec (device form) */
static void snapshot_uniforms_once(rz_program *p, Pending &pd, int num_joints);
static bool g_ignore_host_joints = false;   /* while graphics_draw_mesh_posed takes its snapshot */

/* Resolving a texture can upload it, and an upload flushes what is recorded -- recorded
 * framebuffer copies included, which changes how a texture resolved a moment ago has to be waited
 * for.  A snapshot during which a flush happened is therefore taken again (nothing is left to
 * upload the second time). */
static void snapshot_uniforms(rz_program *p, Pending &pd, int num_joints) {
    for (int attempt = 0; attempt < 3; attempt++) {
        const unsigned long long epoch = g.flush_epoch;
        snapshot_uniforms_once(p, pd, num_joints);
        if (g.flush_epoch == epoch) return;
    }
    fatal("textures keep being uploaded while a draw is recorded");
}

static void snapshot_uniforms_once(rz_program *p, Pending &pd, int num_joints) {
    DrawRec &r = pd.rec;
    memset(&r, 0, sizeof r);
    g.current_gpu_textures.clear();
    g.current_pending_textures.clear();
    r.family = p->family;
    r.flags = (p->double_sided ? RZ_DRAW_DOUBLE_SIDED : 0) | (p->blend ? RZ_DRAW_BLEND : 0);
    const rz_mat4 *joints = nullptr;
    const rz_mat3 *joint_normals = nullptr;
    if (p->family == RZ_FAMILY_SKYBOX) {
        const rz_skybox_uniforms *u = (const rz_skybox_uniforms*)p->uniforms;
        r.num_varyings = 3;
        copy_floats(r.camera_vp, &u->vp_matrix, 16);
        r.skybox = dev_cube(u->skybox);
        if (!r.skybox) fatal("skybox draw without a cube map");
        bool uniform = true;
        for (int i = 0; i < 6; i++) {
            r.maps[i] = dev_tex(u->skybox->faces[i]);
            uniform = uniform && r.maps[i].w == r.maps[0].w && r.maps[i].h == r.maps[0].h && r.maps[i].flags == r.maps[0].flags;
        }
        if (uniform) r.flags |= RZ_DRAW_UNIFORM_CUBE;
        /* a program that blends needs the colour itself, not its bytes */
        if (!p->blend && cube_display(u->skybox, &r.maps[0])) r.flags |= RZ_DRAW_SKY_BYTES;
    } else if (p->family == RZ_FAMILY_BLINN) {
        const rz_blinn_uniforms *u = (const rz_blinn_uniforms*)p->uniforms;
        r.num_varyings = 11;
        copy_floats(r.light_dir, &u->light_dir, 3);
        copy_floats(r.camera_pos, &u->camera_pos, 3);
        copy_floats(r.model, &u->model_matrix, 16);
        copy_floats(r.normal, &u->normal_matrix, 9);
        copy_floats(r.light_vp, &u->light_vp_matrix, 16);
        copy_floats(r.camera_vp, &u->camera_vp_matrix, 16);
        joints = u->joint_matrices;
        joint_normals = u->joint_n_matrices;
        r.ambient = u->ambient_intensity;
        r.punctual = u->punctual_intensity;
        if (!u->shadow_pass) r.shadow_map = dev_tex(u->shadow_map);     /* the shadow pass renders it, it does not sample it */
        copy_floats(r.base, &u->basecolor, 4);
        r.shininess = u->shininess;
        r.maps[0] = dev_tex(u->diffuse_map);
        r.maps[1] = dev_tex(u->specular_map);
        r.maps[2] = dev_tex(u->emission_map);
        r.alpha_cutoff = u->alpha_cutoff;
        if (u->shadow_pass) r.flags |= RZ_DRAW_SHADOW_PASS;
    } else {
        const rz_pbr_uniforms *u = (const rz_pbr_uniforms*)p->uniforms;
        r.num_varyings = 21;
        copy_floats(r.light_dir, &u->light_dir, 3);
        copy_floats(r.camera_pos, &u->camera_pos, 3);
        copy_floats(r.model, &u->model_matrix, 16);
        copy_floats(r.normal, &u->normal_matrix, 9);
        copy_floats(r.light_vp, &u->light_vp_matrix, 16);
        copy_floats(r.camera_vp, &u->camera_vp_matrix, 16);
        joints = u->joint_matrices;
        joint_normals = u->joint_n_matrices;
        r.ambient = u->ambient_intensity;
        r.punctual = u->punctual_intensity;
        if (!u->shadow_pass) r.shadow_map = dev_tex(u->shadow_map);     /* the shadow pass renders it, it does not sample it */
        r.workflow = u->workflow;
        if (u->workflow == 0) {
            copy_floats(r.base, &u->basecolor_factor, 4);
            r.metalness = u->metalness_factor;
            r.roughness = u->roughness_factor;
            r.maps[0] = dev_tex(u->basecolor_map);
            r.maps[1] = dev_tex(u->metalness_map);
            r.maps[2] = dev_tex(u->roughness_map);
        } else {
            copy_floats(r.base, &u->diffuse_factor, 4);
            copy_floats(r.specular, &u->specular_factor, 3);
            r.glossiness = u->glossiness_factor;
            r.maps[0] = dev_tex(u->diffuse_map);
            r.maps[1] = dev_tex(u->specular_map);
            r.maps[2] = dev_tex(u->glossiness_map);
        }
        r.maps[3] = dev_tex(u->normal_map);
        r.maps[4] = dev_tex(u->occlusion_map);
        r.maps[5] = dev_tex(u->emission_map);
        r.ibl = dev_ibl(u->ibldata);
        r.alpha_cutoff = u->alpha_cutoff;
        r.layer_view = u->layer_view;
        if (u->shadow_pass) r.flags |= RZ_DRAW_SHADOW_PASS;
    }
    if (p->family != RZ_FAMILY_SKYBOX) {
        bool narrow = !r.shadow_map.data || (r.shadow_map.flags & RZ_TEX_NARROW_FORMATS);
        for (int i = 0; i < 6; i++) narrow = narrow && (!r.maps[i].data || (r.maps[i].flags & RZ_TEX_NARROW_FORMATS));
        if (r.ibl) narrow = narrow && g.ibl_narrow[((const rz_pbr_uniforms*)p->uniforms)->ibldata];
        if (narrow) r.flags |= RZ_DRAW_NARROW_TEXELS;
    }
    pd.gpu_textures = g.current_gpu_textures;
    pd.pending_textures = g.current_pending_textures;
    /* joint matrices are rewritten in place by the skeleton every frame
     * (skeleton.c:336-337): copy them now */
    pd.num_joints = 0;
    if (joints && !g_ignore_host_joints) {
        if (!joint_normals && !(r.flags & RZ_DRAW_SHADOW_PASS)) fatal("joint matrices without joint normal matrices");
        pd.num_joints = num_joints;
        r.flags |= RZ_DRAW_SKINNED;
        pd.joints_at = g.side.size();
        g.side.insert(g.side.end(), (const char*)joints, (const char*)joints + sizeof(rz_mat4) * num_joints);
        pd.joint_normals_at = g.side.size();
        if (joint_normals)
            g.side.insert(g.side.end(), (const char*)joint_normals, (const char*)joint_normals + sizeof(rz_mat3) * num_joints);
        else
            g.side.insert(g.side.end(), sizeof(rz_mat3) * num_joints, 0);    /* a shadow pass reads none; the layout stays the same */
        while (g.side.size() % 16) g.side.push_back(0);
    }
}

extern "C" void graphics_draw_mesh(rz_framebuffer *fp, rz_program *program, const rz_vertex *vertices, int num_faces) {
    Fb *fb = as_fb(fp);
    if (num_faces <= 0) return;
    Pending pd;                     /* built aside: texture uploads below may flush the list */
    pd.fb = fb;
    pd.program = program;
    pd.transient = false;
    snapshot_uniforms(program, pd, device_mesh(vertices, num_faces).num_joints);
    Mesh &m = device_mesh(vertices, num_faces);
    pd.rec.vertices = m.d_vertices;
    pd.rec.num_faces = num_faces;
    g.pending.push_back(std::move(pd));
}

/* ----------------------------------------------------- poses on the device -- */

extern "C" rz_pose_source *pose_source_create(const rz_skeleton_desc *sk) {
    ensure_ready();
    const int J = sk->num_joints;
    if (J <= 0) fatal("pose_source_create: %d joints", J);
    if (J > RZ_POSE_MAX_JOINTS) return nullptr;
    for (int j = 0; j < J; j++) if (sk->parent[j] >= j) return nullptr;     /* the kernel walks joints in index order */
    std::vector<uint32_t> block(4 + (size_t)J + 16 * (size_t)J + 4 * 3 * (size_t)J, 0u);
    block[0] = (uint32_t)J;
    memcpy(&block[4], sk->parent, sizeof(int) * J);
    memcpy(&block[4 + J], sk->inverse_bind, sizeof(rz_mat4) * J);
    const size_t tracks_at = 4 + (size_t)J + 16 * (size_t)J;
    for (int j = 0; j < J; j++)
        for (int k = 0; k < 3; k++) {
            const int count = sk->key_count[3 * j + k], width = k == 1 ? 4 : 3;
            if (count < 0) fatal("pose_source_create: joint %d has %d keys", j, count);
            PoseTrack t;
            t.count = count;
            t.ascending = count > 0;
            const float *times = sk->key_times[3 * j + k];
            for (int i = 1; i < count; i++) if (!(times[i - 1] < times[i])) t.ascending = 0;
            t.times_at = (uint32_t)block.size();
            block.insert(block.end(), (const uint32_t*)times, (const uint32_t*)times + count);
            t.values_at = (uint32_t)block.size();
            const float *values = sk->key_values[3 * j + k];
            block.insert(block.end(), (const uint32_t*)values, (const uint32_t*)values + (size_t)count * width);
            memcpy(&block[tracks_at + 4 * (3 * (size_t)j + k)], &t, sizeof t);
        }
    rz_pose_source *src = new rz_pose_source();
    src->num_joints = J;
    src->max_time = sk->max_time;
    HIP_OK(hipMalloc((void**)&src->d_block, block.size() * sizeof(uint32_t)));
    HIP_OK(hipMemcpy(src->d_block, block.data(), block.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    g.pose_sources.insert(src);
    return src;
}

static rz_pose_source *as_pose_source(rz_pose_source *src) {
    if (!src || !g.pose_sources.count(src)) fatal("pose source %p was not created by this library", (void*)src);
    return src;
}

extern "C" void pose_source_release(rz_pose_source *src) {
    as_pose_source(src);
    graphics_flush();
    sync_stream();
    HIP_OK(hipFree(src->d_block));
    g.pose_sources.erase(src);
    delete src;
}

/* skeleton.c:316 */
static float wrapped_time(const rz_pose_source *src, float frame_time) { return (float)fmod(frame_time, src->max_time); }

extern "C" void pose_source_evaluate(rz_pose_source *src, const float *frame_times, int count, rz_mat4 *joints, rz_mat3 *normals) {
    as_pose_source(src);
    if (count <= 0) return;
    const size_t J = (size_t)src->num_joints;
    float *d_joints, *d_normals;
    PoseJob *d_jobs;
    HIP_OK(hipMalloc((void**)&d_joints, sizeof(rz_mat4) * J * count));
    HIP_OK(hipMalloc((void**)&d_normals, sizeof(rz_mat3) * J * count));
    HIP_OK(hipMalloc((void**)&d_jobs, sizeof(PoseJob) * count));
    std::vector<PoseJob> jobs((size_t)count);
    for (int i = 0; i < count; i++) {
        jobs[i].skeleton = src->d_block;
        jobs[i].time = wrapped_time(src, frame_times[i]);
        jobs[i].pad = 0;
        jobs[i].joints = d_joints + (size_t)i * J * 16;
        jobs[i].normals = d_normals + (size_t)i * J * 9;
    }
    HIP_OK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(PoseJob) * count, hipMemcpyHostToDevice, g.stream));
    rz_pose_kernel<<<dim3((unsigned)count), dim3(64), 0, g.stream>>>(d_jobs);
    HIP_OK(hipGetLastError());
    if (joints) HIP_OK(hipMemcpyAsync(joints, d_joints, sizeof(rz_mat4) * J * count, hipMemcpyDeviceToHost, g.stream));
    if (normals) HIP_OK(hipMemcpyAsync(normals, d_normals, sizeof(rz_mat3) * J * count, hipMemcpyDeviceToHost, g.stream));
    HIP_OK(hipStreamSynchronize(g.stream));
    HIP_OK(hipFree(d_joints)); HIP_OK(hipFree(d_normals)); HIP_OK(hipFree(d_jobs));
}

/* the joint matrices a program's uniform block points at right now (nullptr: not skinned) */
static void pose_of(const rz_program *p, const rz_mat4 **joints, const rz_mat3 **joint_normals) {
    *joints = nullptr;
    *joint_normals = nullptr;
    if (p->family == RZ_FAMILY_BLINN) {
        *joints = ((const rz_blinn_uniforms*)p->uniforms)->joint_matrices;
        *joint_normals = ((const rz_blinn_uniforms*)p->uniforms)->joint_n_matrices;
    } else if (p->family == RZ_FAMILY_PBR) {
        *joints = ((const rz_pbr_uniforms*)p->uniforms)->joint_matrices;
        *joint_normals = ((const rz_pbr_uniforms*)p->uniforms)->joint_n_matrices;
    }
}

/* Whether one more triangle that refers to joints [0, need) can join the draw `last` (the most
 * recent record; its uniform block already compared equal, so the matrices sit at the same
 * addresses): the matrices that draw snapshotted must still be what the host holds -- the skeleton
 * rewrites them in place (skeleton.c:336-337) without telling anybody.  A triangle that refers to
 * more joints than the draw has seen so far extends the snapshot, which is the last thing in the
 * side buffer. */
static bool same_pose(const rz_program *p, Pending &last, int need) {
    if (last.num_joints == 0) return true;          /* the draw is not skinned: its uniform block says all */
    const rz_mat4 *joints;
    const rz_mat3 *joint_normals;
    pose_of(p, &joints, &joint_normals);
    if (!joints) return false;
    const size_t n = (size_t)last.num_joints;
    if (memcmp(g.side.data() + last.joints_at, joints, sizeof(rz_mat4) * n) != 0) return false;
    if (joint_normals && memcmp(g.side.data() + last.joint_normals_at, joint_normals, sizeof(rz_mat3) * n) != 0) return false;
    if ((size_t)need > n) {
        g.side.resize(last.joints_at);
        g.side.insert(g.side.end(), (const char*)joints, (const char*)joints + sizeof(rz_mat4) * need);
        last.joint_normals_at = g.side.size();
        if (joint_normals)
            g.side.insert(g.side.end(), (const char*)joint_normals, (const char*)joint_normals + sizeof(rz_mat3) * need);
        else
            g.side.insert(g.side.end(), sizeof(rz_mat3) * need, 0);
        while (g.side.size() % 16) g.side.push_back(0);
        last.num_joints = need;
    }
    return true;
}

extern "C" void graphics_draw_mesh_posed(rz_framebuffer *fp, rz_program *program, const rz_vertex *vertices, int num_faces,
                                         rz_pose_source *source, float frame_time, int attached) {
    Fb *fb = as_fb(fp);
    as_pose_source(source);
    if (program->family == RZ_FAMILY_SKYBOX) fatal("graphics_draw_mesh_posed: the skybox family has no skeleton");
    if (attached >= source->num_joints) fatal("graphics_draw_mesh_posed: attached to joint %d of %d", attached, source->num_joints);
    if (num_faces <= 0) return;
    Pending pd;
    pd.fb = fb;
    pd.program = program;
    pd.transient = false;
    g_ignore_host_joints = true;
    snapshot_uniforms(program, pd, 0);
    g_ignore_host_joints = false;
    Mesh &m = device_mesh(vertices, num_faces);
    if (attached < 0 && m.num_joints > source->num_joints)
        fatal("graphics_draw_mesh_posed: the mesh refers to joint %d, the skeleton has %d", m.num_joints - 1, source->num_joints);
    pd.rec.vertices = m.d_vertices;
    pd.rec.num_faces = num_faces;
    pd.pose = source;
    pd.pose_time = wrapped_time(source, frame_time);
    pd.pose_attached = attached;
    if (attached < 0) pd.rec.flags |= RZ_DRAW_SKINNED;
    g.pending.push_back(std::move(pd));
}

/* one more triangle of a draw recorded triangle by triangle (the last one of the list) */
static void append_transient(Pending &pd, const rz_vertex *tri) {
    if (pd.replay) {
        const Replay &r = *pd.replay;
        const size_t have = (size_t)pd.rec.num_faces;
        if (have < r.faces && memcmp(r.host.data() + 3 * have, tri, 3 * sizeof(rz_vertex)) == 0) return;
        /* the stream departs from the remembered one: what repeated it so far is recorded after all */
        g.transient_verts.insert(g.transient_verts.end(), r.host.begin(), r.host.begin() + 3 * have);
        pd.verts_count = 3 * have;
        pd.replay = nullptr;
    }
    g.transient_verts.insert(g.transient_verts.end(), tri, tri + 3);
    pd.verts_count += 3;
}

static const size_t REPLAY_MIN_FACES = 32;
static std::vector<Replay*> g_replays_to_upload;

/* at the end of a flush, while the recorded vertices are still there: remember the stream of a draw that
 * repeated none, for the next frame */
static void remember_stream(Pending &pd) {
    if (pd.replay) { pd.replay->misses = 0; return; }
    if (!g.replay_streams || pd.verts_count < 3 * REPLAY_MIN_FACES) return;
    Replay &r = g.replays[pd.program];
    if (r.made_epoch == g.flush_epoch) return;          /* an earlier draw of this flush made it (shadow pass and main pass; several frames) */
    const rz_vertex *verts = g.transient_verts.data() + pd.verts_at;
    if (r.d_vertices && r.faces * 3 >= pd.verts_count && memcmp(r.host.data(), verts, pd.verts_count * sizeof(rz_vertex)) == 0) return;
    if (r.misses >= 4) return;                          /* this program's vertices change all the time */
    r.misses++;
    r.made_epoch = g.flush_epoch;
    r.host.assign(verts, verts + pd.verts_count);
    r.faces = pd.verts_count / 3;
    g_replays_to_upload.push_back(&r);
}

/* ... and, once the flush has been handed over, their copies on the device (passes in flight may still
 * draw from the copy that is replaced) */
static void upload_remembered_streams() {
    for (Replay *r : g_replays_to_upload) {
        if (r->d_vertices) { sync_stream(); HIP_OK(hipFree(r->d_vertices)); r->d_vertices = nullptr; }
        HIP_OK(hipMalloc(&r->d_vertices, r->host.size() * sizeof(rz_vertex)));
        HIP_OK(hipMemcpy(r->d_vertices, r->host.data(), r->host.size() * sizeof(rz_vertex), hipMemcpyHostToDevice));
    }
    g_replays_to_upload.clear();
}

/* graphics.c:537: one triangle from the program's three attribute blocks.
 * Consecutive calls with an unchanged uniform block -- and, for a skinned
 * program, unchanged joint matrices -- extend one draw. */
extern "C" void graphics_draw_triangle(rz_framebuffer *fp, rz_program *program) {
    Fb *fb = (Fb*)fp;
    if (fb != g.checked_fb || !fb) g.checked_fb = fb = as_fb(fp);
    rz_vertex built[3];
    const rz_vertex *tri = built;
    if (program->family == RZ_FAMILY_PBR && program->attribs[1] == (char*)program->attribs[0] + sizeof(rz_vertex) &&
        program->attribs[2] == (char*)program->attribs[0] + 2 * sizeof(rz_vertex)) {
        tri = (const rz_vertex*)program->attribs[0];    /* the three blocks are one allocation and already have the stream's layout */
    } else {
        for (int j = 0; j < 3; j++) {
            rz_vertex &v = built[j];
            if (program->family == RZ_FAMILY_BLINN) {
                const rz_blinn_attribs *a = (const rz_blinn_attribs*)program->attribs[j];
                memset(&v, 0, sizeof v);
                v.position = a->position; v.texcoord = a->texcoord; v.normal = a->normal;
                v.joint = a->joint; v.weight = a->weight;
            } else if (program->family == RZ_FAMILY_PBR) {
                v = *(const rz_vertex*)program->attribs[j];
            } else {
                memset(&v, 0, sizeof v);
                v.position = ((const rz_skybox_attribs*)program->attribs[j])->position;
            }
        }
    }
    /* joints a skinned program's triangle refers to (1 + the largest index): only such a program's draw keeps a pose */
    int joints_needed = 1;
    {
        const rz_mat4 *joints;
        const rz_mat3 *joint_normals;
        pose_of(program, &joints, &joint_normals);
        if (joints) {
            int top = 0;
            for (int j = 0; j < 3; j++) {
                const rz_vec4 &q = tri[j].joint;
                top = std::max(top, std::max(std::max((int)q.x, (int)q.y), std::max((int)q.z, (int)q.w)));
            }
            joints_needed = top + 1;
        }
    }
    const char *ub = (const char*)program->uniforms;
    if (!g.pending.empty()) {
        Pending &last = g.pending.back();
        const bool copy_between = !g.tex_writes.empty() && g.tex_writes.back().after == g.pending.size();
        if (!copy_between && last.transient && last.fb == fb && last.program == program &&
            last.uniform_bytes.size() == (size_t)program->uniform_bytes &&
            memcmp(last.uniform_bytes.data(), ub, program->uniform_bytes) == 0 &&
            same_pose(program, last, joints_needed)) {
            append_transient(last, tri);
            last.rec.num_faces++;
            return;
        }
    }
    Pending pd;
    pd.fb = fb;
    pd.program = program;
    pd.transient = true;
    snapshot_uniforms(program, pd, joints_needed);      /* may flush what is recorded: the draw's vertices are appended after it */
    pd.uniform_bytes.assign(ub, ub + program->uniform_bytes);
    pd.verts_at = g.transient_verts.size();
    pd.rec.num_faces = 0;
    if (g.replay_streams) {
        auto remembered = g.replays.find(program);
        if (remembered != g.replays.end() && remembered->second.d_vertices) pd.replay = &remembered->second;
    }
    append_transient(pd, tri);
    pd.rec.num_faces = 1;
    g.pending.push_back(std::move(pd));
}

/* ------------------------------------------------------------------- flush -- */

static const size_t CONTROL_BYTES = 64;

static size_t ring_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > g.ring_size) {
        /* a flush larger than the ring (hundreds of thousands of triangles recorded one by one): a larger ring */
        sync_stream();
        size_t size = g.ring_size;
        while (size < bytes) size *= 2;
        HIP_OK(hipHostFree(g.ring_host));
        HIP_OK(hipFree(g.ring_dev));
        HIP_OK(hipHostMalloc((void**)&g.ring_host, size, hipHostMallocDefault));
        HIP_OK(hipMalloc((void**)&g.ring_dev, size));
        g.ring_size = size;
        g.ring_at = 0;
    }
    if (g.ring_at + bytes > g.ring_size) {
        sync_stream();              /* everything that used the ring has finished */
        g.ring_at = 0;
    }
    size_t at = g.ring_at;
    g.ring_at += bytes;
    return at;
}

/* layout of one pass inside the flush's upload region */
struct PassPlan {
    size_t first, last;             /* range in g.pending (while the flush is being put together) */
    size_t T;                       /* triangles */
    size_t off_draws, off_first;    /* offsets from the region start */
    std::vector<size_t> side_at, verts_at;
    bool has_blend, samples_gpu_texture;
    bool narrow_texels;             /* every blinn / pbr draw samples 4-byte images only */
    int families;                   /* bit f: draws of family f that go through the bins */
    size_t sky_first;               /* the pass's last sky_draws draws are sky boxes kept on its table of sky triangles: the first of them */
    int sky_draws;
    Fb *fb;
    int level;                      /* 0: depends on no other pass of the flush */
    size_t slot;                    /* its place in batch order: control block and PassParams of the pass */
};

static void plan_pass(PassPlan &pl, size_t &total) {
    const size_t D = pl.last - pl.first;
    pl.T = 0;
    pl.has_blend = pl.samples_gpu_texture = false;
    pl.narrow_texels = true;
    pl.families = 0;
    pl.fb = g.pending[pl.first].fb;
    for (size_t i = pl.first; i < pl.last; i++) {
        const Pending &pd = g.pending[i];
        pl.T += (size_t)pd.rec.num_faces;
        pl.has_blend |= (pd.rec.flags & RZ_DRAW_BLEND) != 0;
        pl.samples_gpu_texture |= !pd.gpu_textures.empty();
        if (pd.rec.family != RZ_FAMILY_SKYBOX && !(pd.rec.flags & RZ_DRAW_NARROW_TEXELS)) pl.narrow_texels = false;
    }
    /* sky boxes drawn last in a pass without blending (the frame driver's order, test_helper.c:480-506) stay out of the bins */
    pl.sky_first = pl.last;
    pl.sky_draws = 0;
    if (g.sky_table && !g.counting && !pl.has_blend)
        while (pl.sky_first > pl.first && pl.sky_draws < RZ_SKY_DRAWS) {
            const DrawRec &r = g.pending[pl.sky_first - 1].rec;
            if (r.family != RZ_FAMILY_SKYBOX || r.num_faces > RZ_SKY_FACES) break;
            pl.sky_first--;
            pl.sky_draws++;
        }
    /* (a shadow pass's fragments are black: rz_strip_kernel writes them itself, no family's kernel has work from them) */
    for (size_t i = pl.first; i < pl.sky_first; i++)
        if (pl.has_blend || !(g.pending[i].rec.flags & RZ_DRAW_SHADOW_PASS)) pl.families |= 1 << g.pending[i].rec.family;
    total = (total + 255) & ~(size_t)255;
    pl.off_draws = total;
    total += sizeof(DrawRec) * D;
    pl.off_first = total;
    total += sizeof(uint32_t) * (D + 1);
    total = (total + 15) & ~(size_t)15;
    pl.side_at.assign(D, 0);
    pl.verts_at.assign(D, 0);
    for (size_t i = 0; i < D; i++) {
        Pending &pd = g.pending[pl.first + i];
        if (pd.num_joints) {
            pl.side_at[i] = total;
            total += (sizeof(rz_mat4) + sizeof(rz_mat3)) * pd.num_joints;
            total = (total + 15) & ~(size_t)15;
        }
        if (pd.transient && !pd.replay) {
            pl.verts_at[i] = total;
            total += pd.verts_count * sizeof(rz_vertex);
            total = (total + 15) & ~(size_t)15;
        }
    }
}

/* where the flush being put together keeps its device-evaluated poses (graphics_flush) */
static char *g_pose_pool = nullptr;
static std::vector<size_t> g_pose_offsets;          /* per pose job: its matrices in the pool */
static size_t g_attach_jobs_at = 0, g_attach_jobs_filled = 0;

static void fill_pass(const PassPlan &pl, char *hb, char *db) {
    const size_t D = pl.last - pl.first;
    uint32_t *first_tri = (uint32_t*)(hb + pl.off_first);
    uint32_t running = 0;
    for (size_t i = 0; i < D; i++) {
        Pending &pd = g.pending[pl.first + i];
        DrawRec rec = pd.rec;
        if (pd.num_joints) {
            size_t jb = sizeof(rz_mat4) * pd.num_joints;
            memcpy(hb + pl.side_at[i], g.side.data() + pd.joints_at, jb);
            memcpy(hb + pl.side_at[i] + jb, g.side.data() + pd.joint_normals_at, sizeof(rz_mat3) * pd.num_joints);
            rec.joints = (const float*)(db + pl.side_at[i]);
            rec.joint_normals = (const float*)(db + pl.side_at[i] + jb);
            if (rec.flags & RZ_DRAW_SHADOW_PASS) rec.joint_normals = nullptr;
        }
        if (pd.transient && pd.replay) {
            rec.vertices = pd.replay->d_vertices;           /* the draw repeats (a prefix of) the remembered stream */
            g.replayed_draws++;
        } else if (pd.transient) {
            memcpy(hb + pl.verts_at[i], g.transient_verts.data() + pd.verts_at, pd.verts_count * sizeof(rz_vertex));
            rec.vertices = db + pl.verts_at[i];
        }
        if (pd.pose) {
            char *matrices = g_pose_pool + g_pose_offsets[pd.pose_job];
            if (pd.pose_attached < 0) {
                rec.joints = (const float*)matrices;
                rec.joint_normals = (rec.flags & RZ_DRAW_SHADOW_PASS) ? nullptr : (const float*)(matrices + sizeof(rz_mat4) * pd.pose->num_joints);
            } else {
                AttachJob job;
                job.draw = (DrawRec*)(db + pl.off_draws + sizeof(DrawRec) * i);
                job.joint = (const float*)matrices + 16 * (size_t)pd.pose_attached;
                memcpy(hb + g_attach_jobs_at + sizeof(AttachJob) * g_attach_jobs_filled++, &job, sizeof job);
            }
        }
        rec.sky_slot = pl.first + i >= pl.sky_first ? (int32_t)(pl.first + i - pl.sky_first) + 1 : 0;
        memcpy(hb + pl.off_draws + sizeof(DrawRec) * i, &rec, sizeof rec);
        first_tri[i] = running;
        running += (uint32_t)rec.num_faces;
    }
    first_tri[D] = running;
}

/* A batch: passes that do not depend on each other and run in ONE launch
 * sequence (grid y = pass), on one lane. */
struct Batch {
    std::vector<size_t> members;    /* indices into the flush's plan list, in submission order */
    int lane;
    size_t off_params;              /* PassParams[] of the members in the upload region */
    bool has_blend;
    int families;
    bool sky_table = false;         /* one of its passes keeps sky boxes on its table of sky triangles */
};

/* What is kept of a flush until its control blocks have been looked at: enough to run any of its
 * passes again.  Its region of the upload ring:
 *   [control block per pass][flush word][PassParams per pass][PassParams for passes run again][pass tables]
 * Control blocks and PassParams are in batch order (PassPlan::slot). */
struct FlushRecord {
    size_t at = 0;                                          /* the region */
    size_t flag_at = 0, params_at = 0, again_at = 0;        /* offsets in it */
    std::vector<PassPlan> plans;
    std::vector<Batch> batches;
    struct Copy { Tex *tex; Fb *fb; bool depth; size_t producer; };     /* producer: the plan whose pass rendered the source */
    std::vector<Copy> copies;
    /* poses evaluated on the device for this flush: jobs (PoseJob[]) and attach jobs (AttachJob[]) in the region */
    size_t pose_jobs_at = 0, pose_jobs = 0, attach_jobs_at = 0, attach_jobs = 0;
    bool used[MAX_LANES] = {false};
    unsigned long long epoch = 0;                           /* Context::flush_epoch when it was submitted */
    const uint32_t *control(size_t plan) const { return (const uint32_t*)(g.ring_host + at + CONTROL_BYTES * plans[plan].slot); }
    uint32_t *control_dev(size_t plan) const { return (uint32_t*)(g.ring_dev + at + CONTROL_BYTES * plans[plan].slot); }
};
static std::vector<FlushRecord> g_in_flight;

/* whether a batch of passes with blended draws takes the visibility-first route */
static bool splits_blend(size_t batch_tiles, size_t passes) {
    if (g.counting || g.split_blend == 0) return false;
    return g.split_blend == 1 || batch_tiles >= 16384 * passes;
}

/* Entries of the clipper's block table and of the bin lists a pass gets.  First time round an
 * estimate: a block per triangle up to 2^18, 16 list entries per triangle but at least 2^18 --
 * never more than a triangle in every bin with a full fan -- raised to what earlier passes turned out
 * to need (or set by mi355r_set_scratch_estimate).  A pass that is run again gets exactly what it
 * reported (control[0], control[1]). */
static void scratch_entries(const FlushRecord &fr, size_t plan, bool again, size_t *fan_cap, size_t *list_cap) {
    const PassPlan &pl = fr.plans[plan];
    if (again) {
        const uint32_t *c = fr.control(plan);
        *fan_cap = std::max<size_t>(c[0], 1);
        *list_cap = std::max<size_t>(c[1], 1);
        return;
    }
    const size_t nb = (size_t)((pl.fb->pub.width + RZ_BIN - 1) / RZ_BIN) * ((pl.fb->pub.height + RZ_BIN - 1) / RZ_BIN);
    if (g.forced_fan_estimate || g.forced_list_estimate) {
        *fan_cap = std::max<size_t>(std::min(pl.T, g.forced_fan_estimate ? g.forced_fan_estimate : (size_t)1 << 18), 1);
        *list_cap = std::max<size_t>(g.forced_list_estimate ? g.forced_list_estimate : std::max<size_t>(pl.T * 16, 1u << 18), 1);
        return;
    }
    *fan_cap = std::min(pl.T, std::max<size_t>(1u << 18, g.fan_highwater));
    *list_cap = std::min(pl.T * RZ_MAX_FAN * nb, std::max(std::max<size_t>(pl.T * 16, 1u << 18), g.list_highwater));
}

/* sizes the lane's scratch for the batch and (second call) writes the PassParams array.  PREPARE_FIRST:
 * a new flush.  PREPARE_REWRITE: the same flush once more, before anything of it was launched (the
 * lanes' scratch has moved or the estimates were raised): only the scratch part changes.
 * PREPARE_AGAIN: the batch holds passes of a finished flush that are run a second time, with scratch
 * of the sizes they reported. */
enum PrepareMode { PREPARE_FIRST, PREPARE_REWRITE, PREPARE_AGAIN };
static void prepare_batch(const Batch &b, const FlushRecord &fr, bool reserve_only, PrepareMode mode) {
    const bool again = mode == PREPARE_AGAIN;
    const std::vector<PassPlan> &plans = fr.plans;
    char *hb = g.ring_host + fr.at, *db = g.ring_dev + fr.at;
    Lane &L = g.lanes[b.lane];
    hipStream_t st = L.stream;
    size_t tris = 0, fans = 0, lists = 0, bins = 0, pixels = 0, matrix = 0, tiles = 0;
    std::vector<size_t> fan_cap, list_cap;
    for (size_t i : b.members) {
        const PassPlan &pl = plans[i];
        const size_t nb = (size_t)((pl.fb->pub.width + RZ_BIN - 1) / RZ_BIN) * ((pl.fb->pub.height + RZ_BIN - 1) / RZ_BIN);
        size_t fan, list;
        scratch_entries(fr, i, again, &fan, &list);
        fan_cap.push_back(fan);
        list_cap.push_back(list);
        tris += pl.T; fans += fan; lists += list; bins += nb;
        matrix += ((pl.T + 63) / 64) * nb;
        tiles += (size_t)((pl.fb->pub.width + RZ_TILE_W - 1) / RZ_TILE_W) * ((pl.fb->pub.height + RZ_TILE_H - 1) / RZ_TILE_H);
        pixels += (size_t)pl.fb->pub.width * pl.fb->pub.height;
    }
    if (lists > 0xffffffffull || fans * RZ_MAX_FAN > 0x7fffffffull)
        fatal("a batch of %zu passes needs %zu bin-list entries and %zu clipper blocks: beyond the 32-bit offsets of its tables "
              "(fewer passes per batch: MI355R_BATCH)", b.members.size(), lists, fans);
    reserve(L.clip, tris * 3 * 4, st);
    reserve(L.vary, tris * 3 * RZ_VARY_STRIDE, st);
    reserve(L.tri, tris, st);
    reserve(L.bbox, tris, st);
    reserve(L.clip_list, tris, st);
    reserve(L.fan_tri, fans * RZ_MAX_FAN, st);
    reserve(L.fan_vary, fans * RZ_MAX_POLY * RZ_VARY_STRIDE, st);
    reserve(L.bin_range, bins, st);
    reserve(L.sky_bins, bins, st);
    reserve(L.sky_table, b.members.size() * RZ_SKY_TABLE, st);
    reserve(L.bin_class, bins, st);
    reserve(L.bin_counts, matrix, st);
    /* tiles are dealt to the parts of a list by their index: a part holds at most its even share */
    /* (32x8 tiles, or the two 16x8 blocks of each that rz_strip_kernel lists) */
    const size_t queue_capacity = (2 * tiles + RZ_QUEUE_WAYS - 1) / RZ_QUEUE_WAYS + b.members.size();    /* every pass's share rounded up */
    reserve(L.tile_listed, tiles, st);
    reserve(L.queue, RZ_LISTS * RZ_QUEUE_WAYS * queue_capacity, st);    /* blinn, pbr, RZ_ORDERED_LIST, RZ_VISIBILITY_LIST */
    reserve(L.queue_counts, RZ_LISTS * RZ_QUEUE_WAYS * RZ_QUEUE_PAD, st);
    reserve(L.bin_list, lists, st);
    reserve(L.bin_boxes, lists, st);
    if (!b.has_blend || splits_blend(tiles, b.members.size())) reserve(L.ref, pixels, st);
    if (reserve_only) return;

    size_t tri_off = 0, fan_off = 0, list_off = 0, bin_off = 0, ref_off = 0, matrix_off = 0, tile_off = 0;
    for (size_t k = 0; k < b.members.size(); k++) {
        const size_t i = b.members[k];
        const PassPlan &pl = plans[i];
        Fb *fb = pl.fb;
        const int W = fb->pub.width, H = fb->pub.height;
        PassParams P;
        if (mode != PREPARE_FIRST) {
            /* as it was submitted: target, pending clears, tables -- only the scratch below changes;
             * a pass run again waits for nothing of its flush */
            memcpy(&P, hb + fr.params_at + sizeof(PassParams) * pl.slot, sizeof P);
            if (again) P.level = 0;
        } else {
            memset(&P, 0, sizeof P);
            P.color = fb->d_color;
            P.depth = fb->d_depth;
            P.width = W;
            P.height = H;
            P.clear_color_pending = fb->clear_color_pending;
            P.clear_depth_pending = fb->clear_depth_pending;
            P.clear_color = fb->clear_color;
            P.clear_depth = fb->clear_depth;
            P.draws = (const DrawRec*)(db + pl.off_draws);
            P.first_tri = (const uint32_t*)(db + pl.off_first);
            P.num_draws = (int)(pl.last - pl.first);
            P.num_tris = (int)pl.T;
            P.control = (uint32_t*)(db + CONTROL_BYTES * pl.slot);
            P.flush_flag = (uint32_t*)(db + fr.flag_at);
            P.level = pl.level;
            P.draws_off = pl.off_draws;
            fb->clear_color_pending = fb->clear_depth_pending = false;
            g.host_counters[0] += pl.T;
            g.host_counters[4] += 1;
            g.host_counters[5] += (unsigned long long)(pl.last - pl.first);
        }
        P.clip = L.clip.ptr + tri_off * 3 * 4;
        P.vary = L.vary.ptr + tri_off * 3 * RZ_VARY_STRIDE;
        P.tri = L.tri.ptr + tri_off;
        P.bbox = L.bbox.ptr + tri_off;
        P.clip_list = L.clip_list.ptr + tri_off;
        P.fan_tri = L.fan_tri.ptr + fan_off * RZ_MAX_FAN;
        P.fan_vary = L.fan_vary.ptr + fan_off * RZ_MAX_POLY * RZ_VARY_STRIDE;
        P.fan_capacity = (int)std::min<size_t>(fan_cap[k], 0x7fffffff);
        P.bins_x = (W + RZ_BIN - 1) / RZ_BIN;
        P.bins_y = (H + RZ_BIN - 1) / RZ_BIN;
        P.bin_range = L.bin_range.ptr + bin_off;
        P.bin_counts = L.bin_counts.ptr + matrix_off;
        P.bin_list = L.bin_list.ptr + list_off;
        P.bin_boxes = L.bin_boxes.ptr + list_off;
        P.bin_capacity = (uint32_t)list_cap[k];
        P.ref = L.ref.ptr ? L.ref.ptr + ref_off : nullptr;
        P.sky_draws = pl.sky_draws;
        P.sky_table = L.sky_table.ptr + k * RZ_SKY_TABLE;
        P.counters = g.counting ? g.d_counters : nullptr;
        P.scan_list = !b.has_blend ? -1 : splits_blend(tiles, b.members.size()) ? (int)RZ_VISIBILITY_LIST : (int)RZ_ORDERED_LIST;
        P.micro_area = g.micro_area;
        P.micro_min_list = g.micro_min_list;
        P.tiles_x = (W + RZ_TILE_W - 1) / RZ_TILE_W;
        P.tiles_y = (H + RZ_TILE_H - 1) / RZ_TILE_H;
        P.tile_listed = L.tile_listed.ptr + tile_off;
        P.queue = L.queue.ptr;
        P.queue_counts = L.queue_counts.ptr;
        P.queue_capacity = (uint32_t)queue_capacity;
        P.tile_off = (uint32_t)tile_off;
        tile_off += (size_t)P.tiles_x * P.tiles_y;
        P.tri_off = (uint32_t)tri_off;
        P.fan_off = (uint32_t)fan_off;
        P.list_off = (uint32_t)list_off;
        P.bin_off = (uint32_t)bin_off;
        P.ref_off = ref_off;
        memcpy(hb + b.off_params + sizeof(PassParams) * k, &P, sizeof P);
        g.last_pass = P;
        g.have_last_pass = true;
        {
            size_t before_sky = 0;
            for (size_t d = pl.first; d < pl.sky_first; d++) before_sky += (size_t)g.pending[d].rec.num_faces;
            g.last_pass_sky_first = (int)before_sky;
        }
        tri_off += pl.T; fan_off += fan_cap[k]; list_off += list_cap[k];
        bin_off += (size_t)P.bins_x * P.bins_y;
        matrix_off += ((pl.T + 63) / 64) * (size_t)P.bins_x * P.bins_y;
        ref_off += (size_t)W * H;
    }
}

static void launch_batch(const Batch &b, const std::vector<PassPlan> &plans, char *db) {
    Lane &L = g.lanes[b.lane];
    hipStream_t st = L.stream;
    const unsigned n = (unsigned)b.members.size();
    size_t max_T = 0, max_bins = 0, max_tiles_x = 0, max_tiles_y = 0;
    for (size_t i : b.members) {
        const PassPlan &pl = plans[i];
        const int W = pl.fb->pub.width, H = pl.fb->pub.height;
        max_T = std::max(max_T, pl.T);
        max_bins = std::max(max_bins, (size_t)((W + RZ_BIN - 1) / RZ_BIN) * ((H + RZ_BIN - 1) / RZ_BIN));
        max_tiles_x = std::max(max_tiles_x, (size_t)((W + RZ_TILE_W - 1) / RZ_TILE_W));
        max_tiles_y = std::max(max_tiles_y, (size_t)((H + RZ_TILE_H - 1) / RZ_TILE_H));
    }
    const PassParams *passes = (const PassParams*)(db + b.off_params);
    static const bool paranoid = getenv("MI355R_SYNC") != nullptr;     /* debugging: fence + check after every kernel */
    int stage = 0;
    auto stamp = [&]() {
        if (paranoid) {
            hipError_t err = hipStreamSynchronize(st);
            if (err != hipSuccess)
                fatal("kernel stage %d of the batch starting at pass %zu on lane %d failed: %s", stage, b.members[0], b.lane, hipGetErrorString(err));
            stage++;
        }
        if (!g.timing) return;
        hipEvent_t e;
        HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventRecord(e, st));
        g.timing_events.push_back(e);
    };
    const dim3 vgrid((unsigned)((max_T + RZ_VS_TRIS - 1) / RZ_VS_TRIS), n), tgrid((unsigned)((max_T + 255) / 256), n);
    const dim3 tiles((unsigned)max_tiles_x, (unsigned)max_tiles_y, n);
    /* tile_order (kernels.h): whole bins, i.e. an even number of tile columns and a multiple of a bin's tile rows */
    const unsigned bin_columns = (unsigned)(max_tiles_x + 1) / 2, bin_rows = (unsigned)(max_tiles_y * RZ_TILE_H + RZ_BIN - 1) / RZ_BIN;
    /* the shade kernels stride over work lists: enough workgroups to fill the device, no more */
    size_t batch_tiles = 0;
    for (size_t i : b.members)
        batch_tiles += (size_t)((plans[i].fb->pub.width + RZ_TILE_W - 1) / RZ_TILE_W) * ((plans[i].fb->pub.height + RZ_TILE_H - 1) / RZ_TILE_H);
    const uint32_t queue_capacity = (uint32_t)((2 * batch_tiles + RZ_QUEUE_WAYS - 1) / RZ_QUEUE_WAYS + b.members.size());
    const dim3 shade_grid((unsigned)std::min<size_t>(max_tiles_x * max_tiles_y * n, (size_t)g.shade_workgroups));
    {
        /* what the kernels below index without a check of their own: wrong sizes here must stop the host, not fault the GPU */
        size_t pixels = 0, tris = 0;
        size_t bins = 0;
        for (size_t i : b.members) {
            pixels += (size_t)plans[i].fb->pub.width * plans[i].fb->pub.height;
            tris += plans[i].T;
            bins += (size_t)((plans[i].fb->pub.width + RZ_BIN - 1) / RZ_BIN) * ((plans[i].fb->pub.height + RZ_BIN - 1) / RZ_BIN);
        }
        if (L.bin_range.cap < bins || L.bin_class.cap < bins || L.sky_bins.cap < bins || L.sky_table.cap < b.members.size() * RZ_SKY_TABLE) fatal("per-bin scratch of lane %d is smaller than the batch it is about to run", b.lane);
        const bool uses_ref = !(b.has_blend && !splits_blend(batch_tiles, n));
        if (L.queue_counts.cap < RZ_LISTS * RZ_QUEUE_WAYS * RZ_QUEUE_PAD || L.queue.cap < RZ_LISTS * RZ_QUEUE_WAYS * (size_t)queue_capacity ||
            L.tile_listed.cap < batch_tiles || L.bin_boxes.cap < L.bin_list.cap || (uses_ref && (!L.ref.ptr || L.ref.cap < pixels)) || L.tri.cap < tris || L.bbox.cap < tris)
            fatal("scratch of lane %d is smaller than the batch it is about to run", b.lane);
    }
    stamp();
    rz_begin_kernel<<<dim3(64, n), dim3(256), 0, st>>>(passes);
    rz_vertex_kernel<<<vgrid, dim3(256), 0, st>>>(passes);      /* with the triangles' set-up */
    stamp();
    rz_clip_kernel<<<dim3(256, n), dim3(64), 0, st>>>(passes);
    stamp();
    rz_bin_count_kernel<<<tgrid, dim3(256), 0, st>>>(passes);
    rz_bin_scan_kernel<<<dim3((unsigned)((max_bins + 63) / 64), n), dim3(64 * RZ_SCAN_SEGMENTS), 0, st>>>(passes);
    {
        /* a chunk whose triangles lie far apart walks its bins in rounds of 64: with few chunks in the launch (a frame
         * or two) the rounds are shared out over four waves, with many one wave per chunk keeps more chunks in flight */
        const dim3 grid((unsigned)((max_T + 63) / 64 * n));
        const int waves = g.fill_waves ? g.fill_waves : grid.x < 4096u ? 4 : 1;
        if (waves == 1) rz_bin_fill_kernel<1><<<grid, dim3(64), 0, st>>>(passes, n);
        else if (waves == 2) rz_bin_fill_kernel<2><<<grid, dim3(128), 0, st>>>(passes, n);
        else rz_bin_fill_kernel<4><<<grid, dim3(256), 0, st>>>(passes, n);
    }
    stamp();
    bool narrow = true;
    for (size_t i : b.members) narrow = narrow && plans[i].narrow_texels;
    /* a pass with blended draws: either everything in the ordered kernel (visibility and shading
     * together, fragment stacks), or visibility first, the ordered kernel for the tiles where
     * blended fragments pile up and the shade kernels for the rest */
    const int only = b.families == (1 << RZ_FAMILY_BLINN) ? 0 : b.families == (1 << RZ_FAMILY_PBR) ? 1 : -1;
    bool in_strips = false;                         /* the shade lists hold 16x8 blocks */
    if (b.has_blend && !splits_blend(batch_tiles, n)) {
        /* over the tiles of the bins that are not empty (listed by rz_bin_scan_kernel) */
        auto kernel = rz_raster_list_kernel<-1, false>;
        if (only == 0) kernel = narrow ? rz_raster_list_kernel<0, true> : rz_raster_list_kernel<0, false>;
        if (only == 1) kernel = narrow ? rz_raster_list_kernel<1, true> : rz_raster_list_kernel<1, false>;
        kernel<<<shade_grid, dim3(256), 0, st>>>(
            passes, L.tri.ptr, L.fan_tri.ptr, L.bin_list.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr,
            L.queue.ptr, L.queue_counts.ptr, queue_capacity);
        snprintf(g.last_raster_kernel, sizeof g.last_raster_kernel, "rz_raster_list_kernel<%d, %s>", only, narrow && only >= 0 ? "true" : "false");
        stamp(); stamp(); stamp(); stamp();
    } else {
        if (b.has_blend) {
            rz_blend_visibility_kernel<<<shade_grid, dim3(128), 0, st>>>(passes, L.bin_list.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr,
                                                                     L.queue.ptr, L.queue_counts.ptr, queue_capacity);
            auto kernel = rz_raster_list_kernel<-1, false>;
            if (only == 0) kernel = narrow ? rz_raster_list_kernel<0, true> : rz_raster_list_kernel<0, false>;
            if (only == 1) kernel = narrow ? rz_raster_list_kernel<1, true> : rz_raster_list_kernel<1, false>;
            kernel<<<shade_grid, dim3(256), 0, st>>>(
                passes, L.tri.ptr, L.fan_tri.ptr, L.bin_list.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr,
                L.queue.ptr, L.queue_counts.ptr, queue_capacity);
            snprintf(g.last_raster_kernel, sizeof g.last_raster_kernel, "rz_blend_visibility_kernel");
        } else {
            /* visibility, a strip of 16x8 blocks per wave */
            in_strips = true;
            uint32_t longest = 0;
            for (size_t i : b.members) longest = std::max(longest, plans[i].fb->longest_list);
            /* tall strips save workgroup starts, which pays when the launch has many times more workgroups than the GPU holds
             * at once; a launch of one or two passes (a viewer's frame-at-a-time loop) is as long as its longest wave, and
             * that wave is four times shorter with one block per wave (helmet, one frame per flush: 98 -> 43 us) */
            const int few = b.members.size() <= 2 ? 1 : b.members.size() <= 4 ? 2 : 8;
            const int strip = !g.strip_long ? g.strip : std::min(few, longest >= g.strip_long ? 1 : longest >= g.strip_long / 2 ? std::min(g.strip, 2) : g.strip);
            const unsigned strips_per_bin = RZ_BIN / (8u * strip);
            auto kernel = strip == 8 ? rz_strip_kernel<8> : strip == 4 ? rz_strip_kernel<4> : strip == 2 ? rz_strip_kernel<2> : rz_strip_kernel<1>;
            /* empty bins are finished by the kernel that says so; a pass with a table of sky triangles gets its bins' records for rz_sky_kernel */
            rz_bin_class_kernel<<<dim3((unsigned)max_bins, n), dim3(64), 0, st>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.bin_range.ptr,
                                                                             L.sky_bins.ptr, L.bin_class.ptr);
            if (g.verbose) fprintf(stderr, "mi355r: batch of %u passes, longest bin list last time %u entries, strips of %d\n", n, longest, strip);
            snprintf(g.last_raster_kernel, sizeof g.last_raster_kernel, "rz_strip_kernel<%d>", strip);
            /* passes of one size: a strip then finds its bin's class without reading its pass's parameters */
            int uniform_w = plans[b.members[0]].fb->pub.width, uniform_h = plans[b.members[0]].fb->pub.height;
            for (size_t i : b.members)
                if (plans[i].fb->pub.width != uniform_w || plans[i].fb->pub.height != uniform_h) uniform_w = uniform_h = 0;
            kernel<<<dim3(n, bin_columns * 2, bin_rows * strips_per_bin), dim3(128), 0, st>>>(
                passes, L.bin_list.ptr, L.bin_boxes.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr, L.bin_class.ptr, uniform_w, uniform_h);
        }
        stamp();
        if (b.sky_table)
            rz_sky_kernel<2><<<tiles, dim3(128), 0, st>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.sky_bins.ptr);
        if (b.families & (1 << RZ_FAMILY_SKYBOX)) {     /* skybox fragments that came through the bins */
            auto kernel = in_strips ? rz_shade_list_kernel<2, false, true> : rz_shade_list_kernel<2, false, false>;
            kernel<<<shade_grid, dim3(256), 0, st>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.ref.ptr,
                                                     L.queue.ptr, L.queue_counts.ptr, queue_capacity);
        }
        stamp();
        if (b.families & (1 << RZ_FAMILY_BLINN)) {
            auto kernel = in_strips ? (narrow ? rz_shade_list_kernel<0, true, true> : rz_shade_list_kernel<0, false, true>)
                                    : (narrow ? rz_shade_list_kernel<0, true, false> : rz_shade_list_kernel<0, false, false>);
            kernel<<<shade_grid, dim3(256), 0, st>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.ref.ptr,
                                                     L.queue.ptr, L.queue_counts.ptr, queue_capacity);
        }
        stamp();
        if (b.families & (1 << RZ_FAMILY_PBR)) {
            auto kernel = in_strips ? (narrow ? rz_shade_list_kernel<1, true, true> : rz_shade_list_kernel<1, false, true>)
                                    : (narrow ? rz_shade_list_kernel<1, true, false> : rz_shade_list_kernel<1, false, false>);
            kernel<<<shade_grid, dim3(256), 0, st>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.ref.ptr,
                                                     L.queue.ptr, L.queue_counts.ptr, queue_capacity);
        }
        stamp();
    }
    HIP_OK(hipGetLastError());
    if (g.timing) g.timing_passes.push_back(n);
}

/* one framebuffer -> texture copy job (texture.c:130-158); `producer`: the control block of the pass
 * that rendered the source, on the device, or nullptr */
static CopyJob copy_job(Tex *t, Fb *fb, bool depth, const uint32_t *producer) {
    CopyJob j;
    memset(&j, 0, sizeof j);
    j.from = depth ? (const void*)fb->d_depth : (const void*)fb->d_color;
    j.to = t->d_data;
    j.w = t->pub.width; j.h = t->pub.height;
    j.flags = t->flags; j.is_depth = depth ? 1 : 0;
    j.producer = producer;
    return j;
}

/* Runs again, in submission order and one batch at a time, the passes of a finished flush that did
 * not run: those whose scratch was too small -- their control blocks hold what they need -- and
 * those that were left out because they might have depended on one of them; then the recorded
 * copies out of their framebuffers.  Every stream is idle when this is called and again when it
 * returns. */
static void run_again(FlushRecord &fr) {
    char *hb = g.ring_host + fr.at, *db = g.ring_dev + fr.at;
    size_t again_used = 0, passes_again = 0;
    for (const Batch &b : fr.batches) {
        Batch r;
        r.lane = b.lane;
        r.has_blend = b.has_blend;
        r.families = 0;
        for (size_t i : b.members) {
            const uint32_t *c = fr.control(i);
            if (c[2] & RZ_ERR_QUEUE_OVERFLOW) fatal("shade work list overflow (internal error)");
            if (c[2] == 0) continue;
            r.members.push_back(i);
            r.families |= fr.plans[i].families;
            r.sky_table = r.sky_table || fr.plans[i].sky_draws > 0;
            if (c[2] & RZ_ERR_FAN_OVERFLOW) g.fan_highwater = std::max<size_t>(g.fan_highwater, c[0]);
            if (c[2] & RZ_ERR_BIN_OVERFLOW) g.list_highwater = std::max<size_t>(g.list_highwater, c[1]);
        }
        if (r.members.empty()) continue;
        r.off_params = fr.again_at + sizeof(PassParams) * again_used;
        again_used += r.members.size();
        passes_again += r.members.size();
        Lane &L = g.lanes[r.lane];
        /* A pass that ran out of clipper blocks did not bin the triangles it had no block for, so the
         * list entries it reported are too few: its second run has enough blocks, reports the true
         * number of entries and may need a third.  Only the passes that still fail are launched again:
         * a blended pass carries no clear, so rendering it twice would blend it twice. */
        const std::vector<size_t> ran = r.members;
        for (int attempt = 0;; attempt++) {
            prepare_batch(r, fr, true, PREPARE_AGAIN);
            prepare_batch(r, fr, false, PREPARE_AGAIN);
            HIP_OK(hipMemcpyAsync(db + r.off_params, hb + r.off_params, sizeof(PassParams) * r.members.size(), hipMemcpyHostToDevice, L.stream));
            launch_batch(r, fr.plans, db);
            for (size_t i : r.members)
                HIP_OK(hipMemcpyAsync(hb + CONTROL_BYTES * fr.plans[i].slot, db + CONTROL_BYTES * fr.plans[i].slot, CONTROL_BYTES,
                                      hipMemcpyDeviceToHost, L.stream));
            HIP_OK(hipStreamSynchronize(L.stream));
            std::vector<size_t> still;
            int families = 0;
            for (size_t i : r.members) {
                const uint32_t *c = fr.control(i);
                if (c[2] == 0) continue;
                still.push_back(i);
                families |= fr.plans[i].families;
                if (c[2] & RZ_ERR_BIN_OVERFLOW) g.list_highwater = std::max<size_t>(g.list_highwater, c[1]);
            }
            if (still.empty()) break;
            if (attempt == 2)
                for (size_t i : still)
                    fatal("a pass run again with the scratch it asked for (%u clipper blocks, %u list entries) failed again: error bits %u",
                          fr.control(i)[0], fr.control(i)[1], fr.control(i)[2]);
            r.members = still;
            r.families = families;
        }
        for (const FlushRecord::Copy &c : fr.copies) {
            if (std::find(ran.begin(), ran.end(), c.producer) == ran.end()) continue;
            CopyJobs jobs;
            memset(&jobs, 0, sizeof jobs);
            jobs.job[0] = copy_job(c.tex, c.fb, c.depth, nullptr);
            rz_buffers_to_textures_kernel<<<dim3((c.tex->pub.width + 15) / 16, (c.tex->pub.height + 15) / 16, 1), dim3(256), 0, L.stream>>>(jobs);
            HIP_OK(hipGetLastError());
        }
        HIP_OK(hipStreamSynchronize(L.stream));
    }
    g.passes_run_again += passes_again;
    if (g.verbose) fprintf(stderr, "mi355r: %zu passes of a flush ran again with larger scratch\n", passes_again);
}

/* Looks at the control blocks of the flushes in flight -- waiting for their lanes unless the caller
 * knows them idle -- and runs again what did not run. */
/* framebuffers_read_device_rgb_begin .. _end: the copies of a set of frames that the host has not waited for yet */
struct FramesRead {
    bool active = false;
    std::vector<Fb*> fbs;
    char *rgb = nullptr;
    size_t stride = 0;
    hipEvent_t done[MAX_LANES];
    bool used[MAX_LANES] = {false};
    unsigned long long before_epoch = 0;        /* flushes submitted before the copies: those with a smaller epoch */
};
static FramesRead g_frames_read;
static void launch_frames_read(bool again);

static bool settle_in_flight(bool lanes_idle) {
    bool ran_again = false, ran_again_before_read = false;
    while (!g_in_flight.empty()) {
        FlushRecord fr = std::move(g_in_flight.back());
        g_in_flight.pop_back();
        if (!lanes_idle)
            for (int l = 0; l < MAX_LANES; l++) if (fr.used[l]) HIP_OK(hipStreamSynchronize(g.lanes[l].stream));
        bool failed = false;
        for (size_t i = 0; i < fr.plans.size(); i++) fr.plans[i].fb->longest_list = 0;
        for (size_t i = 0; i < fr.plans.size(); i++) {
            const uint32_t *c = fr.control(i);
            failed = failed || c[2] != 0;
            g.host_counters[3] += c[1];
            fr.plans[i].fb->longest_list = std::max(fr.plans[i].fb->longest_list, c[4]);
        }
        if (failed) {
            if (!lanes_idle) {      /* other lanes may be busy with later work on the same framebuffers and textures */
                for (int i = 0; i < g.num_lanes; i++) HIP_OK(hipStreamSynchronize(g.lanes[i].stream));
                HIP_OK(hipStreamSynchronize(g.copy_stream));
            }
            run_again(fr);
            ran_again = true;
            if (g_frames_read.active && fr.epoch < g_frames_read.before_epoch) ran_again_before_read = true;
        }
    }
    /* frames that were copied out behind passes which had not run: copied again behind the passes' second run.
     * A flush is settled before the next one's first launch, so no later pass has touched the framebuffers yet;
     * clears the host has asked for since are only flags until a pass consumes them, and stay flags here */
    if (ran_again_before_read) launch_frames_read(true);
    return ran_again;
}

static void check_passes_in_flight() { (void)settle_in_flight(true); }

/* A framebuffer is about to be handed out and its lane has drained: its own passes of the flush in
 * flight either all ran (their control blocks came back on that lane) or the flush is settled now. */
static void settle_framebuffer(Fb *fb) {
    for (const FlushRecord &fr : g_in_flight)
        for (size_t i = 0; i < fr.plans.size(); i++)
            if (fr.plans[i].fb == fb && fr.control(i)[2] != 0) { settle_in_flight(false); return; }
}

/* framebuffer_read_device_rgb for many framebuffers at once (the frames of a step, about to be gathered): their copies are
 * launched behind the rendering on each framebuffer's own lane and the host waits once, for all lanes,
 * instead of once or twice per frame.  Should one of their passes turn out not to have run (out of
 * scratch), the flush is settled and the copies are made again. */
extern "C" void framebuffers_read_device_rgb(rz_framebuffer *const *fps, int count, void *rgb, size_t stride_bytes) {
    if (count <= 0) return;
    if (!rgb || ((uintptr_t)rgb & 3u) || (stride_bytes & 3u))
        fatal("framebuffers_read_device_rgb: destination and stride must be 4-byte aligned (%p, %zu)", rgb, stride_bytes);
    std::vector<Fb*> fbs((size_t)count);
    for (int i = 0; i < count; i++) {
        fbs[i] = as_fb(fps[i]);
        if ((size_t)fbs[i]->pub.width * fbs[i]->pub.height * 3 > stride_bytes) fatal("framebuffers_read_device_rgb: frame %d is larger than the stride", i);
    }
    graphics_flush();
    for (int attempt = 0; attempt < 2; attempt++) {
        for (int i = 0; i < count; i++) {
            Fb *fb = fbs[i];
            apply_pending_clear(fb);
            const size_t n = (size_t)fb->pub.width * fb->pub.height;
            hipStream_t st = g.lanes[fb->lane < 0 ? 0 : fb->lane].stream;
            rz_strip_alpha_kernel<<<dim3((unsigned)((n / 4 + 1 + 255) / 256)), dim3(256), 0, st>>>(
                fb->d_color, (uint32_t*)((char*)rgb + (size_t)i * stride_bytes), n);
            HIP_OK(hipGetLastError());
        }
        for (int l = 0; l < g.num_lanes; l++) HIP_OK(hipStreamSynchronize(g.lanes[l].stream));
        bool left_out = false;
        for (const FlushRecord &fr : g_in_flight)
            for (size_t k = 0; k < fr.plans.size() && !left_out; k++)
                if (fr.control(k)[2] != 0 && std::find(fbs.begin(), fbs.end(), fr.plans[k].fb) != fbs.end()) left_out = true;
        if (!left_out) return;
        settle_in_flight(false);
    }
}

/* The same in two halves, for a caller that has something to do while the frames are rendered and copied (a
 * render loop that records its next step): _begin submits what is recorded and launches the copies behind the
 * rendering, each on its framebuffer's lane, and returns; _end waits for those copies -- not for what was
 * submitted after them.  Later passes into the same framebuffers come after the copies in stream order.  Passes
 * that turn out not to have run are run again when the flush is settled (at the latest before the next flush's
 * first launch), and the copies are launched again behind them (settle_in_flight).  One set at a time. */
/* again: the copies are launched a second time, behind passes that have just run again.  A clear that is pending
 * now was asked for AFTER _begin -- it belongs to the caller's next step (a render loop clears and records its next
 * frames while these copies are outstanding) -- and must not reach the frame that is being copied out. */
static void launch_frames_read(bool again) {
    FramesRead &r = g_frames_read;
    /* (a clear on its own settles the flush in flight first, which may launch the copies: before the events below) */
    if (!again) for (Fb *fb : r.fbs) apply_pending_clear(fb);
    for (int l = 0; l < MAX_LANES; l++) if (r.used[l]) { HIP_OK(hipEventDestroy(r.done[l])); r.used[l] = false; }
    for (size_t i = 0; i < r.fbs.size(); i++) {
        Fb *fb = r.fbs[i];
        const size_t n = (size_t)fb->pub.width * fb->pub.height;
        const int lane = fb->lane < 0 ? 0 : fb->lane;
        rz_strip_alpha_kernel<<<dim3((unsigned)((n / 4 + 1 + 255) / 256)), dim3(256), 0, g.lanes[lane].stream>>>(
            fb->d_color, (uint32_t*)(r.rgb + i * r.stride), n);
        HIP_OK(hipGetLastError());
        r.used[lane] = true;
    }
    for (int l = 0; l < MAX_LANES; l++)
        if (r.used[l]) {
            HIP_OK(hipEventCreateWithFlags(&r.done[l], hipEventDisableTiming));
            HIP_OK(hipEventRecord(r.done[l], g.lanes[l].stream));
        }
}

extern "C" void framebuffers_read_device_rgb_begin(rz_framebuffer *const *fps, int count, void *rgb, size_t stride_bytes) {
    FramesRead &r = g_frames_read;
    if (r.active) fatal("framebuffers_read_device_rgb_begin: the set before has not been waited for (framebuffers_read_device_rgb_end)");
    if (count <= 0) return;
    if (!rgb || ((uintptr_t)rgb & 3u) || (stride_bytes & 3u))
        fatal("framebuffers_read_device_rgb_begin: destination and stride must be 4-byte aligned (%p, %zu)", rgb, stride_bytes);
    r.fbs.assign((size_t)count, nullptr);
    for (int i = 0; i < count; i++) {
        r.fbs[i] = as_fb(fps[i]);
        if ((size_t)r.fbs[i]->pub.width * r.fbs[i]->pub.height * 3 > stride_bytes) fatal("framebuffers_read_device_rgb_begin: frame %d is larger than the stride", i);
    }
    r.rgb = (char*)rgb;
    r.stride = stride_bytes;
    graphics_flush();
    r.before_epoch = g.flush_epoch;
    r.active = true;
    launch_frames_read(false);
}

extern "C" void framebuffers_read_device_rgb_end(void) {
    FramesRead &r = g_frames_read;
    if (!r.active) return;
    for (int attempt = 0; attempt < 2; attempt++) {
        for (int l = 0; l < MAX_LANES; l++) if (r.used[l]) HIP_OK(hipEventSynchronize(r.done[l]));
        /* their flush may still be the one in flight (nothing was submitted since): look at its control blocks now */
        bool left_out = false;
        for (const FlushRecord &fr : g_in_flight)
            for (size_t k = 0; fr.epoch < r.before_epoch && k < fr.plans.size() && !left_out; k++)
                if (fr.control(k)[2] != 0 && std::find(r.fbs.begin(), r.fbs.end(), fr.plans[k].fb) != r.fbs.end()) left_out = true;
        if (!left_out) break;
        settle_in_flight(false);            /* runs them again and launches the copies again */
    }
    for (int l = 0; l < MAX_LANES; l++) if (r.used[l]) { HIP_OK(hipEventDestroy(r.done[l])); r.used[l] = false; }
    r.active = false;
    r.fbs.clear();
}

/* ---- texture_sync without allocations ---------------------------------------
 * texture_sync above stages the row-major float4 image in a buffer it allocates and frees per call.
 * Callers that cannot allocate device memory (a viewer's loop that shows the shadow map every frame,
 * harnesses that forbid hipMalloc) get the same result through a staging
 * area that is part of the library's image on the device: a file-scope __device__ array of
 * RZ_SYNC_SCRATCH_TEXELS float4 texels.  Every GPU the library runs on holds its own copy of that
 * array, at its own address, until the process ends; the address is looked up once per device
 * (scratch_on_device) and only that pointer is ever handed to kernels and copies.  Every call on a
 * GPU uses the same array: calls that overlap on one GPU would race (the library is single-threaded,
 * like the reference).  Nothing in it has to start at zero: the kernel writes every texel the copy
 * then reads.  The call waits for the texture's stream (twice): not for use under stream capture. */
#define RZ_SYNC_SCRATCH_TEXELS (1024u * 1024u)      /* 16 MiB: a 1024x1024 image, twice the default shadow map's side */
static __device__ __align__(256) float4 rz_sync_scratch[RZ_SYNC_SCRATCH_TEXELS];

static const int RZ_MAX_DEVICES = 64;
static float4 *g_sync_scratch_ptr[RZ_MAX_DEVICES] = {nullptr};

/* the array's address on `device`, which must be the current device; looked up on first use (mi355r_init) */
static float4 *scratch_on_device(int device) {
    if (device < 0 || device >= RZ_MAX_DEVICES) fatal("device %d: the library keeps per-device state for devices 0..%d", device, RZ_MAX_DEVICES - 1);
    if (!g_sync_scratch_ptr[device]) {
        int current = -1;
        HIP_OK(hipGetDevice(&current));
        if (current != device) fatal("the staging array of device %d is looked up while device %d is current", device, current);
        void *ptr = nullptr;
        HIP_OK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(rz_sync_scratch)));
        if (!ptr) fatal("hipGetSymbolAddress gave no address for the staging array on device %d", device);
        g_sync_scratch_ptr[device] = (float4*)ptr;
    }
    return g_sync_scratch_ptr[device];
}

extern "C" int texture_sync_noalloc(rz_texture *p) {
    Tex *t = as_tex(p);
    if (!t->host_stale) return RZ_NOALLOC_OK;       /* the host image is current */
    const size_t texels = (size_t)t->pub.width * t->pub.height;
    if (texels > RZ_SYNC_SCRATCH_TEXELS) {
        fprintf(stderr, "mi355r: texture_sync_noalloc: %dx%d = %zu texels, the staging area holds %u (use texture_sync)\n",
                t->pub.width, t->pub.height, texels, RZ_SYNC_SCRATCH_TEXELS);
        return RZ_NOALLOC_TOO_LARGE;
    }
    if (t->pending_write) {
        /* the copy that fills it is recorded but not submitted: submitting sizes scratch, which allocates */
        fprintf(stderr, "mi355r: texture_sync_noalloc: the copy into this texture is still recorded; call graphics_flush first\n");
        return RZ_NOALLOC_NOT_SUBMITTED;
    }
    if (!t->d_data) fatal("texture_sync_noalloc: the texture has no device image");
    if (!t->pub.buffer) {
        /* an image that came in packed (texture_create_packed) and whose floats nobody has asked for yet: texture_sync makes the host image */
        fprintf(stderr, "mi355r: texture_sync_noalloc: the texture has no host image yet; call texture_sync once\n");
        return RZ_NOALLOC_NO_HOST_IMAGE;
    }
    if (!t->host_pinned) {
        /* the copy to a pageable host image goes through staging memory of the runtime's own */
        fprintf(stderr, "mi355r: texture_sync_noalloc: the texture's host image is not page-locked; call texture_sync\n");
        return RZ_NOALLOC_NOT_PINNED;
    }
    float4 *staging = scratch_on_device(g.device);
    hipStream_t st = g.lanes[t->write_lane < 0 ? 0 : t->write_lane].stream;     /* behind the copy that wrote the image */
    /* the passes of the flush in flight that fed the copy: if one of them ran out of scratch, the
     * image is not there yet and settling the flush may allocate */
    HIP_OK(hipStreamSynchronize(st));
    for (const FlushRecord &fr : g_in_flight)
        for (const FlushRecord::Copy &c : fr.copies)
            if (c.tex == t && fr.control(c.producer)[2] != 0) {
                fprintf(stderr, "mi355r: texture_sync_noalloc: the pass that renders this texture's source has to run again; call texture_sync\n");
                return RZ_NOALLOC_NOT_SUBMITTED;
            }
    const int w = t->pub.width, h = t->pub.height;
    rz_texture_untile_kernel<<<dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, st>>>(t->d_data, staging, w, h, t->flags);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(t->pub.buffer, staging, texels * sizeof(float4), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    t->host_stale = false;
    return RZ_NOALLOC_OK;
}

/* the last pass into `fb` of the flush in flight, if any */
static bool last_pass_in_flight(Fb *fb, FlushRecord **record, size_t *plan) {
    for (size_t r = g_in_flight.size(); r-- > 0;)
        for (size_t i = g_in_flight[r].plans.size(); i-- > 0;)
            if (g_in_flight[r].plans[i].fb == fb) { *record = &g_in_flight[r]; *plan = i; return true; }
    return false;
}

/* the framebuffer is about to be written outside a pass (a clear on its own): a pass of the flush in
 * flight that has to run again must do so first */
static void settle_before_touching(Fb *fb) {
    FlushRecord *fr;
    size_t plan;
    if (last_pass_in_flight(fb, &fr, &plan)) settle_in_flight(false);
}

/* texture_from_{color,depth}buffer at once (nothing recorded for the framebuffer): on its lane, behind
 * the passes already submitted.  When the last of those belongs to the flush in flight the copy joins
 * that flush's record: it does nothing if the pass did not run, and is repeated after the pass has
 * run again. */
static void launch_texture_write(Tex *t, Fb *fb, bool depth) {
    const int w = t->pub.width, h = t->pub.height;
    hipStream_t st = begin_texture_write(t, fb);
    FlushRecord *fr;
    size_t plan;
    const uint32_t *producer = nullptr;
    if (last_pass_in_flight(fb, &fr, &plan)) {
        producer = fr->control_dev(plan);
        fr->copies.push_back({t, fb, depth, plan});
    }
    CopyJobs jobs;
    memset(&jobs, 0, sizeof jobs);
    jobs.job[0] = copy_job(t, fb, depth, producer);
    rz_buffers_to_textures_kernel<<<dim3((w + 15) / 16, (h + 15) / 16, 1), dim3(256), 0, st>>>(jobs);
    HIP_OK(hipGetLastError());
    end_texture_write(t, fb);
}

/* MI355R_HOST_PROFILE=1: where the host's time in graphics_flush goes, printed when the process ends */
struct FlushProfile {
    bool on = getenv("MI355R_HOST_PROFILE") != nullptr;
    double seconds[8] = {0};
    unsigned long long flushes = 0, passes = 0;
    std::chrono::steady_clock::time_point last;
    void start() { if (on) last = std::chrono::steady_clock::now(); }
    void mark(int slot) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        seconds[slot] += std::chrono::duration<double>(now - last).count();
        last = now;
    }
    ~FlushProfile() {
        if (!on || !flushes) return;
        static const char *names[8] = {"plan", "batches", "tables", "scratch+params", "upload", "wait for the flush before", "launches", "copies"};
        fprintf(stderr, "mi355r: %llu flushes, %llu passes; host microseconds per pass:", flushes, passes);
        for (int i = 0; i < 8; i++) fprintf(stderr, " %s %.2f", names[i], seconds[i] * 1e6 / (double)passes);
        fprintf(stderr, "\n");
    }
};
static FlushProfile g_flush_profile;

/* One upload region per flush (FlushRecord) -> one host-to-device copy in front of the kernels and
 * one small copy of its control blocks back behind every batch, whatever the number of passes.
 *
 * Passes are grouped into batches (one launch sequence each) level by level: a
 * pass comes one level after the last pass into its framebuffer and after the
 * pass a recorded framebuffer copy follows when it samples the copied texture;
 * a batch holds passes of one level and one kind of rasteriser.  A batch runs
 * on the lane of its first framebuffer, or the next lane in turn; framebuffers
 * and GPU-written textures that come from other lanes are ordered by events.
 *
 * The flush before this one is settled first (its control blocks looked at, passes that ran out of
 * scratch run again): what this flush renders on top of is then what it should be.  The host has
 * recorded this flush while the GPU ran that one, so the wait is short. */
extern "C" void graphics_flush(void) {
    if (!g.ready || g.pending.empty()) return;
    FlushProfile &prof = g_flush_profile;
    prof.start();
    if (g.events.size() > 4096) sync_stream();          /* bound the number of live events; no event of this flush exists yet */
    FlushRecord fr;
    fr.epoch = g.flush_epoch;
    std::vector<PassPlan> &plans = fr.plans;
    size_t n = g.pending.size(), first = 0;
    while (first < n) {
        size_t last = first + 1;
        /* A pass ends where the framebuffer changes -- and in front of its first blended draw:
         * the draws before it need no colour below them and go through the split rasteriser and
         * the per-family shade kernels, only the rest needs ordered in-kernel shading.  It also ends
         * at the limits of what a triangle's record can say about it (16 bits of draw index). */
        const bool starts_blended = (g.pending[first].rec.flags & RZ_DRAW_BLEND) != 0;
        auto copy_point = [&](size_t at) {      /* a recorded copy of this framebuffer sits in front of draw `at` */
            for (auto &w : g.tex_writes) if (w.after == at && w.fb == g.pending[first].fb) return true;
            return false;
        };
        size_t tris = (size_t)g.pending[first].rec.num_faces;
        while (last < n && g.pending[last].fb == g.pending[first].fb && !copy_point(last) &&
               (starts_blended || !(g.pending[last].rec.flags & RZ_DRAW_BLEND)) &&
               last - first < 65535 && tris + (size_t)g.pending[last].rec.num_faces <= 0x0fffffffu) {
            tris += (size_t)g.pending[last].rec.num_faces;
            last++;
        }
        if (tris > 0x0fffffffu) fatal("%zu triangles in one draw (limit 268435455)", tris);
        PassPlan pl;
        pl.first = first;
        pl.last = last;
        plans.push_back(std::move(pl));
        first = last;
    }
    size_t total = CONTROL_BYTES * plans.size();
    total = (total + 255) & ~(size_t)255;
    fr.flag_at = total;
    total += 256;
    fr.params_at = total;
    total += sizeof(PassParams) * plans.size();
    fr.again_at = total;
    total += sizeof(PassParams) * plans.size();
    for (auto &pl : plans) plan_pass(pl, total);
    /* poses to evaluate on the device: one job per (source, time) of the flush, whatever the number of
     * draws, passes and models that share it */
    std::vector<PoseJob> pose_jobs;
    {
        std::unordered_map<unsigned long long, size_t> seen[1];
        std::vector<std::pair<rz_pose_source*, size_t>> by_source;      /* small: a scene has a skeleton or two */
        std::vector<size_t> job_joints;
        g_pose_offsets.clear();
        size_t pool_bytes = 0;
        for (auto &pd : g.pending) {
            if (!pd.pose) continue;
            uint32_t bits;
            memcpy(&bits, &pd.pose_time, sizeof bits);
            size_t source_index = 0;
            while (source_index < by_source.size() && by_source[source_index].first != pd.pose) source_index++;
            if (source_index == by_source.size()) by_source.push_back({pd.pose, source_index});
            const unsigned long long key = ((unsigned long long)source_index << 32) | bits;
            auto found = seen[0].find(key);
            if (found == seen[0].end()) {
                found = seen[0].emplace(key, pose_jobs.size()).first;
                PoseJob job;
                memset(&job, 0, sizeof job);
                job.skeleton = pd.pose->d_block;
                job.time = pd.pose_time;
                pose_jobs.push_back(job);
                job_joints.push_back((size_t)pd.pose->num_joints);
                g_pose_offsets.push_back(pool_bytes);
                pool_bytes += ((sizeof(rz_mat4) + sizeof(rz_mat3)) * (size_t)pd.pose->num_joints + 255) & ~(size_t)255;
            }
            pd.pose_job = found->second;
            if (pd.pose_attached >= 0) fr.attach_jobs++;
        }
        fr.pose_jobs = pose_jobs.size();
        total = (total + 15) & ~(size_t)15;
        fr.pose_jobs_at = total;
        total += sizeof(PoseJob) * fr.pose_jobs;
        fr.attach_jobs_at = total;
        total += sizeof(AttachJob) * fr.attach_jobs;
        /* this flush's pool: the one the flush before the flush in flight used, long finished */
        DevBuf<char> &pool = g.pose_pool[g.flush_epoch & 1];
        if (pool_bytes > pool.cap) {
            if (pool.ptr) HIP_OK(hipFree(pool.ptr));
            pool.cap = std::max(pool_bytes, pool.cap * 2);
            HIP_OK(hipMalloc((void**)&pool.ptr, pool.cap));
        }
        g_pose_pool = pool.ptr;
        for (size_t k = 0; k < pose_jobs.size(); k++) {
            pose_jobs[k].joints = (float*)(pool.ptr + g_pose_offsets[k]);
            pose_jobs[k].normals = (float*)(pool.ptr + g_pose_offsets[k] + sizeof(rz_mat4) * job_joints[k]);
        }
    }
    prof.mark(0);

    /* batches and lanes */
    const int lanes = g.timing ? 1 : g.num_lanes;      /* per-kernel timing wants kernels alone on the GPU */
    /* Batches.  Inside one flush a pass depends only on the earlier passes into its own
     * framebuffer (textures written on the GPU were written before this flush started), so passes
     * are taken level by level: level = number of earlier passes into the same framebuffer.  A
     * batch holds passes of one level and one kind of rasteriser, in submission order. */
    std::vector<Batch> &batches = fr.batches;
    std::unordered_map<size_t, std::vector<size_t>> copies_after;   /* pass -> the recorded copies (fr.copies) it feeds */
    {
        /* a recorded copy runs behind the last pass into its framebuffer that ends at or before it;
         * a pass that samples the copied texture comes one level after that pass */
        std::unordered_map<Tex*, size_t> producer;
        for (auto &w : g.tex_writes) {
            size_t found = plans.size();
            for (size_t i = 0; i < plans.size(); i++)
                if (plans[i].fb == w.fb && plans[i].last <= w.after) found = i;
            if (found == plans.size()) fatal("a recorded framebuffer copy has no pass to follow");
            producer[w.tex] = found;
            copies_after[found].push_back(fr.copies.size());
            fr.copies.push_back({w.tex, w.fb, w.depth, found});
        }
        std::unordered_map<Fb*, int> last_level;
        int levels = 0;
        for (size_t i = 0; i < plans.size(); i++) {
            auto seen = last_level.find(plans[i].fb);
            int l = seen == last_level.end() ? 0 : seen->second + 1;
            for (size_t d = plans[i].first; d < plans[i].last; d++)
                for (Tex *t : g.pending[d].pending_textures) {
                    auto from = producer.find(t);
                    if (from == producer.end() || from->second >= i) fatal("a draw samples a texture whose copy is not recorded before it");
                    l = std::max(l, plans[from->second].level + 1);
                }
            plans[i].level = l;
            last_level[plans[i].fb] = l;
            levels = std::max(levels, l + 1);
        }
        size_t params_used = 0;
        for (int l = 0; l < levels; l++) {
            for (int blend = 0; blend < 2; blend++) {
                Batch b;
                auto close = [&]() {
                    if (b.members.empty()) return;
                    b.off_params = fr.params_at + sizeof(PassParams) * params_used;
                    for (size_t i : b.members) plans[i].slot = params_used++;
                    int lane = plans[b.members[0]].fb->lane;
                    if (lane < 0 || lane >= lanes) { lane = g.next_lane % lanes; g.next_lane = (g.next_lane + 1) % lanes; }
                    b.lane = lane;
                    batches.push_back(std::move(b));
                    b = Batch();
                };
                for (size_t i = 0; i < plans.size(); i++) {
                    if (plans[i].level != l || (int)plans[i].has_blend != blend) continue;
                    if (b.members.empty()) { b.has_blend = blend != 0; b.families = 0; b.sky_table = false; }
                    b.members.push_back(i);
                    b.families |= plans[i].families;
                    b.sky_table = b.sky_table || plans[i].sky_draws > 0;
                    if (b.members.size() >= g.max_batch) close();
                }
                close();
            }
        }
    }
    prof.mark(1);

    fr.at = ring_alloc(total);
    char *hb = g.ring_host + fr.at, *db = g.ring_dev + fr.at;
    memset(hb + fr.flag_at, 0, 256);
    g_attach_jobs_at = fr.attach_jobs_at;
    g_attach_jobs_filled = 0;
    for (auto &pl : plans) fill_pass(pl, hb, db);
    if (fr.pose_jobs) memcpy(hb + fr.pose_jobs_at, pose_jobs.data(), sizeof(PoseJob) * fr.pose_jobs);
    prof.mark(2);
    /* scratch first, for every batch: growing a buffer moves it, and the tables written next hold its address */
    for (auto &b : batches) prepare_batch(b, fr, true, PREPARE_FIRST);
    for (auto &b : batches) prepare_batch(b, fr, false, PREPARE_FIRST);
    prof.mark(3);
    HIP_OK(hipMemcpyAsync(db + fr.flag_at, hb + fr.flag_at, total - fr.flag_at, hipMemcpyHostToDevice, g.copy_stream));
    prof.mark(4);
    /* Now the flush before this one: everything above went on while the GPU was still busy with it, and
     * nothing has been launched yet.  If some of its passes had to run again the lanes' scratch may
     * have grown and moved, and the estimates have been raised: this flush's parameters are written
     * and sent once more. */
    if (settle_in_flight(false)) {
        HIP_OK(hipStreamSynchronize(g.copy_stream));
        for (auto &b : batches) prepare_batch(b, fr, true, PREPARE_REWRITE);
        for (auto &b : batches) prepare_batch(b, fr, false, PREPARE_REWRITE);
        HIP_OK(hipMemcpyAsync(db + fr.flag_at, hb + fr.flag_at, total - fr.flag_at, hipMemcpyHostToDevice, g.copy_stream));
    }
    /* the flush's poses, and the model matrices of draws attached to a joint, behind the upload */
    if (fr.pose_jobs) {
        rz_pose_kernel<<<dim3((unsigned)fr.pose_jobs), dim3(64), 0, g.copy_stream>>>((const PoseJob*)(db + fr.pose_jobs_at));
        if (fr.attach_jobs)
            rz_attach_kernel<<<dim3((unsigned)((fr.attach_jobs + 63) / 64)), dim3(64), 0, g.copy_stream>>>((const AttachJob*)(db + fr.attach_jobs_at), (int)fr.attach_jobs);
        HIP_OK(hipGetLastError());
    }
    hipEvent_t uploaded = new_event(g.copy_stream);
    prof.mark(5);

    bool *used = fr.used;
    for (auto &b : batches) {
        Lane &L = g.lanes[b.lane];
        if (!used[b.lane]) { HIP_OK(hipStreamWaitEvent(L.stream, uploaded, 0)); used[b.lane] = true; }
        std::vector<Tex*> sampled;
        bool awaited[MAX_LANES] = {false};
        for (size_t i : b.members) {
            const PassPlan &pl = plans[i];
            if (pl.fb->lane >= 0 && pl.fb->lane != b.lane && !awaited[pl.fb->lane]) {   /* last touched on another lane */
                HIP_OK(hipStreamWaitEvent(L.stream, new_event(g.lanes[pl.fb->lane].stream), 0));
                awaited[pl.fb->lane] = true;        /* everything submitted to that lane so far: once is enough */
            }
            pl.fb->lane = b.lane;
            for (size_t d = pl.first; d < pl.last; d++) {
                for (Tex *t : g.pending[d].gpu_textures)
                    if (std::find(sampled.begin(), sampled.end(), t) == sampled.end()) sampled.push_back(t);
                for (Tex *t : g.pending[d].pending_textures)    /* its copy followed a batch of a lower level: launched by now */
                    if (std::find(sampled.begin(), sampled.end(), t) == sampled.end()) sampled.push_back(t);
            }
        }
        {
            /* per lane that wrote some of them, the latest write: a stream finishes its work in order */
            Tex *latest[MAX_LANES] = {nullptr};
            for (Tex *t : sampled)
                if (t->write_event && t->write_lane != b.lane && (!latest[t->write_lane] || t->write_seq > latest[t->write_lane]->write_seq))
                    latest[t->write_lane] = t;
            for (int l = 0; l < MAX_LANES; l++)
                if (latest[l]) HIP_OK(hipStreamWaitEvent(L.stream, latest[l]->write_event, 0));
        }
        launch_batch(b, plans, db);
        if (!sampled.empty()) {
            hipEvent_t done = new_event(L.stream);
            for (Tex *t : sampled) t->read_events.push_back(done);
        }
        prof.mark(6);
        {
            /* the recorded copies this batch feeds, on its lane, RZ_COPY_GROUP per launch */
            std::vector<size_t> feeds;
            for (size_t i : b.members) {
                auto copies = copies_after.find(i);
                if (copies != copies_after.end()) feeds.insert(feeds.end(), copies->second.begin(), copies->second.end());
            }
            for (size_t at = 0; at < feeds.size(); at += RZ_COPY_GROUP) {
                const size_t count = std::min<size_t>(RZ_COPY_GROUP, feeds.size() - at);
                CopyJobs jobs;
                memset(&jobs, 0, sizeof jobs);
                std::vector<hipEvent_t> readers;
                int max_w = 0, max_h = 0;
                for (size_t k = 0; k < count; k++) {
                    const FlushRecord::Copy &w = fr.copies[feeds[at + k]];
                    Tex *t = w.tex;
                    if (w.fb->lane != b.lane) fatal("a recorded copy is not on its framebuffer's lane");
                    for (hipEvent_t e : t->read_events) if (std::find(readers.begin(), readers.end(), e) == readers.end()) readers.push_back(e);
                    t->read_events.clear();
                    if (t->write_event && t->write_lane != b.lane) readers.push_back(t->write_event);
                    jobs.job[k] = copy_job(t, w.fb, w.depth, fr.control_dev(w.producer));
                    max_w = std::max(max_w, t->pub.width); max_h = std::max(max_h, t->pub.height);
                }
                for (hipEvent_t e : readers) HIP_OK(hipStreamWaitEvent(L.stream, e, 0));   /* passes that still sample the old contents */
                rz_buffers_to_textures_kernel<<<dim3((max_w + 15) / 16, (max_h + 15) / 16, (unsigned)count), dim3(256), 0, L.stream>>>(jobs);
                HIP_OK(hipGetLastError());
                hipEvent_t written = new_event(L.stream);
                for (size_t k = 0; k < count; k++) {
                    const FlushRecord::Copy &w = fr.copies[feeds[at + k]];
                    end_texture_write(w.tex, w.fb, written);
                    w.tex->pending_write = false;
                }
            }
        }
        /* the batch's control blocks come back behind it, on its lane */
        HIP_OK(hipMemcpyAsync(hb + CONTROL_BYTES * plans[b.members[0]].slot, db + CONTROL_BYTES * plans[b.members[0]].slot,
                              CONTROL_BYTES * b.members.size(), hipMemcpyDeviceToHost, L.stream));
        prof.mark(7);
    }
    prof.flushes++;
    prof.passes += plans.size();
    for (Pending &pd : g.pending) if (pd.transient) remember_stream(pd);
    g.pending.clear();
    g.side.clear();
    g.transient_verts.clear();
    g.tex_writes.clear();
    g.flush_epoch++;
    g_in_flight.push_back(std::move(fr));
    upload_remembered_streams();
    prof.mark(7);
}

/* --------------------------------------------------------------- inspection -- */

extern "C" int mi355r_inspect_pass(rz_pass_info *info) {
    if (!g.ready || !g.have_last_pass) return 0;
    graphics_flush();
    sync_stream();
    const PassParams &P = g.last_pass;
    uint32_t control[8];
    HIP_OK(hipMemcpy(control, P.control, sizeof control, hipMemcpyDeviceToHost));
    info->num_tris = P.num_tris;
    info->width = P.width;
    info->height = P.height;
    info->bin_edge = RZ_BIN;
    info->bins_x = P.bins_x;
    info->bins_y = P.bins_y;
    info->bin_entries = control[1];
    return 1;
}

extern "C" void mi355r_inspect_boxes(unsigned int *boxes, unsigned int *fans) {
    if (!g.have_last_pass) fatal("mi355r_inspect_boxes: no pass to look at");
    const PassParams &P = g.last_pass;
    const size_t T = (size_t)P.num_tris;
    std::vector<TriRec> recs(T);
    HIP_OK(hipMemcpy(boxes, P.bbox, T * sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(recs.data(), P.tri, T * sizeof(TriRec), hipMemcpyDeviceToHost));
    /* triangles of sky boxes that are kept on the pass's table have no box in the binning kernels' array: theirs
     * are put together from the table and their records */
    if (P.sky_draws > 0) {
        std::vector<uint32_t> table(RZ_SKY_TABLE), first_tri((size_t)P.num_draws + 1);
        HIP_OK(hipMemcpy(table.data(), P.sky_table, table.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(first_tri.data(), P.first_tri, first_tri.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int k = 0; k < P.sky_draws; k++) {
            const int d = P.num_draws - P.sky_draws + k;
            for (uint32_t t = first_tri[d]; t < first_tri[d + 1]; t++) {
                const uint32_t *entry = &table[((size_t)k * RZ_SKY_FACES + (t - first_tri[d])) * RZ_MAX_FAN];
                bool any = false;
                for (int j = 0; j < RZ_MAX_FAN; j++) any = any || entry[j] != 0xffffffffu;
                if (!any) continue;
                const uint32_t fan = RZ_META_CLIPPED(recs[t].meta) ? RZ_META_FAN(recs[t].meta) : 1u;
                boxes[2 * t] = recs[t].bx;
                boxes[2 * t + 1] = (recs[t].by & 0x3fffu) | ((recs[t].by >> 16) << 14) | (fan << 28);
            }
        }
    }
    for (size_t t = 0; t < T; t++) {
        /* device packing (pack_bin_box): y0 | y1 << 14 | entries << 28 */
        const bool empty = (boxes[2 * t] & 0xffffu) > (boxes[2 * t] >> 16);
        const uint32_t y = boxes[2 * t + 1];
        fans[t] = empty ? 0u : y >> 28;
        boxes[2 * t + 1] = empty ? RZ_BBOX_EMPTY_X : ((y & 0x3fffu) | (((y >> 14) & 0x3fffu) << 16));
        if (!empty && fans[t] != (RZ_META_CLIPPED(recs[t].meta) ? RZ_META_FAN(recs[t].meta) : 1u))
            fatal("triangle %zu: packed entry count %u disagrees with its record", t, fans[t]);
    }
}

extern "C" void mi355r_inspect_bins(unsigned int *counts, unsigned int *entries) {
    if (!g.have_last_pass) fatal("mi355r_inspect_bins: no pass to look at");
    const PassParams &P = g.last_pass;
    const size_t T = (size_t)P.num_tris, bins = (size_t)P.bins_x * P.bins_y;
    std::vector<TriRec> recs(T);
    std::vector<uint2> boxes(T), ranges(bins);
    HIP_OK(hipMemcpy(recs.data(), P.tri, T * sizeof(TriRec), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(boxes.data(), P.bbox, T * sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ranges.data(), P.bin_range, bins * sizeof(uint2), hipMemcpyDeviceToHost));
    /* fan block -> input triangle (a clipped triangle's record carries its block in ax) */
    std::unordered_map<uint32_t, uint32_t> owner;
    for (size_t t = 0; t < T; t++) {
        const bool empty = (boxes[t].x & 0xffffu) > (boxes[t].x >> 16);
        if (!empty && RZ_META_CLIPPED(recs[t].meta)) {
            uint32_t block;
            memcpy(&block, &recs[t].ax, sizeof block);
            owner[block] = (uint32_t)t;
        }
    }
    size_t out = 0;
    std::vector<BinEntry> list;
    for (size_t b = 0; b < bins; b++) {
        counts[b] = ranges[b].y;
        list.resize(ranges[b].y);
        if (ranges[b].y)
            HIP_OK(hipMemcpy(list.data(), P.bin_list + ranges[b].x, ranges[b].y * sizeof(BinEntry), hipMemcpyDeviceToHost));
        for (const BinEntry &e : list) {
            if (e.tri & 0x80000000u) {
                const uint32_t idx = e.tri & 0x7fffffffu;
                auto it = owner.find(idx / RZ_MAX_FAN);
                if (it == owner.end()) fatal("bin entry refers to fan block %u that no triangle owns", idx / RZ_MAX_FAN);
                entries[out++] = it->second;
                entries[out++] = idx % RZ_MAX_FAN;
            } else {
                entries[out++] = e.tri;
                entries[out++] = 0;
            }
        }
    }
}

/* -------------------------------------------------------------- measurement -- */

extern "C" void mi355r_enable_counters(int on) {
    ensure_ready();
    graphics_flush();
    g.counting = on != 0;
}

extern "C" void mi355r_reset_counters(void) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    HIP_OK(hipMemset(g.d_counters, 0, 12 * sizeof(unsigned long long)));
    HIP_OK(hipDeviceSynchronize());
    memset(g.host_counters, 0, sizeof g.host_counters);
}

extern "C" void mi355r_get_counters(unsigned long long out[12]) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    unsigned long long dev[12];
    HIP_OK(hipMemcpy(dev, g.d_counters, sizeof dev, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = g.host_counters[i];
    for (int i = 8; i < 12; i++) out[i] = 0;
    out[1] = dev[1];
    out[2] = dev[2];
    out[6] = dev[6];
    out[7] = dev[7];
    out[8] = dev[8];
    out[9] = dev[9];
}

/* Per-kernel device time, measured with HIP events on the pipeline's own
 * stream around every launch while enabled.  out[0..6]: summed milliseconds of
 * the vertex, setup+clip, bin, raster, skybox-shade, blinn-shade and pbr-shade
 * kernels (the shade slots stay zero for passes whose shading runs inside the
 * raster kernel); out[7]: launch sequences timed (each covers a batch of
 * passes); returns the number of passes. */
extern "C" void mi355r_enable_timing(int on) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    g.timing = on != 0;
}

extern "C" unsigned long long mi355r_get_timing(double out[8], int reset) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    for (size_t i = 0, b = 0; i + 7 < g.timing_events.size(); i += 8, b++) {
        for (int k = 0; k < 7; k++) {
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, g.timing_events[i + k], g.timing_events[i + k + 1]));
            g.kernel_ms[k] += ms;
        }
        g.kernel_launches += b < g.timing_passes.size() ? g.timing_passes[b] : 1;
        g.timed_sequences++;
    }
    g.timing_passes.clear();
    for (hipEvent_t e : g.timing_events) HIP_OK(hipEventDestroy(e));
    g.timing_events.clear();
    for (int k = 0; k < 7; k++) out[k] = g.kernel_ms[k];
    out[7] = (double)g.timed_sequences;
    unsigned long long n = g.kernel_launches;
    if (reset) { for (int k = 0; k < 7; k++) g.kernel_ms[k] = 0; g.kernel_launches = 0; g.timed_sequences = 0; }
    return n;
}

/* tests: force the first-time sizes of the clipper's block table and the bin lists (0 = the
 * library's own estimate), and count the passes that had to run again */
extern "C" void mi355r_set_scratch_estimate(unsigned long long fan_blocks, unsigned long long list_entries) {
    ensure_ready();
    graphics_flush();
    g.forced_fan_estimate = (size_t)fan_blocks;
    g.forced_list_estimate = (size_t)list_entries;
    if (!fan_blocks && !list_entries) g.fan_highwater = g.list_highwater = 0;
}

extern "C" unsigned long long mi355r_replayed_draws(void) { return g.replayed_draws; }

extern "C" const char *mi355r_last_raster_kernel(void) { return g.last_raster_kernel; }
#ifdef RZ_PROBE
extern "C" int mi355r_debug_probe(unsigned long long *out, int max_records) {
    HIP_OK(hipDeviceSynchronize());
    unsigned int n = 0;
    HIP_OK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(rz_probe_count), sizeof n));
    n = std::min(n, (unsigned)RZ_PROBE_MAX);
    n = std::min(n, (unsigned)max_records);
    if (out && n) HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(rz_probe_data), (size_t)n * RZ_PROBE_WORDS * sizeof(unsigned long long)));
    unsigned int zero = 0;
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(rz_probe_count), &zero, sizeof zero));
    return (int)n;
}
#endif

extern "C" int mi355r_inspect_sky_first(void) { return g.have_last_pass ? g.last_pass_sky_first : 0; }

extern "C" unsigned long long mi355r_passes_run_again(void) {
    ensure_ready();
    graphics_flush();
    sync_stream();
    return g.passes_run_again;
}

extern "C" void mi355r_shutdown(void) {
    if (!g.ready) return;
    graphics_flush();
    sync_stream();
}
