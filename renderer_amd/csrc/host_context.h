/* Host side, part 1 of 6 (included by api.hip, one translation unit: the kernels' device-side tables have one copy):
 * the objects behind the C ABI's handles, the library's context, execution lanes, start-up. */

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

/* the maps of a material side by side (DrawRec::bundle) */
struct Bundle {
    uint32_t *d = nullptr;
    size_t texels = 0;
    int words = 0;
    unsigned long long version[6] = {};     /* Tex::image_version of the maps it was made from */
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
    /* the sky kernel of a batch runs beside the batch's shade kernels: it keeps the vector units busy, they wait for
     * memory, and neither reads what the other writes (fork after the strips, join before whatever comes next) */
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    DevBuf<float> clip, vary, fan_vary;
    DevBuf<TriRec> tri, fan_tri;
    DevBuf<uint2> bbox;
    DevBuf<uint32_t> clip_list, ref;
    DevBuf<uint2> bin_range;
    DevBuf<SkyBin> sky_bins;            /* per bin of the batch: bin_class_of's record */
    DevBuf<uint32_t> bin_class;         /* per bin of the batch: what rz_strip_kernel has to do with it */
    DevBuf<uint32_t> bin_counts, queue, queue_counts, tile_listed;
    DevBuf<BinEntry> bin_list;
    DevBuf<uint2> bin_boxes;
};

static const int RZ_TEX_NARROW_FORMATS = RZ_TEX_R32F | RZ_TEX_UNORM8 | RZ_TEX_SRGB8 | RZ_TEX_RGBE8;
static const size_t MAX_BATCH = 256;    /* passes per launch sequence, upper bound */
static const int MAX_LANES = 8;

struct Context {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;       /* lane 0's stream: uploads and utilities */
    hipStream_t copy_stream = nullptr;  /* table uploads / control read-back of a flush */
    Lane lanes[MAX_LANES];
    int num_lanes = 1, next_lane = 0;
    size_t max_batch = 64;                  /* passes per launch sequence (MI355R_BATCH) */
    bool sky_bytes = true;                  /* MI355R_SKY_BYTES=0: skybox fragments fetch the face's texel and convert it themselves */
    bool pack_textures = true;              /* MI355R_PACK=0: keep every image as float4 */
    int micro_min_list = 256;               /* MI355R_MICRO_MIN_LIST: small triangles are rasterised one per lane only in bins with at least this many list entries */
    int crowd_min = 8192;                   /* MI355R_CROWD_MIN: bins with at least this many list entries have their blocks rasterised by several waves each, beside the strips (0: none) */
    bool crowd_always = false;              /* (set with MI355R_CROWD_MIN: also for framebuffers whose last frame is not known to have had such a bin) */
    int crowd_split = 4;                    /* MI355R_CROWD_SPLIT: that many (2, 4, 8, 16) */
    int fast_start = 1;                     /* MI355R_FAST_START=0: every batch starts with rz_begin_kernel; 2: none that need not (default: those of up to four passes) */
    bool sky_side = true;                   /* MI355R_SKY_SIDE=0: the sky kernel runs in line, on its batch's own stream */
    bool strip_team = true;                 /* MI355R_STRIP_TEAM=0: a launch of at most two passes takes the plain strips, not two waves per block (rz_strip_team_kernel) */
    bool sky_with_strips = true;            /* MI355R_SKY_WITH_STRIPS=0: ... and that launch leaves the sky of empty bins to rz_sky_kernel */
    bool sky_table = true;                  /* MI355R_SKY_TABLE=0: a sky box goes through the bins like any other draw */
    int fill_waves = 0;                     /* MI355R_FILL_WAVES: waves that share a chunk of 64 triangles in rz_bin_fill_kernel (1, 2, 4, 8; unset: by the launch's size) */
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
    size_t ring_size = 0, ring_turn = 0;    /* the slot the next flush takes (ring_alloc) */
    bool ring_whole = false;                /* the last flush took the whole ring */
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
    std::map<std::array<const void*, 6>, Bundle> bundles;           /* a material's textures (in map order, nullptr = none) -> their bundle */
    bool bundle_maps = true;                /* MI355R_BUNDLE=0: every map is sampled from its own image */
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
        if (env) { const int v = atoi(env); if (v == 1 || v == 2 || v == 4 || v == 8) g.fill_waves = v; }
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
        env = getenv("MI355R_BUNDLE");
        if (env) g.bundle_maps = atoi(env) != 0;
        env = getenv("MI355R_PACK");
        if (env) g.pack_textures = atoi(env) != 0;
        env = getenv("MI355R_BATCH");
        if (env && atoi(env) >= 1) g.max_batch = std::min<size_t>((size_t)atoi(env), MAX_BATCH);
    }
    for (int i = 0; i < g.num_lanes; i++) HIP_OK(hipStreamCreateWithFlags(&g.lanes[i].stream, hipStreamNonBlocking));
    if (const char *env = getenv("MI355R_SKY_SIDE")) g.sky_side = atoi(env) != 0;
    if (const char *env = getenv("MI355R_FAST_START")) g.fast_start = std::max(0, std::min(atoi(env), 2));
    if (const char *env = getenv("MI355R_STRIP_TEAM")) g.strip_team = atoi(env) != 0;
    if (const char *env = getenv("MI355R_SKY_WITH_STRIPS")) g.sky_with_strips = atoi(env) != 0;
    if (const char *env = getenv("MI355R_CROWD_MIN")) { g.crowd_min = std::max(0, atoi(env)); g.crowd_always = true; }
    if (const char *env = getenv("MI355R_CROWD_SPLIT")) { const int v = atoi(env); if (v == 2 || v == 4 || v == 8 || v == 16) g.crowd_split = v; }
    for (int i = 0; i < g.num_lanes && (g.sky_side || g.crowd_min > 0); i++) {
        HIP_OK(hipStreamCreateWithFlags(&g.lanes[i].side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&g.lanes[i].fork, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&g.lanes[i].join, hipEventDisableTiming));
    }
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
