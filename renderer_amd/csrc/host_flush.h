/* Host side, part 4 of 6: the flush -- pending draws become passes, passes become batches of up to 64 on the execution
 * lanes, each batch one sequence of kernel launches; scratch that ran out, passes that run again, settling a flush. */

/* ------------------------------------------------------------------- flush -- */

static const size_t CONTROL_BYTES = 64;

/* A flush's upload region.  The ring is taken in RING_SLOTS equal slots, one per flush, in turn: when a flush is put
 * together only the flush before it can still be running (the one before that was settled, passes run again
 * included, before that flush launched anything), so a slot that comes round again after RING_SLOTS flushes is free
 * without asking the GPU -- and the processor writes the tables into memory it wrote a few flushes ago, which its
 * caches still hold, instead of fetching 64 MB of lines in turn from memory only to overwrite them.  A flush that
 * does not fit a slot (hundreds of thousands of triangles recorded one by one) takes the whole ring, grown if need
 * be, behind a wait for everything that used it; so does the flush after such a one. */
static const size_t RING_SLOTS = 4;

static size_t ring_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > g.ring_size) {
        sync_stream();
        size_t size = g.ring_size;
        while (size < bytes) size *= 2;
        HIP_OK(hipHostFree(g.ring_host));
        HIP_OK(hipFree(g.ring_dev));
        HIP_OK(hipHostMalloc((void**)&g.ring_host, size, hipHostMallocDefault));
        HIP_OK(hipMalloc((void**)&g.ring_dev, size));
        g.ring_size = size;
        g.ring_whole = true;        /* (nothing is in flight any more; the flag only sends the next flush through the wait below) */
    }
    const size_t slot_bytes = (g.ring_size / RING_SLOTS) & ~(size_t)255;
    if (bytes > slot_bytes || g.ring_whole) {
        sync_stream();              /* everything that used the ring has finished */
        g.ring_whole = bytes > slot_bytes;
        g.ring_turn = 0;
        if (g.ring_whole) return 0;
    }
    return (g.ring_turn++ % RING_SLOTS) * slot_bytes;
}

/* layout of one pass inside the flush's upload region */
struct PassPlan {
    size_t first, last;             /* range in g.pending (while the flush is being put together) */
    size_t T;                       /* triangles */
    size_t off_draws, off_first;    /* offsets from the region start */
    size_t off_sky_table;           /* ... of the pass's table of sky triangles (sky_draws > 0) */
    std::vector<size_t> side_at, verts_at;
    bool has_blend, samples_gpu_texture;
    bool narrow_texels;             /* every blinn / pbr draw samples 4-byte images only */
    int families;                   /* bit f: draws of family f that go through the bins */
    bool sky_bytes;                 /* every sky box on the pass's table has its cube's plane of display bytes */
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
    pl.sky_bytes = true;
    if (g.sky_table && !g.counting && !pl.has_blend)
        while (pl.sky_first > pl.first && pl.sky_draws < RZ_SKY_DRAWS) {
            const DrawRec &r = g.pending[pl.sky_first - 1].rec;
            if (r.family != RZ_FAMILY_SKYBOX || r.num_faces > RZ_SKY_FACES) break;
            pl.sky_first--;
            pl.sky_draws++;
            pl.sky_bytes = pl.sky_bytes && (r.flags & RZ_DRAW_SKY_BYTES) != 0;
        }
    /* (a shadow pass's fragments are black: rz_strip_kernel writes them itself, no family's kernel has work from them) */
    for (size_t i = pl.first; i < pl.sky_first; i++)
        if (pl.has_blend || !(g.pending[i].rec.flags & RZ_DRAW_SHADOW_PASS)) pl.families |= 1 << g.pending[i].rec.family;
    total = (total + 255) & ~(size_t)255;
    pl.off_sky_table = total;
    if (pl.sky_draws > 0) total += sizeof(uint32_t) * RZ_SKY_TABLE;
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
    bool sky_bytes = true;          /* ... and every such sky box has its cube's plane of display bytes */
    bool fast_start = false;        /* no pass of it waits for another pass of the flush, none is run again: no rz_begin_kernel */
    bool beside = false;            /* other batches of its level run beside it on other lanes (a flush of more passes than a batch holds) */
    bool crowd = false;             /* its crowded bins are rz_crowd_kernel's (settled when the batch is put together: its passes' parameters and its launches must agree) */
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
/* does a batch hand its crowded bins to rz_crowd_kernel?  (asked when its passes' parameters are written and again
 * when it is launched) */
/* does a batch hand its crowded bins to rz_crowd_kernel?  Only where a framebuffer's last frame had such a bin (control
 * word 4): a fork, a launch that finds an empty list and a join cost every other batch 2 %. */
template <class Plans>
static bool wants_crowd(const Batch &b, const Plans &plans) {
    static const bool paranoid = getenv("MI355R_SYNC") != nullptr;
    if (!(g.crowd_min > 0 && !g.counting && !g.timing && !paranoid && !b.has_blend && g.lanes[b.lane].side != nullptr)) return false;
    if (g.crowd_always) return true;
    uint32_t longest = 0;
    for (size_t i : b.members) longest = std::max(longest, plans[i].fb->longest_list);
    return longest >= (uint32_t)g.crowd_min;
}

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
    reserve(L.clip_list, 2 * tris, st);                 /* (triangle, draw) */
    reserve(L.fan_tri, fans * RZ_MAX_FAN, st);
    reserve(L.fan_vary, fans * RZ_MAX_POLY * RZ_VARY_STRIDE, st);
    reserve(L.bin_range, bins, st);
    reserve(L.sky_bins, bins, st);
    reserve(L.bin_class, bins, st);
    reserve(L.bin_counts, matrix, st);
    /* tiles are dealt to the parts of a list by their index: a part holds at most its even share */
    /* (32x8 tiles, or the two 16x8 blocks of each that rz_strip_kernel lists) */
    const size_t queue_capacity = (2 * tiles + RZ_QUEUE_WAYS - 1) / RZ_QUEUE_WAYS + b.members.size();    /* every pass's share rounded up */
    reserve(L.tile_listed, tiles, st);
    reserve(L.queue, RZ_LISTS * RZ_QUEUE_WAYS * queue_capacity, st);    /* blinn, pbr, RZ_ORDERED_LIST, RZ_VISIBILITY_LIST */
    reserve(L.queue_counts, (RZ_BATCH_CONTESTED + 1u) * RZ_QUEUE_PAD, st);
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
        P.clip_list = L.clip_list.ptr + 2 * tri_off;
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
        /* the table of sky triangles lives in the flush's upload region, where the host has marked every entry as
         * empty: the vertex kernel finds it ready (a batch run again: rz_begin_kernel marks it anew) */
        if (mode == PREPARE_FIRST) {
            P.sky_table = pl.sky_draws > 0 ? (uint32_t*)(db + pl.off_sky_table) : nullptr;
            if (pl.sky_draws > 0) memset(hb + pl.off_sky_table, 0xff, sizeof(uint32_t) * RZ_SKY_TABLE);
        }
        P.counters = g.counting ? g.d_counters : nullptr;
        P.scan_list = !b.has_blend ? -1 : splits_blend(tiles, b.members.size()) ? (int)RZ_VISIBILITY_LIST : (int)RZ_ORDERED_LIST;
        P.micro_area = g.micro_area;
        P.micro_min_list = g.micro_min_list;
        P.crowd_min = b.crowd ? g.crowd_min : 0;
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
        if (L.bin_range.cap < bins || L.bin_class.cap < bins || L.sky_bins.cap < bins) fatal("per-bin scratch of lane %d is smaller than the batch it is about to run", b.lane);
        const bool uses_ref = !(b.has_blend && !splits_blend(batch_tiles, n));
        if (L.queue_counts.cap < (RZ_BATCH_CONTESTED + 1u) * RZ_QUEUE_PAD || L.queue.cap < RZ_LISTS * RZ_QUEUE_WAYS * (size_t)queue_capacity ||
            L.tile_listed.cap < batch_tiles || L.bin_boxes.cap < L.bin_list.cap || (uses_ref && (!L.ref.ptr || L.ref.cap < pixels)) || L.tri.cap < tris || L.bbox.cap < tris)
            fatal("scratch of lane %d is smaller than the batch it is about to run", b.lane);
    }
    stamp();
    /* What the batch's kernels count in or add into starts at zero.  A batch whose passes depend on no other pass of
     * their flush has its control blocks and sky tables from the upload, ready, and the rest -- lane scratch that nothing
     * before the bin count looks at -- is zeroed by workgroups behind those of the vertex kernel, in its launch: one
     * kernel less in the chain of a frame drawn and waited for.  Any other batch (passes that may have to be left out,
     * passes run again) starts with rz_begin_kernel. */
    if (b.fast_start) {
        rz_vertex_zero_kernel<<<dim3(vgrid.x + RZ_BEGIN_BLOCKS, n), dim3(256), 0, st>>>(passes, vgrid.x);
    } else {
        rz_begin_kernel<<<dim3(RZ_BEGIN_BLOCKS, n), dim3(256), 0, st>>>(passes);
        rz_vertex_kernel<<<vgrid, dim3(256), 0, st>>>(passes);      /* with the triangles' set-up */
    }
    stamp();
    rz_clip_kernel<<<dim3(256, n), dim3(64), 0, st>>>(passes);
    stamp();
    rz_bin_count_kernel<<<tgrid, dim3(256), 0, st>>>(passes);
    rz_bin_scan_kernel<<<dim3((unsigned)((max_bins + RZ_SCAN_BINS - 1) / RZ_SCAN_BINS), n), dim3(RZ_SCAN_BINS * RZ_SCAN_SEGMENTS), 0, st>>>(passes);
    /* a pass with blended draws: either everything in the ordered kernel (visibility and shading
     * together, fragment stacks), or visibility first, the ordered kernel for the tiles where
     * blended fragments pile up and the shade kernels for the rest */
    const bool ordered_only = b.has_blend && !splits_blend(batch_tiles, n);
    {
        /* a chunk whose triangles lie far apart walks its bins in rounds of 64: with few chunks in the launch (a frame
         * or two) the rounds are shared out over four or eight waves, with many one wave per chunk keeps more chunks in
         * flight.  Not in a flush of several batches per level: beside other lanes' batches, whose workgroups of a wave
         * or two take every wave slot as it frees, workgroups of 256 or 512 threads wait long for a CU with room for
         * all their waves (the 60 small blended passes of a kgirl batch: 8.5 us alone, 75-160 us there) */
        const unsigned fill_blocks = (unsigned)((max_T + 63) / 64 * n);
        const int waves = g.fill_waves ? g.fill_waves : b.beside ? 1 : fill_blocks < 1024u ? 8 : fill_blocks < 4096u ? 4 : 1;
        /* the strips' passes: what a strip has to do with its bin is worked out by the same launch, a wave per bin in
         * workgroups behind those that fill (empty bins are finished there; a pass with a table of sky triangles gets
         * its bins' records for rz_sky_kernel) */
        ClassTables classes;
        classes.tri_base = L.tri.ptr; classes.fan_base = L.fan_tri.ptr; classes.bin_range_base = L.bin_range.ptr;
        classes.sky_base = L.sky_bins.ptr; classes.class_base = L.bin_class.ptr;
        classes.max_bins = (uint32_t)max_bins; classes.fill_blocks = fill_blocks;
        const unsigned class_blocks = b.has_blend ? 0u : (unsigned)((max_bins * n + (size_t)waves - 1) / (size_t)waves);
        const dim3 grid(fill_blocks + class_blocks);
        if (waves == 1) rz_bin_fill_kernel<1><<<grid, dim3(64), 0, st>>>(passes, n, classes);
        else if (waves == 2) rz_bin_fill_kernel<2><<<grid, dim3(128), 0, st>>>(passes, n, classes);
        else if (waves == 8) rz_bin_fill_kernel<8><<<grid, dim3(512), 0, st>>>(passes, n, classes);
        else rz_bin_fill_kernel<4><<<grid, dim3(256), 0, st>>>(passes, n, classes);
    }
    stamp();
    bool narrow = true;
    for (size_t i : b.members) narrow = narrow && plans[i].narrow_texels;
    const int only = b.families == (1 << RZ_FAMILY_BLINN) ? 0 : b.families == (1 << RZ_FAMILY_PBR) ? 1 : -1;
    bool in_strips = false;                         /* the shade lists hold 16x8 blocks */
    bool crowd_forked = false;
    bool sky_with_strips = false;                   /* the strips' launch takes the sky of empty bins along */
    if (ordered_only) {
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
            if (g.verbose) fprintf(stderr, "mi355r: batch of %u passes, longest bin list last time %u entries, strips of %d\n", n, longest, strip);
            snprintf(g.last_raster_kernel, sizeof g.last_raster_kernel, "rz_strip_kernel<%d>", strip);
            /* a launch of a frame or two: a team of waves per block, the sky of empty bins in the same launch
             * (fragment counts follow the order the triangles are taken in: no teams while they are kept) */
            const bool team = strip == 1 && b.members.size() <= 2 && !g.counting && g.strip_team && !b.crowd;
            sky_with_strips = team && b.sky_table && g.sky_with_strips && !g.timing;
            /* passes of one size: a strip then finds its bin's class without reading its pass's parameters */
            int uniform_w = plans[b.members[0]].fb->pub.width, uniform_h = plans[b.members[0]].fb->pub.height;
            for (size_t i : b.members)
                if (plans[i].fb->pub.width != uniform_w || plans[i].fb->pub.height != uniform_h) uniform_w = uniform_h = 0;
            if (b.crowd) {
                /* the blocks of crowded bins (listed by the class kernel) beside the strips of the others */
                HIP_OK(hipEventRecord(L.fork, st));
                HIP_OK(hipStreamWaitEvent(L.side, L.fork, 0));
                auto crowd = g.crowd_split == 2 ? rz_crowd_kernel<2> : g.crowd_split == 4 ? rz_crowd_kernel<4> : g.crowd_split == 16 ? rz_crowd_kernel<16> : rz_crowd_kernel<8>;
                crowd<<<shade_grid, dim3(64u * (unsigned)g.crowd_split), 0, L.side>>>(
                    passes, L.bin_list.ptr, L.bin_boxes.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr, L.queue.ptr, L.queue_counts.ptr, queue_capacity);
                HIP_OK(hipEventRecord(L.join, L.side));
                crowd_forked = true;
            }
            if (team) {
                snprintf(g.last_raster_kernel, sizeof g.last_raster_kernel, "rz_strip_team_kernel<2>");
                SkyTables sky;
                sky.tri_base = L.tri.ptr; sky.fan_base = L.fan_tri.ptr; sky.sky_base = sky_with_strips ? L.sky_bins.ptr : nullptr;
                rz_strip_team_kernel<2><<<dim3(n, bin_columns * 2, bin_rows * strips_per_bin), dim3(256), 0, st>>>(
                    passes, L.bin_list.ptr, L.bin_boxes.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr, L.bin_class.ptr, uniform_w, uniform_h, sky);
            } else {
                kernel<<<dim3(n, bin_columns * 2, bin_rows * strips_per_bin), dim3(128), 0, st>>>(
                    passes, L.bin_list.ptr, L.bin_boxes.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.bin_range.ptr, L.bin_class.ptr, uniform_w, uniform_h);
            }
        }
        stamp();
        if (crowd_forked) HIP_OK(hipStreamWaitEvent(st, L.join, 0));
        bool forked = false;
        if (b.sky_table && !sky_with_strips) {
            hipStream_t sky_stream = st;
            /* (not for a launch of a frame or two: there the fork and the join cost more than the overlap gains --
             * helmet, one frame per flush: 0.172 against 0.159 ms per frame) */
            if (L.side && g.sky_side && !g.timing && !paranoid && b.members.size() > 4) {
                HIP_OK(hipEventRecord(L.fork, st));
                HIP_OK(hipStreamWaitEvent(L.side, L.fork, 0));
                sky_stream = L.side;
                forked = true;
            }
            /* beside the shade kernels the sky keeps its hands off the bins of passes where it may hide a fragment they
             * shade (whichever wrote last would show); those bins follow behind the join */
            auto sky_kernel = b.sky_bytes ? (forked ? rz_sky_side_kernel<2> : rz_sky_kernel<2>) : (forked ? rz_sky_side_float_kernel<2> : rz_sky_float_kernel<2>);
            sky_kernel<<<tiles, dim3(128), 0, sky_stream>>>(passes, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.sky_bins.ptr);
            if (forked) HIP_OK(hipEventRecord(L.join, L.side));
        }
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
        if (forked) {
            HIP_OK(hipStreamWaitEvent(st, L.join, 0));
            /* (hardly ever more than a look at the batch's flag word: a small grid) */
            rz_sky_late_kernel<2><<<dim3(std::min(shade_grid.x, 512u)), dim3(128), 0, st>>>(passes, n, L.tri.ptr, L.fan_tri.ptr, L.vary.ptr, L.fan_vary.ptr, db, L.sky_bins.ptr, L.queue_counts.ptr);
        }
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
            r.sky_bytes = r.sky_bytes && fr.plans[i].sky_bytes;
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
                    b.crowd = wants_crowd(b, plans);
                    /* (a launch of a frame or two is as long as its chain of kernels; in a large batch the zeroing inside the
                     * vertex kernel's launch, at that kernel's occupancy, costs more than the kernel it saves: 1.54 -> 1.66 us
                     * per helmet frame) */
                    b.fast_start = l == 0 && g.fast_start > 0 && (g.fast_start > 1 || b.members.size() <= 4);
                    batches.push_back(std::move(b));
                    b = Batch();
                };
                for (size_t i = 0; i < plans.size(); i++) {
                    if (plans[i].level != l || (int)plans[i].has_blend != blend) continue;
                    if (b.members.empty()) { b.has_blend = blend != 0; b.families = 0; b.sky_table = false; b.sky_bytes = true; }
                    b.members.push_back(i);
                    b.families |= plans[i].families;
                    b.sky_table = b.sky_table || plans[i].sky_draws > 0;
                    b.sky_bytes = b.sky_bytes && plans[i].sky_bytes;
                    if (b.members.size() >= g.max_batch) close();
                }
                close();
            }
        }
        size_t first_level = 0;
        for (const Batch &b : batches) first_level += plans[b.members[0]].level == 0 ? 1 : 0;
        for (Batch &b : batches) b.beside = first_level > 1;
    }
    prof.mark(1);

    fr.at = ring_alloc(total);
    char *hb = g.ring_host + fr.at, *db = g.ring_dev + fr.at;
    memset(hb, 0, fr.flag_at + 256);            /* the passes' control blocks and the flush's word */
    g_attach_jobs_at = fr.attach_jobs_at;
    g_attach_jobs_filled = 0;
    for (auto &pl : plans) fill_pass(pl, hb, db);
    if (fr.pose_jobs) memcpy(hb + fr.pose_jobs_at, pose_jobs.data(), sizeof(PoseJob) * fr.pose_jobs);
    prof.mark(2);
    /* scratch first, for every batch: growing a buffer moves it, and the tables written next hold its address */
    for (auto &b : batches) prepare_batch(b, fr, true, PREPARE_FIRST);
    for (auto &b : batches) prepare_batch(b, fr, false, PREPARE_FIRST);
    prof.mark(3);
    HIP_OK(hipMemcpyAsync(db, hb, total, hipMemcpyHostToDevice, g.copy_stream));    /* (control blocks included: zero) */
    prof.mark(4);
    /* Now the flush before this one: everything above went on while the GPU was still busy with it, and
     * nothing has been launched yet.  If some of its passes had to run again the lanes' scratch may
     * have grown and moved, and the estimates have been raised: this flush's parameters are written
     * and sent once more. */
    if (settle_in_flight(false)) {
        HIP_OK(hipStreamSynchronize(g.copy_stream));
        for (auto &b : batches) prepare_batch(b, fr, true, PREPARE_REWRITE);
        for (auto &b : batches) prepare_batch(b, fr, false, PREPARE_REWRITE);
        HIP_OK(hipMemcpyAsync(db, hb, total, hipMemcpyHostToDevice, g.copy_stream));
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
                bool quads = true;                  /* depth buffers with sides of whole 16s into tiled images */
                for (size_t k = 0; k < count; k++) {
                    const FlushRecord::Copy &w = fr.copies[feeds[at + k]];
                    Tex *t = w.tex;
                    if (w.fb->lane != b.lane) fatal("a recorded copy is not on its framebuffer's lane");
                    for (hipEvent_t e : t->read_events) if (std::find(readers.begin(), readers.end(), e) == readers.end()) readers.push_back(e);
                    t->read_events.clear();
                    if (t->write_event && t->write_lane != b.lane) readers.push_back(t->write_event);
                    jobs.job[k] = copy_job(t, w.fb, w.depth, fr.control_dev(w.producer));
                    max_w = std::max(max_w, t->pub.width); max_h = std::max(max_h, t->pub.height);
                    quads = quads && w.depth && (t->flags & RZ_TEX_TILED) && t->pub.width % 16 == 0 && t->pub.height % 16 == 0;
                }
                for (hipEvent_t e : readers) HIP_OK(hipStreamWaitEvent(L.stream, e, 0));   /* passes that still sample the old contents */
                const dim3 copy_grid((max_w + 15) / 16, (max_h + 15) / 16, (unsigned)count);
                if (quads) rz_depth_to_textures_kernel<<<copy_grid, dim3(64), 0, L.stream>>>(jobs);
                else rz_buffers_to_textures_kernel<<<copy_grid, dim3(256), 0, L.stream>>>(jobs);
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
