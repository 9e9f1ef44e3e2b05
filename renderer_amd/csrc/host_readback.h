/* Host side, part 5 of 6: read-back and presentation -- framebuffer_sync / _read*, the blit, 2-D overlays, RGB frames
 * for a gather, texture_sync without allocations. */

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

/* a frame's RGBA8 colour as RGB8 at `rgb` (device memory), on `st` */
static void launch_strip_alpha(const uint32_t *color, void *rgb, size_t n, hipStream_t st) {
    if (n % 16 == 0 && n / 4 < 0xffffffffull && ((uintptr_t)rgb & 15u) == 0 && n != 0)
        rz_strip_alpha_wide_kernel<<<dim3((unsigned)((n + RZ_ALPHA_CHUNK - 1) / RZ_ALPHA_CHUNK)), dim3(256), 0, st>>>(
            (const uint4*)color, (uint4*)rgb, (uint32_t)(n / 4));
    else
        rz_strip_alpha_kernel<<<dim3((unsigned)((n / 16 + 1 + 255) / 256)), dim3(256), 0, st>>>(color, (uint32_t*)rgb, n);
}

/* the RGB8 copies of a set of frames, frame i at rgb + i * stride, each on its framebuffer's lane behind what was
 * submitted there; frames of one lane, one size and an aligned destination share launches.  Returns the lanes used. */
static void launch_frames_strip(const std::vector<Fb*> &fbs, char *rgb, size_t stride, bool used[MAX_LANES]) {
    for (int lane = 0; lane < g.num_lanes; lane++) {
        AlphaFrames group;
        unsigned held = 0;
        size_t group_n = 0;
        auto send = [&]() {
            if (!held) return;
            rz_strip_alpha_frames_kernel<<<dim3((unsigned)((group_n + RZ_ALPHA_CHUNK - 1) / RZ_ALPHA_CHUNK), held), dim3(256), 0, g.lanes[lane].stream>>>(group, (uint32_t)(group_n / 4));
            HIP_OK(hipGetLastError());
            held = 0;
        };
        for (size_t i = 0; i < fbs.size(); i++) {
            Fb *fb = fbs[i];
            if ((fb->lane < 0 ? 0 : fb->lane) != lane) continue;
            used[lane] = true;
            const size_t n = (size_t)fb->pub.width * fb->pub.height;
            char *dst = rgb + i * stride;
            const bool wide = n % 16 == 0 && n / 4 < 0xffffffffull && ((uintptr_t)dst & 15u) == 0 && n != 0;
            if (!wide) {
                launch_strip_alpha(fb->d_color, dst, n, g.lanes[lane].stream);
                HIP_OK(hipGetLastError());
                continue;
            }
            if (held && (n != group_n || held == RZ_ALPHA_FRAMES)) send();
            group_n = n;
            group.color[held] = (const uint4*)fb->d_color;
            group.rgb[held] = (uint4*)dst;
            held++;
        }
        send();
    }
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
    launch_strip_alpha(fb->d_color, rgb, n, st);
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
        for (Fb *fb : fbs) apply_pending_clear(fb);
        bool used[MAX_LANES] = {false};
        launch_frames_strip(fbs, (char*)rgb, stride_bytes, used);
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
    launch_frames_strip(r.fbs, r.rgb, r.stride, r.used);
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
