/* Host side, part 2 of 6: resources -- framebuffers, textures and their device images (formats, uploads, copies from
 * framebuffers), cube maps and IBL data, programs, meshes, pose sources. */

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

/* The bundle of a material's maps (DrawRec::bundle), made when a draw first uses that set of textures and again when one
 * of their images has been rewritten since; nullptr when the maps do not qualify: fewer than two of them, sizes, layouts
 * or a format that is not 4 bytes wide, or an image that the GPU writes. */
static const uint32_t *material_bundle(rz_texture *const *maps, int count, const DevTex *devs, int words) {
    if (!g.bundle_maps) return nullptr;
    std::array<const void*, 6> key = {};
    Tex *tex[6] = {};
    int present = 0, w = 0, h = 0, layout = 0;
    for (int k = 0; k < count; k++) {
        if (!devs[k].data) continue;
        Tex *t = as_tex(maps[k]);
        if (!(devs[k].flags & RZ_TEX_NARROW_FORMATS) || t->pending_write || t->write_lane >= 0) return nullptr;
        if (present == 0) { w = devs[k].w; h = devs[k].h; layout = devs[k].flags & RZ_TEX_LAYOUT; }
        else if (devs[k].w != w || devs[k].h != h || (devs[k].flags & RZ_TEX_LAYOUT) != layout) return nullptr;
        tex[k] = t;
        key[k] = t;
        present++;
    }
    if (present < 2) return nullptr;
    Bundle &b = g.bundles[key];
    const size_t texels = (size_t)w * (size_t)h;
    bool fresh = b.d != nullptr && b.texels == texels && b.words == words;
    for (int k = 0; k < count; k++) fresh = fresh && b.version[k] == (tex[k] ? tex[k]->image_version : 0ull);
    if (fresh) return b.d;
    /* recorded draws and passes in flight may still read the bundle that is about to be rewritten */
    if (b.d) { graphics_flush(); sync_stream(); }
    if (b.d && (b.texels != texels || b.words != words)) { HIP_OK(hipFree(b.d)); b.d = nullptr; }
    if (!b.d) HIP_OK(hipMalloc((void**)&b.d, texels * (size_t)words * sizeof(uint32_t)));
    b.texels = texels;
    b.words = words;
    BundleMaps from;
    memset(&from, 0, sizeof from);
    for (int k = 0; k < count; k++) {
        from.map[k] = tex[k] ? (const uint32_t*)tex[k]->d_data : nullptr;
        b.version[k] = tex[k] ? tex[k]->image_version : 0ull;
    }
    rz_bundle_kernel<<<dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, g.stream>>>(b.d, from, texels, words);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (g.verbose) fprintf(stderr, "mi355r: bundle of %d maps, %dx%d, %d words per texel\n", present, w, h, words);
    return b.d;
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
    for (auto &kv : g.bundles) if (kv.second.d) HIP_OK(hipFree(kv.second.d));
    g.bundles.clear();
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
