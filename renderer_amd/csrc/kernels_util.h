/* Part of kernels.h (included there, in this order; not a header of its own): framebuffer and texture utilities (K6). */

/* ========================================================================
 * K6  framebuffer / texture utilities
 * ======================================================================== */

/* framebuffer_clear_color / _depth (graphics.c:38-55) when no draw follows */
__global__ void rz_clear_kernel(uint32_t *color, float *depth, size_t n, int do_color, uint32_t c,
                                int do_depth, float d) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (do_color) color[i] = c;
    if (do_depth) depth[i] = d;
}

/* texture_from_depthbuffer (texture.c:147-158): the texel is (d, d, d, 1), stored as one float and
 * expanded on fetch; texture_from_colorbuffer (texture.c:130-145): byte / 255.0f per channel. */

/* one launch for several such copies (grid z = copy): e.g. the recorded copies behind one batch of passes */
#define RZ_COPY_GROUP 32
struct CopyJob { const void *from; void *to; int32_t w, h, flags, is_depth; const uint32_t *producer; };
struct CopyJobs { CopyJob job[RZ_COPY_GROUP]; };
__global__ void rz_buffers_to_textures_kernel(const CopyJobs jobs) {
    const CopyJob &j = jobs.job[blockIdx.z];
    /* a workgroup's 16x16 texels are four tiles of the image's 8x8 layout: a wave takes one of them, so that it writes 256
     * contiguous bytes (a row of 16 texels per 16 lanes wrote them as 32-byte pieces of two tiles: a 2048 x 2048 shadow map
     * copy ran at 3.2 TB/s of read + written bytes) */
    const int tile = threadIdx.x >> 6, within = threadIdx.x & 63;
    int x = blockIdx.x * 16 + (tile & 1) * 8 + (within & 7), y = blockIdx.y * 16 + (tile >> 1) * 8 + (within >> 3);
    if (x >= j.w || y >= j.h) return;
    /* the control block of the pass that rendered the source: when that pass did not run (it is run
     * again later, and this copy after it) the texture keeps what it holds */
    if (j.producer && j.producer[2] != 0u) return;
    if (j.is_depth) {
        ((float*)j.to)[texel_index(j.w, j.flags, x, y)] = ((const float*)j.from)[(size_t)y * j.w + x];
    } else {
        uint32_t c = ((const uint32_t*)j.from)[(size_t)y * j.w + x];
        ((float4*)j.to)[texel_index(j.w, j.flags, x, y)] =
            make_float4((float)(c & 0xffu) / 255.0f, (float)((c >> 8) & 0xffu) / 255.0f,
                        (float)((c >> 16) & 0xffu) / 255.0f, (float)(c >> 24) / 255.0f);
    }
}

/* The same for depth buffers whose sides are multiples of 16 into tiled images, four texels per thread: one wave per
 * 16x16 texels (grid as above, 64 threads), a lane reads 16 bytes of a row and writes them as 16 bytes of a tile.  With
 * a thread per texel the copy of 32 shadow maps of 2048 x 2048 was two million waves of two instructions' work each and
 * ran at the rate waves can be started, 3.2 TB/s of read + written bytes. */
__global__ void __launch_bounds__(64) rz_depth_to_textures_kernel(const CopyJobs jobs) {
    const CopyJob &j = jobs.job[blockIdx.z];
    const int lane = threadIdx.x, row = lane >> 2, quad = lane & 3;
    const int x = blockIdx.x * 16 + quad * 4, y = blockIdx.y * 16 + row;
    if (x >= j.w || y >= j.h) return;
    if (j.producer && j.producer[2] != 0u) return;
    const float4 d = *(const float4*)((const float*)j.from + (size_t)y * j.w + x);
    *(float4*)((float*)j.to + texel_index(j.w, j.flags, x, y)) = d;       /* x is a multiple of 4: the four texels are neighbours in their tile */
}

/* a material's maps side by side (DrawRec::bundle): word k of texel i = map k's texel i, 0 where there is no map */
struct BundleMaps { const uint32_t *map[8]; };
__global__ void rz_bundle_kernel(uint32_t *__restrict__ bundle, const BundleMaps maps, size_t texels, int words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= texels) return;
    for (int k = 0; k < words; k++) bundle[i * (size_t)words + k] = maps.map[k] ? maps.map[k][i] : 0u;
}

/* RGBA8 -> RGB8.  A thread takes sixteen pixels, 64 bytes in as four 16-byte loads and 48 out as three 16-byte stores,
 * when the destination is 16-byte aligned (a frame of a gather buffer is); otherwise, and for the last pixels of a frame
 * that is no multiple of sixteen, four pixels with 4-byte stores or pixel by pixel.  (Four pixels per thread everywhere:
 * 14.5 MB of a 1080p frame moved in 5.2 us.) */
RZ_DEV void strip_alpha_four(uint4 c, uint32_t &o0, uint32_t &o1, uint32_t &o2) {
    const uint32_t a = c.x & 0xffffffu, b = c.y & 0xffffffu, d = c.z & 0xffffffu, e = c.w & 0xffffffu;
    o0 = a | (b << 24);
    o1 = (b >> 8) | (d << 16);
    o2 = (d >> 16) | (e << 8);
}
__global__ void rz_strip_alpha_kernel(const uint32_t *__restrict__ color, uint32_t *__restrict__ rgb, size_t pixels) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;         /* group of sixteen pixels */
    if (t * 16 >= pixels) return;
    const bool aligned = ((uintptr_t)rgb & 15u) == 0;                       /* uniform */
    if (aligned && t * 16 + 16 <= pixels) {
        const uint4 *in = (const uint4*)color + t * 4;
        const uint4 c0 = in[0], c1 = in[1], c2 = in[2], c3 = in[3];
        uint4 o0, o1, o2;
        strip_alpha_four(c0, o0.x, o0.y, o0.z);
        strip_alpha_four(c1, o0.w, o1.x, o1.y);
        strip_alpha_four(c2, o1.z, o1.w, o2.x);
        strip_alpha_four(c3, o2.y, o2.z, o2.w);
        uint4 *out = (uint4*)rgb + t * 3;
        out[0] = o0; out[1] = o1; out[2] = o2;
        return;
    }
    for (size_t q = t * 4; q < t * 4 + 4 && q * 4 < pixels; q++) {         /* groups of four pixels */
        if (q * 4 + 4 <= pixels) {
            uint32_t o0, o1, o2;
            strip_alpha_four(((const uint4*)color)[q], o0, o1, o2);
            rgb[q * 3 + 0] = o0; rgb[q * 3 + 1] = o1; rgb[q * 3 + 2] = o2;
        } else {
            unsigned char *out = (unsigned char*)rgb;
            for (size_t p = q * 4; p < pixels; p++) {
                const uint32_t c = color[p];
                out[p * 3] = (unsigned char)c; out[p * 3 + 1] = (unsigned char)(c >> 8); out[p * 3 + 2] = (unsigned char)(c >> 16);
            }
        }
    }
}

/* The same for a frame of a multiple of sixteen pixels into a 16-byte aligned destination, with every load and every
 * store of a wave one contiguous kilobyte: a workgroup takes 4 096 pixels, lane after lane 16 bytes in, the twelve bytes
 * each makes go to LDS side by side (a stride of three words: no two lanes of a wave on one bank), and come out again
 * lane after lane 16 bytes.  (Sixteen pixels per thread, each lane its own 64 bytes in and 48 out: 4.6 us per 1080p frame.) */
#define RZ_ALPHA_CHUNK 4096u
RZ_DEV void strip_alpha_chunk(const uint4 *__restrict__ color, uint4 *__restrict__ rgb, uint32_t quads, uint32_t chunk, uint32_t *s_out) {
    const uint32_t first = chunk * (RZ_ALPHA_CHUNK / 4u);           /* the workgroup's first group of four pixels */
    const uint32_t here = min(RZ_ALPHA_CHUNK / 4u, quads - first);  /* a multiple of four: whole 16-byte pieces come out */
#pragma unroll
    for (uint32_t turn = 0; turn < RZ_ALPHA_CHUNK / 4u / 256u; turn++) {
        const uint32_t i = turn * 256u + threadIdx.x;
        if (i < here) {
            uint32_t o0, o1, o2;
            strip_alpha_four(color[first + i], o0, o1, o2);
            s_out[3u * i] = o0; s_out[3u * i + 1u] = o1; s_out[3u * i + 2u] = o2;
        }
    }
    __syncthreads();
    uint4 *out = rgb + (size_t)first / 4u * 3u;
    for (uint32_t j = threadIdx.x; j < here / 4u * 3u; j += 256u) out[j] = ((const uint4*)s_out)[j];
}

__global__ void __launch_bounds__(256) rz_strip_alpha_wide_kernel(const uint4 *__restrict__ color, uint4 *__restrict__ rgb, uint32_t quads) {
    __shared__ uint32_t s_out[RZ_ALPHA_CHUNK / 4u * 3u];
    strip_alpha_chunk(color, rgb, quads, blockIdx.x, s_out);
}

/* ... and for up to RZ_ALPHA_FRAMES frames of one size in one launch (the frames of a step on one lane, about to be
 * gathered: launched one by one, each 3.5 us kernel waited for the one before it to drain), blockIdx.y = frame */
#define RZ_ALPHA_FRAMES 32
struct AlphaFrames { const uint4 *color[RZ_ALPHA_FRAMES]; uint4 *rgb[RZ_ALPHA_FRAMES]; };
__global__ void __launch_bounds__(256) rz_strip_alpha_frames_kernel(const AlphaFrames frames, uint32_t quads) {
    __shared__ uint32_t s_out[RZ_ALPHA_CHUNK / 4u * 3u];
    strip_alpha_chunk(frames.color[blockIdx.y], frames.rgb[blockIdx.y], quads, blockIdx.x, s_out);
}

/* private_blit_bgr / private_blit_rgb (private.c:8-50): vertical flip, optional red/blue swap; the
 * fourth byte of the staged pixel is zero (the host keeps the destination's own) */
__global__ void rz_blit_kernel(const uint32_t *__restrict__ color, uint32_t *__restrict__ staged, int w, int h, int swap) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t c = color[(size_t)y * w + x];
    const uint32_t r = c & 0xffu, g = (c >> 8) & 0xffu, b = (c >> 16) & 0xffu;
    staged[(size_t)(h - 1 - y) * w + x] = swap ? (b | (g << 8) | (r << 16)) : (r | (g << 8) | (b << 16));
}

/* 2-D overlays (draw2d.c).  A line segment walks its longer axis pixel by pixel -- one thread per
 * step -- and takes the other coordinate as (int)(a + (b - a) * t), t = step / distance in single
 * precision (draw2d.c:43-73: rows when the row distance is the larger one, columns otherwise, a
 * single pixel when both are zero).  Up to three segments of one colour per launch (a triangle's
 * outline, draw2d.c:114-126): blockIdx.y = segment; where they meet they write the same word. */
struct OverlaySegments { int32_t r0[3], c0[3], r1[3], c1[3]; uint32_t pixel; int32_t width; };
__global__ void __launch_bounds__(64) rz_overlay_segments_kernel(uint32_t *__restrict__ color, const OverlaySegments s) {
    const int k = blockIdx.y;
    const int r0 = s.r0[k], c0 = s.c0[k], r1 = s.r1[k], c1 = s.c1[k];
    const int dr = abs(r0 - r1), dc = abs(c0 - c1);
    const bool along_rows = dr > dc;
    int a0 = along_rows ? r0 : c0, a1 = along_rows ? r1 : c1;
    int b0 = along_rows ? c0 : r0, b1 = along_rows ? c1 : r1;
    const int dist = along_rows ? dr : dc;
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    const int a = a0 + (int)(blockIdx.x * 64 + threadIdx.x);
    if (a > a1) return;
    int b = b0;
    if (dist != 0) {
        const float t = (float)(a - a0) / (float)dist;
        b = (int)((float)b0 + (float)(b1 - b0) * t);
    }
    const int row = along_rows ? a : b, col = along_rows ? b : a;
    color[(size_t)row * s.width + col] = s.pixel;
}

/* draw2d.c:75-93: rows [0, h) x columns [0, w) of the image, from its row 0 on, are added byte-wise
 * (wrapping) to red, green and blue of the pixels from (row, col) on; alpha stays */
__global__ void __launch_bounds__(64) rz_overlay_texture_kernel(uint32_t *__restrict__ color, int fb_width, const void *__restrict__ texels,
                                                                int tex_width, int flags, int row, int col, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const uint32_t i = texel_index(tex_width, flags, x, y);
    f4 t;
    if (flags & RZ_TEX_NARROW) t = decode_word(flags, ((const float*)texels)[i]);
    else { const float4 q = ((const float4*)texels)[i]; t = mk4(q.x, q.y, q.z, q.w); }
    auto u8 = [](float v) { return (uint32_t)(int)(v * 255) & 0xffu; };        /* maths.c:33 */
    uint32_t *p = color + (size_t)(row + y) * fb_width + (col + x);
    const uint32_t d = *p;
    *p = (d & 0xff000000u) | ((d + u8(t.x)) & 0xffu) | (((d >> 8) + u8(t.y)) & 0xffu) << 8 | (((d >> 16) + u8(t.z)) & 0xffu) << 16;
}

/* row-major float4 image -> device layout (8x8 tiles where the size allows) */
__global__ void rz_texture_tile_kernel(const float4 *linear, float4 *texels, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, x, y)] = linear[(size_t)y * w + x];
}

/* the same for the 4-byte packed formats */
__global__ void rz_texture_tile32_kernel(const uint32_t *linear, uint32_t *texels, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    texels[texel_index(w, flags, x, y)] = linear[(size_t)y * w + x];
}

/* device layout -> row-major float4 (texture_sync) */
/* the RZ_TEX_DISPLAY8 plane of an image, texel by texel in the image's own layout */
__global__ void rz_texture_display_kernel(const void *texels, uint32_t *bytes, size_t count, int flags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    f4 c;
    if (flags & RZ_TEX_NARROW) c = decode_word(flags, __uint_as_float(((const uint32_t*)texels)[i]));
    else { const float4 q = ((const float4*)texels)[i]; c = mk4(q.x, q.y, q.z, q.w); }
    c = mk4(d_sat(c.x), d_sat(c.y), d_sat(c.z), d_sat(c.w));
    bytes[i] = pack_rgb(c, 0u);
}

__global__ void rz_texture_untile_kernel(const void *texels, float4 *linear, int w, int h, int flags) {
    int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    uint32_t i = texel_index(w, flags, x, y);
    float4 v;
    if (flags & RZ_TEX_NARROW) {            /* one word per texel: decoded to the floats the reference's texture holds */
        const f4 t = decode_word(flags, ((const float*)texels)[i]);
        v = make_float4(t.x, t.y, t.z, t.w);
    }
    else v = ((const float4*)texels)[i];
    linear[(size_t)y * w + x] = v;
}
