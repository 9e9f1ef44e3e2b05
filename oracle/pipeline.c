/*
 * TEST INFRASTRUCTURE -- CPU oracle of the triangle pipeline.
 *
 * A single-threaded restatement of renderer/core/graphics.c (framebuffer,
 * program, clip, cull, rasterise, early-Z, varyings, blend, write) and of the
 * per-frame parts of renderer/core/texture.c (samplers, depth->texture).  It
 * exports the same functions as the GPU library (include/mi355r.h), so the
 * scene layer in renderer_amd/host links against either; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use this build.
 *
 * Parity pinning: bit-identical to the reference compiled strictly
 * (-O2 -ffp-contract=off) on every golden frame under tests/golden
 * (tests/test_oracle_golden.py).
 */
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "omath.h"
#include "oracle.h"

/* ======================= geometry trace ================================= */

/* What mi355r_inspect_* reports: per input triangle of the current pass its
 * pixel box and the number of triangles rasterised for it.  A pass is the run
 * of draws into one framebuffer since the last clear / read / sync. */
static struct {
    rz_framebuffer *fb;
    int closed, blended;
    int width, height;
    size_t n, cap;
    unsigned *boxes, *fans;
} g_trace;

static int g_prim_box[4];           /* x0, x1, y0, y1 of the triangle raster() saw last */

static void trace_close(rz_framebuffer *fb) {
    if (g_trace.fb == fb) g_trace.closed = 1;
}

static void trace_add(rz_framebuffer *fb, int blend, const int box[4], int fan) {
    size_t t;
    /* a pass also ends in front of its first blended draw (include/mi355r.h) */
    if (g_trace.fb != fb || g_trace.closed || (blend && !g_trace.blended)) {
        g_trace.fb = fb; g_trace.closed = 0; g_trace.n = 0; g_trace.blended = blend;
        g_trace.width = fb->width; g_trace.height = fb->height;
    }
    if (g_trace.n == g_trace.cap) {
        g_trace.cap = g_trace.cap ? g_trace.cap * 2 : 4096;
        g_trace.boxes = (unsigned*)realloc(g_trace.boxes, g_trace.cap * 2 * sizeof(unsigned));
        g_trace.fans = (unsigned*)realloc(g_trace.fans, g_trace.cap * sizeof(unsigned));
    }
    t = g_trace.n++;
    if (fan == 0 || box[0] > box[1] || box[2] > box[3]) {
        g_trace.boxes[2 * t] = 0x0000ffffu; g_trace.boxes[2 * t + 1] = 0x0000ffffu;
        g_trace.fans[t] = 0;
    } else {
        g_trace.boxes[2 * t] = (unsigned)box[0] | ((unsigned)box[1] << 16);
        g_trace.boxes[2 * t + 1] = (unsigned)box[2] | ((unsigned)box[3] << 16);
        g_trace.fans[t] = (unsigned)fan;
    }
}

/* ======================= framebuffer (graphics.c:11-55) ================= */

rz_framebuffer *framebuffer_create(int width, int height) {
    rz_framebuffer *fb;
    rz_vec4 black = {0, 0, 0, 1};
    if (width <= 0 || height <= 0) { fprintf(stderr, "oracle: bad framebuffer size\n"); abort(); }
    fb = (rz_framebuffer*)malloc(sizeof *fb);
    fb->width = width;
    fb->height = height;
    fb->color_buffer = (unsigned char*)malloc((size_t)width * height * 4);
    fb->depth_buffer = (float*)malloc(sizeof(float) * (size_t)width * height);
    framebuffer_clear_color(fb, black);
    framebuffer_clear_depth(fb, 1);
    return fb;
}

void framebuffer_release(rz_framebuffer *fb) {
    free(fb->color_buffer);
    free(fb->depth_buffer);
    free(fb);
}

void framebuffer_clear_color(rz_framebuffer *fb, rz_vec4 color) {
    size_t n = (size_t)fb->width * fb->height, i;
    unsigned char px[4];
    trace_close(fb);
    px[0] = o_to_u8(color.x); px[1] = o_to_u8(color.y);
    px[2] = o_to_u8(color.z); px[3] = o_to_u8(color.w);
    for (i = 0; i < n; i++) memcpy(fb->color_buffer + i * 4, px, 4);
}

void framebuffer_clear_depth(rz_framebuffer *fb, float depth) {
    size_t n = (size_t)fb->width * fb->height, i;
    trace_close(fb);
    for (i = 0; i < n; i++) fb->depth_buffer[i] = depth;
}

void framebuffer_sync(rz_framebuffer *fb) { trace_close(fb); }
void framebuffer_sync_color(rz_framebuffer *fb) { trace_close(fb); }
void graphics_flush(void) {}

void framebuffer_read(rz_framebuffer *fb, void *color, void *depth) {
    size_t n = (size_t)fb->width * fb->height;
    trace_close(fb);
    if (color) memcpy(color, fb->color_buffer, n * 4);
    if (depth) memcpy(depth, fb->depth_buffer, n * sizeof(float));
}

/* private_blit_bgr / private_blit_rgb (private.c:8-50): destination row 0 is the top of the image,
 * three bytes per pixel are written, the fourth is left alone */
void framebuffer_blit(rz_framebuffer *fb, unsigned char *destination, int swap_red_blue) {
    int r, c;
    trace_close(fb);
    for (r = 0; r < fb->height; r++) {
        for (c = 0; c < fb->width; c++) {
            int flipped_r = fb->height - 1 - r;
            const unsigned char *src = &fb->color_buffer[(r * fb->width + c) * 4];
            unsigned char *dst = &destination[(flipped_r * fb->width + c) * 4];
            dst[0] = swap_red_blue ? src[2] : src[0];
            dst[1] = src[1];
            dst[2] = swap_red_blue ? src[0] : src[2];
        }
    }
}

/* ======================= 2-D overlays (draw2d.c) ======================== */

/* draw2d.c:24-28: a point in [0,1]^2 -> the pixel it names, rounded to nearest */
static void overlay_pixel(const rz_framebuffer *fb, rz_vec2 p, int *row, int *col) {
    *row = (int)((fb->height - 1) * o_sat(p.y) + 0.5f);
    *col = (int)((fb->width - 1) * o_sat(p.x) + 0.5f);
}

static void overlay_bytes(rz_vec4 c, unsigned char px[4]) {            /* draw2d.c:17-22 */
    px[0] = o_to_u8(c.x); px[1] = o_to_u8(c.y); px[2] = o_to_u8(c.z); px[3] = o_to_u8(c.w);
}

/* draw2d.c:43-73: the longer axis is walked pixel by pixel, the other one is
 * (int)(a + (b - a) * t) with t = step / distance in single precision; equal
 * distances take the column walk.  All four bytes of a pixel are replaced. */
static void overlay_segment(rz_framebuffer *fb, const unsigned char px[4], int r0, int c0, int r1, int c1) {
    const int dr = abs(r0 - r1), dc = abs(c0 - c1);
    const int along_rows = dr > dc;
    int a0 = along_rows ? r0 : c0, a1 = along_rows ? r1 : c1;      /* the walked axis */
    int b0 = along_rows ? c0 : r0, b1 = along_rows ? c1 : r1;      /* the derived one */
    const int dist = along_rows ? dr : dc;
    int a;
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    for (a = a0; a <= a1; a++) {
        int b = b0, row, col;
        if (dist != 0) {
            float t = (float)(a - a0) / (float)dist;
            b = (int)(b0 + (b1 - b0) * t);
        }
        row = along_rows ? a : b;
        col = along_rows ? b : a;
        memcpy(fb->color_buffer + ((size_t)row * fb->width + col) * 4, px, 4);
    }
}

void draw2d_draw_point(rz_framebuffer *fb, rz_vec4 color, rz_vec2 point) {          /* draw2d.c:95 */
    unsigned char px[4];
    int row, col;
    trace_close(fb);
    overlay_bytes(color, px);
    overlay_pixel(fb, point, &row, &col);
    overlay_segment(fb, px, row, col, row, col);
}

void draw2d_draw_line(rz_framebuffer *fb, rz_vec4 color, rz_vec2 point0, rz_vec2 point1) {  /* draw2d.c:104 */
    unsigned char px[4];
    int r0, c0, r1, c1;
    trace_close(fb);
    overlay_bytes(color, px);
    overlay_pixel(fb, point0, &r0, &c0);
    overlay_pixel(fb, point1, &r1, &c1);
    overlay_segment(fb, px, r0, c0, r1, c1);
}

void draw2d_draw_triangle(rz_framebuffer *fb, rz_vec4 color, rz_vec2 point0, rz_vec2 point1, rz_vec2 point2) {  /* draw2d.c:114 */
    unsigned char px[4];
    int r[3], c[3], k;
    trace_close(fb);
    overlay_bytes(color, px);
    overlay_pixel(fb, point0, &r[0], &c[0]);
    overlay_pixel(fb, point1, &r[1], &c[1]);
    overlay_pixel(fb, point2, &r[2], &c[2]);
    for (k = 0; k < 3; k++) overlay_segment(fb, px, r[k], c[k], r[(k + 1) % 3], c[(k + 1) % 3]);
}

/* draw2d.c:75-93, 128-138: the texture's rows from its row 0 on are ADDED, byte-wise and wrapping,
 * to red, green and blue of the pixels right of and above the origin; what sticks out is cut off */
void draw2d_draw_texture(rz_framebuffer *fb, rz_texture *texture, rz_vec2 origin) {
    int row, col, w, h, r, c;
    trace_close(fb);
    overlay_pixel(fb, origin, &row, &col);
    w = fb->width - col;  if (texture->width < w) w = texture->width;
    h = fb->height - row; if (texture->height < h) h = texture->height;
    for (r = 0; r < h; r++)
        for (c = 0; c < w; c++) {
            const rz_vec4 *s = &texture->buffer[(size_t)r * texture->width + c];
            unsigned char *d = &fb->color_buffer[((size_t)(row + r) * fb->width + (col + c)) * 4];
            d[0] = (unsigned char)(d[0] + o_to_u8(s->x));
            d[1] = (unsigned char)(d[1] + o_to_u8(s->y));
            d[2] = (unsigned char)(d[2] + o_to_u8(s->z));
        }
}

/* ======================= program (graphics.c:59-141) ==================== */

#define MAX_POLY 10             /* a triangle clipped by 7 planes has <= 10 corners */

struct rz_program {
    rz_vertex_shader *vs;
    rz_fragment_shader *fs;
    int attrib_bytes, varying_floats, uniform_bytes;
    int double_sided, blend;
    int family;
    void *attribs[3];
    float *fragment_varyings;
    void *uniforms;
    /* two polygon buffers the clipper ping-pongs between */
    rz_vec4 pos[2][MAX_POLY];
    float *vary[2][MAX_POLY];
};

rz_program *program_create(rz_vertex_shader *vs, rz_fragment_shader *fs,
                           int sizeof_attribs, int sizeof_varyings, int sizeof_uniforms,
                           int double_sided, int enable_blend) {
    rz_program *p;
    int i, s;
    assert(sizeof_attribs > 0 && sizeof_varyings > 0 && sizeof_uniforms > 0);
    assert(sizeof_varyings % (int)sizeof(float) == 0);
    p = (rz_program*)calloc(1, sizeof *p);
    p->vs = vs; p->fs = fs;
    p->attrib_bytes = sizeof_attribs;
    p->varying_floats = sizeof_varyings / (int)sizeof(float);
    p->uniform_bytes = sizeof_uniforms;
    p->double_sided = double_sided;
    p->blend = enable_blend;
    if (vs == blinn_vertex_shader) p->family = RZ_FAMILY_BLINN;
    else if (vs == pbr_vertex_shader) p->family = RZ_FAMILY_PBR;
    else if (vs == skybox_vertex_shader) p->family = RZ_FAMILY_SKYBOX;
    else p->family = -1;
    for (i = 0; i < 3; i++) p->attribs[i] = calloc(1, sizeof_attribs);
    p->fragment_varyings = (float*)calloc(1, sizeof_varyings);
    p->uniforms = calloc(1, sizeof_uniforms);
    for (s = 0; s < 2; s++) for (i = 0; i < MAX_POLY; i++) p->vary[s][i] = (float*)calloc(1, sizeof_varyings);
    return p;
}

void program_release(rz_program *p) {
    int i, s;
    for (i = 0; i < 3; i++) free(p->attribs[i]);
    free(p->fragment_varyings);
    free(p->uniforms);
    for (s = 0; s < 2; s++) for (i = 0; i < MAX_POLY; i++) free(p->vary[s][i]);
    free(p);
}

void *program_get_attribs(rz_program *p, int nth) { assert(nth >= 0 && nth < 3); return p->attribs[nth]; }
void *program_get_uniforms(rz_program *p) { return p->uniforms; }
void program_register_family(rz_vertex_shader *vs, rz_fragment_shader *fs, rz_family family) {
    (void)vs; (void)fs; (void)family;
}

/* ======================= clipping (graphics.c:151-313) ================== */

enum { PLANE_W, PLANE_PX, PLANE_NX, PLANE_PY, PLANE_NY, PLANE_PZ, PLANE_NZ };

static int inside(rz_vec4 v, int plane) {
    switch (plane) {
    case PLANE_W:  return v.w >= O_EPS;
    case PLANE_PX: return v.x <= +v.w;
    case PLANE_NX: return v.x >= -v.w;
    case PLANE_PY: return v.y <= +v.w;
    case PLANE_NY: return v.y >= -v.w;
    case PLANE_PZ: return v.z <= +v.w;
    default:       return v.z >= -v.w;
    }
}

/* parameter of the crossing point on the edge prev->curr (graphics.c:183-203):
 * d_prev / (d_prev - d_curr) with d the plane's signed distance, except for the
 * w plane whose numerator carries the epsilon but whose denominator does not */
static float crossing(rz_vec4 a, rz_vec4 b, int plane) {
    switch (plane) {
    case PLANE_W:  return (a.w - O_EPS) / (a.w - b.w);
    case PLANE_PX: return (a.w - a.x) / ((a.w - a.x) - (b.w - b.x));
    case PLANE_NX: return (a.w + a.x) / ((a.w + a.x) - (b.w + b.x));
    case PLANE_PY: return (a.w - a.y) / ((a.w - a.y) - (b.w - b.y));
    case PLANE_NY: return (a.w + a.y) / ((a.w + a.y) - (b.w + b.y));
    case PLANE_PZ: return (a.w - a.z) / ((a.w - a.z) - (b.w - b.z));
    default:       return (a.w + a.z) / ((a.w + a.z) - (b.w + b.z));
    }
}

/* one Sutherland-Hodgman pass; for every corner, first the crossing of the
 * edge arriving at it (if that edge changes side), then the corner itself if
 * it is inside (graphics.c:205-253) */
static int clip_pass(rz_program *p, int plane, int n, int src) {
    int dst = src ^ 1, out = 0, i, k;
    int nf = p->varying_floats;
    for (i = 0; i < n; i++) {
        int j = (i - 1 + n) % n;
        rz_vec4 a = p->pos[src][j], b = p->pos[src][i];
        const float *va = p->vary[src][j], *vb = p->vary[src][i];
        int a_in = inside(a, plane), b_in = inside(b, plane);
        if (a_in != b_in) {
            float t = crossing(a, b, plane);
            float *vo = p->vary[dst][out];
            p->pos[dst][out] = o_v4(o_mix(a.x, b.x, t), o_mix(a.y, b.y, t),
                                    o_mix(a.z, b.z, t), o_mix(a.w, b.w, t));
            for (k = 0; k < nf; k++) vo[k] = o_mix(va[k], vb[k], t);
            out++;
        }
        if (b_in) {
            p->pos[dst][out] = b;
            memcpy(p->vary[dst][out], vb, sizeof(float) * nf);
            out++;
        }
    }
    assert(out <= MAX_POLY);
    return out;
}

static int fully_visible(rz_vec4 v) {           /* graphics.c:275 */
    return fabs(v.x) <= v.w && fabs(v.y) <= v.w && fabs(v.z) <= v.w;
}

/* returns the corner count and, through *side, which buffer holds the result
 * (graphics.c:279-313: trivial accept, else planes W +X -X +Y -Y +Z -Z) */
static int clip_triangle(rz_program *p, int *side) {
    int n = 3, cur = 0, plane;
    if (fully_visible(p->pos[0][0]) && fully_visible(p->pos[0][1]) && fully_visible(p->pos[0][2])) {
        *side = 0;
        return 3;
    }
    for (plane = PLANE_W; plane <= PLANE_NZ; plane++) {
        n = clip_pass(p, plane, n, cur);
        cur ^= 1;
        if (n < 3) { *side = cur; return 0; }
    }
    *side = cur;
    return n;
}

/* ======================= rasteriser (graphics.c:324-535) ================ */

oracle_stats g_oracle_stats;

/* returns 1 when the triangle was back-face culled */
static int raster(rz_framebuffer *fb, rz_program *p, const rz_vec4 clip[3], float *const vary[3]) {
    int width = fb->width, height = fb->height;
    rz_vec3 ndc[3];
    float sx[3], sy[3], sz[3], rw[3];
    float area2;
    int backface, i, x, y, k;
    int x0, x1, y0, y1;
    float minx, maxx, miny, maxy;
    int nf = p->varying_floats;

    for (i = 0; i < 3; i++) ndc[i] = o_div3(o_xyz(clip[i]), clip[i].w);       /* :486-489 */
    area2 = ndc[0].x * ndc[1].y - ndc[0].y * ndc[1].x +
            ndc[1].x * ndc[2].y - ndc[1].y * ndc[2].x +
            ndc[2].x * ndc[0].y - ndc[2].y * ndc[0].x;                         /* :324-332 */
    backface = area2 <= 0;
    if (backface && !p->double_sided) return 1;
    for (i = 0; i < 3; i++) rw[i] = 1 / clip[i].w;                            /* :499-501 */
    for (i = 0; i < 3; i++) {                                                  /* :338-343 */
        sx[i] = (ndc[i].x + 1) * 0.5f * (float)width;
        sy[i] = (ndc[i].y + 1) * 0.5f * (float)height;
        sz[i] = (ndc[i].z + 1) * 0.5f;
    }
    minx = o_min(o_min(sx[0], sx[1]), sx[2]); maxx = o_max(o_max(sx[0], sx[1]), sx[2]);
    miny = o_min(o_min(sy[0], sy[1]), sy[2]); maxy = o_max(o_max(sy[0], sy[1]), sy[2]);
    x0 = (int)floor(minx); if (x0 < 0) x0 = 0;                                 /* :355-364 */
    y0 = (int)floor(miny); if (y0 < 0) y0 = 0;
    x1 = (int)ceil(maxx); if (x1 > width - 1) x1 = width - 1;
    y1 = (int)ceil(maxy); if (y1 > height - 1) y1 = height - 1;
    g_prim_box[0] = x0; g_prim_box[1] = x1; g_prim_box[2] = y0; g_prim_box[3] = y1;
    g_oracle_stats.prims++;

    for (x = x0; x <= x1; x++) {
        for (y = y0; y <= y1; y++) {
            /* barycentrics at the pixel centre (:385-397) */
            float px = (float)x + 0.5f, py = (float)y + 0.5f;
            float abx = sx[1] - sx[0], aby = sy[1] - sy[0];
            float acx = sx[2] - sx[0], acy = sy[2] - sy[0];
            float apx = px - sx[0], apy = py - sy[0];
            float factor = 1 / (abx * acy - aby * acx);
            float s = (acy * apx - acx * apy) * factor;
            float t = (abx * apy - aby * apx) * factor;
            float w0 = 1 - s - t, w1 = s, w2 = t;
            g_oracle_stats.tests++;
            if (w0 > -O_EPS && w1 > -O_EPS && w2 > -O_EPS) {
                int index = y * width + x;
                float depth = sz[0] * w0 + sz[1] * w1 + sz[2] * w2;           /* :403-408 */
                g_oracle_stats.covered++;
                if (depth <= fb->depth_buffer[index]) {                        /* :523 */
                    /* perspective-correct varyings (:418-435) */
                    float q0 = rw[0] * w0, q1 = rw[1] * w1, q2 = rw[2] * w2;
                    float norm = 1 / (q0 + q1 + q2);
                    float *dst = p->fragment_varyings;
                    rz_vec4 color;
                    int discard = 0;
                    for (k = 0; k < nf; k++) {
                        float sum = vary[0][k] * q0 + vary[1][k] * q1 + vary[2][k] * q2;
                        dst[k] = sum * norm;
                    }
                    /* fragment, blend, write (:437-470) */
                    g_oracle_stats.shaded++;
                    color = p->fs(dst, p->uniforms, &discard, backface);
                    if (discard) continue;
                    color = o_sat4(color);
                    if (p->blend) {
                        unsigned char *d = fb->color_buffer + (size_t)index * 4;
                        color.x = color.x * color.w + o_from_u8(d[0]) * (1 - color.w);
                        color.y = color.y * color.w + o_from_u8(d[1]) * (1 - color.w);
                        color.z = color.z * color.w + o_from_u8(d[2]) * (1 - color.w);
                    }
                    fb->color_buffer[(size_t)index * 4 + 0] = o_to_u8(color.x);
                    fb->color_buffer[(size_t)index * 4 + 1] = o_to_u8(color.y);
                    fb->color_buffer[(size_t)index * 4 + 2] = o_to_u8(color.z);
                    fb->depth_buffer[index] = depth;
                }
            }
        }
    }
    return 0;
}

/* graphics.c:537-608 */
void graphics_draw_triangle(rz_framebuffer *fb, rz_program *p) {
    int n, side, i, fan = 0;
    int box[4] = {65535, -1, 65535, -1};    /* union of the fan's non-empty pixel boxes */
    g_oracle_stats.triangles++;
    for (i = 0; i < 3; i++) p->pos[0][i] = p->vs(p->attribs[i], p->vary[0][i], p->uniforms);
    n = clip_triangle(p, &side);
    for (i = 0; i + 2 < n; i++) {
        rz_vec4 c[3];
        float *v[3];
        c[0] = p->pos[side][0]; c[1] = p->pos[side][i + 1]; c[2] = p->pos[side][i + 2];
        v[0] = p->vary[side][0]; v[1] = p->vary[side][i + 1]; v[2] = p->vary[side][i + 2];
        if (raster(fb, p, c, v)) break;     /* a culled fan triangle ends the fan */
        fan++;
        if (g_prim_box[0] <= g_prim_box[1] && g_prim_box[2] <= g_prim_box[3]) {
            if (g_prim_box[0] < box[0]) box[0] = g_prim_box[0];
            if (g_prim_box[1] > box[1]) box[1] = g_prim_box[1];
            if (g_prim_box[2] < box[2]) box[2] = g_prim_box[2];
            if (g_prim_box[3] > box[3]) box[3] = g_prim_box[3];
        }
    }
    trace_add(fb, p->blend, box, fan);
}

/* the per-face loops of the shaders' draw_model (blinn_shader.c:323-341,
 * pbr_shader.c:561-577, skybox_shader.c:57-64) */
void graphics_draw_mesh(rz_framebuffer *fb, rz_program *p, const rz_vertex *verts, int num_faces) {
    int f, j;
    for (f = 0; f < num_faces; f++) {
        for (j = 0; j < 3; j++) {
            const rz_vertex *v = &verts[f * 3 + j];
            if (p->family == RZ_FAMILY_BLINN) {
                rz_blinn_attribs *a = (rz_blinn_attribs*)p->attribs[j];
                a->position = v->position; a->texcoord = v->texcoord; a->normal = v->normal;
                a->joint = v->joint; a->weight = v->weight;
            } else if (p->family == RZ_FAMILY_PBR) {
                *(rz_vertex*)p->attribs[j] = *v;
            } else if (p->family == RZ_FAMILY_SKYBOX) {
                ((rz_skybox_attribs*)p->attribs[j])->position = v->position;
            } else {
                fprintf(stderr, "oracle: graphics_draw_mesh on an unknown shader family\n");
                abort();
            }
        }
        graphics_draw_triangle(fb, p);
    }
}

void mesh_invalidate(const rz_vertex *verts) { (void)verts; }

/* ======================= textures (texture.c) =========================== */

rz_texture *texture_create(int width, int height) {                   /* :12 */
    rz_texture *t;
    assert(width > 0 && height > 0);
    t = (rz_texture*)malloc(sizeof *t);
    t->width = width; t->height = height;
    t->buffer = (rz_vec4*)calloc((size_t)width * height, sizeof(rz_vec4));
    return t;
}

void texture_release(rz_texture *t) { free(t->buffer); free(t); }    /* :27 */
/* the oracle works on the reference's float images: the loader fills texture_create's buffer */
rz_texture *texture_create_packed(int width, int height, int format, const unsigned int *words) {
    (void)width; (void)height; (void)format; (void)words;
    return NULL;
}
void texture_sync(rz_texture *t) { (void)t; }
void texture_invalidate(rz_texture *t) { (void)t; }

void texture_from_colorbuffer(rz_texture *t, rz_framebuffer *fb) {    /* :130 */
    size_t n = (size_t)t->width * t->height, i;
    assert(t->width == fb->width && t->height == fb->height);
    for (i = 0; i < n; i++) {
        const unsigned char *c = fb->color_buffer + i * 4;
        t->buffer[i] = o_v4(o_from_u8(c[0]), o_from_u8(c[1]), o_from_u8(c[2]), o_from_u8(c[3]));
    }
}

void texture_from_depthbuffer(rz_texture *t, rz_framebuffer *fb) {    /* :147 */
    size_t n = (size_t)t->width * t->height, i;
    assert(t->width == fb->width && t->height == fb->height);
    for (i = 0; i < n; i++) {
        float d = fb->depth_buffer[i];
        t->buffer[i] = o_v4(d, d, d, 1);
    }
}

/* nearest texel; index = (int)((size - 1) * coordinate) (:160-176) */
rz_vec4 texture_repeat_sample(rz_texture *t, rz_vec2 uv) {
    float u = uv.x - (float)floor(uv.x);
    float v = uv.y - (float)floor(uv.y);
    int c = (int)((t->width - 1) * u);
    int r = (int)((t->height - 1) * v);
    return t->buffer[r * t->width + c];
}

rz_vec4 texture_clamp_sample(rz_texture *t, rz_vec2 uv) {
    float u = o_sat(uv.x);
    float v = o_sat(uv.y);
    int c = (int)((t->width - 1) * u);
    int r = (int)((t->height - 1) * v);
    return t->buffer[r * t->width + c];
}

rz_vec4 texture_sample(rz_texture *t, rz_vec2 uv) { return texture_repeat_sample(t, uv); }

/* major-axis face selection with strict comparisons, so ties fall through
 * x -> y -> z (:210-255); returns the face and its (s, t) in [0, 1] */
static int cube_face(rz_vec3 d, rz_vec2 *st) {
    float ax = (float)fabs(d.x), ay = (float)fabs(d.y), az = (float)fabs(d.z);
    float ma, sc, tc;
    int face;
    if (ax > ay && ax > az) {
        ma = ax;
        if (d.x > 0) { face = 0; sc = -d.z; tc = -d.y; }
        else         { face = 1; sc = +d.z; tc = -d.y; }
    } else if (ay > az) {
        ma = ay;
        if (d.y > 0) { face = 2; sc = +d.x; tc = +d.z; }
        else         { face = 3; sc = +d.x; tc = -d.z; }
    } else {
        ma = az;
        if (d.z > 0) { face = 4; sc = +d.x; tc = -d.y; }
        else         { face = 5; sc = -d.x; tc = -d.y; }
    }
    st->x = (sc / ma + 1) / 2;
    st->y = (tc / ma + 1) / 2;
    return face;
}

rz_vec4 cubemap_repeat_sample(rz_cubemap *cube, rz_vec3 dir) {        /* :257 */
    rz_vec2 st;
    int face = cube_face(dir, &st);
    st.y = 1 - st.y;
    return texture_repeat_sample(cube->faces[face], st);
}

rz_vec4 cubemap_clamp_sample(rz_cubemap *cube, rz_vec3 dir) {         /* :264 */
    rz_vec2 st;
    int face = cube_face(dir, &st);
    st.y = 1 - st.y;
    return texture_clamp_sample(cube->faces[face], st);
}

rz_vec4 cubemap_sample(rz_cubemap *cube, rz_vec3 dir) { return cubemap_repeat_sample(cube, dir); }

/* ======================= introspection ================================== */

#define ORACLE_BIN_EDGE 64

int mi355r_inspect_pass(rz_pass_info *info) {
    size_t bins;
    unsigned *counts;
    if (!g_trace.fb || g_trace.n == 0) return 0;
    info->num_tris = (int)g_trace.n;
    info->width = g_trace.width;
    info->height = g_trace.height;
    info->bin_edge = ORACLE_BIN_EDGE;
    info->bins_x = (g_trace.width + ORACLE_BIN_EDGE - 1) / ORACLE_BIN_EDGE;
    info->bins_y = (g_trace.height + ORACLE_BIN_EDGE - 1) / ORACLE_BIN_EDGE;
    bins = (size_t)info->bins_x * info->bins_y;
    counts = (unsigned*)malloc(bins * sizeof(unsigned));
    info->bin_entries = (unsigned)oracle_bin(g_trace.boxes, g_trace.fans, (int)g_trace.n, g_trace.width, g_trace.height,
                                             ORACLE_BIN_EDGE, counts, NULL);
    free(counts);
    return 1;
}

void mi355r_inspect_boxes(unsigned int *boxes, unsigned int *fans) {
    memcpy(boxes, g_trace.boxes, g_trace.n * 2 * sizeof(unsigned));
    memcpy(fans, g_trace.fans, g_trace.n * sizeof(unsigned));
}

void mi355r_inspect_bins(unsigned int *counts, unsigned int *entries) {
    oracle_bin(g_trace.boxes, g_trace.fans, (int)g_trace.n, g_trace.width, g_trace.height, ORACLE_BIN_EDGE, counts, entries);
}

const char *mi355r_backend(void) { return "cpu-oracle"; }
void oracle_get_stats(oracle_stats *out) { *out = g_oracle_stats; }
void oracle_reset_stats(void) { memset(&g_oracle_stats, 0, sizeof g_oracle_stats); }
