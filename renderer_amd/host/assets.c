/*
 * Asset ingest: images, textures from images, meshes, skeletons, caches.
 *
 * File formats and the texel / vertex values derived from them follow the
 * reference loaders (renderer/core/image.c, mesh.c, skeleton.c, texture.c:32-128,
 * renderer/shaders/cache_helper.c); citations are per function.  The results
 * feed the GPU pipeline and, in the test build, the CPU oracle, so value
 * conversion order matters and is kept.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "assets.h"
#include "hmath.h"

static void die(const char *what, const char *name) {
    fprintf(stderr, "mi355r assets: %s: %s\n", what, name ? name : "");
    abort();
}

static const char *extension_of(const char *filename) {
    const char *dot = strrchr(filename, '.');
    return dot ? dot + 1 : "";
}

/* relative names resolve against the assets directory (the reference chdirs
 * into it instead, platforms/linux.c:52-63) */
static char g_assets_dir[1024] = ".";

void rzh_set_assets_dir(const char *dir) {
    snprintf(g_assets_dir, sizeof g_assets_dir, "%s", dir);
}

FILE *rzh_open_asset(const char *filename) {
    char path[2048];
    FILE *f;
    if (filename[0] == '/') snprintf(path, sizeof path, "%s", filename);
    else snprintf(path, sizeof path, "%s/%s", g_assets_dir, filename);
    f = fopen(path, "rb");
    if (!f) die("cannot open", path);
    return f;
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) die("out of memory", NULL);
    return p;
}

/* =========================== images ===================================== */

static rzh_image *image_new(int w, int h, int channels, int is_hdr) {
    rzh_image *im = (rzh_image*)xmalloc(sizeof *im);
    size_t n = (size_t)w * h * channels;
    im->is_hdr = is_hdr; im->width = w; im->height = h; im->channels = channels;
    im->ldr = NULL; im->hdr = NULL; im->rgbe = NULL;
    if (is_hdr) { im->hdr = (float*)xmalloc(n * sizeof(float)); memset(im->hdr, 0, n * sizeof(float)); }
    else { im->ldr = (unsigned char*)xmalloc(n); memset(im->ldr, 0, n); }
    return im;
}

void rzh_image_free(rzh_image *im) {
    if (!im) return;
    free(im->ldr); free(im->hdr); free(im->rgbe); free(im);
}

/* whole file in memory: the decoders below walk a byte cursor */
typedef struct { unsigned char *data; size_t size, pos; const char *name; } cursor_t;

static cursor_t slurp(const char *filename) {
    cursor_t c;
    FILE *f = rzh_open_asset(filename);
    long n;
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    c.data = (unsigned char*)xmalloc((size_t)n + 1);
    if (fread(c.data, 1, (size_t)n, f) != (size_t)n) die("short read", filename);
    fclose(f);
    c.data[n] = 0; c.size = (size_t)n; c.pos = 0; c.name = filename;
    return c;
}

static unsigned char next_byte(cursor_t *c) {
    if (c->pos >= c->size) die("truncated file", c->name);
    return c->data[c->pos++];
}

/* TGA: image.c:172-265.  8/24/32 bpp, types 2/3 (raw) and 10/11 (RLE); the
 * descriptor's origin bits flip the image so that row 0 ends up at the bottom
 * and column 0 at the left; BGR(A) is swapped to RGB(A). */
static rzh_image *load_tga(const char *filename) {
    cursor_t c = slurp(filename);
    unsigned char *h = c.data;
    int w, hgt, depth, type, ch, rle, flip_h, flip_v;
    size_t total, filled = 0;
    rzh_image *im;
    if (c.size < 18) die("bad tga", filename);
    w = h[12] | (h[13] << 8); hgt = h[14] | (h[15] << 8);
    depth = h[16]; type = h[2];
    if (w <= 0 || hgt <= 0 || h[0] != 0) die("bad tga header", filename);
    if (depth != 8 && depth != 24 && depth != 32) die("unsupported tga depth", filename);
    if (type != 2 && type != 3 && type != 10 && type != 11) die("unsupported tga type", filename);
    ch = depth / 8; rle = type >= 10;
    flip_h = h[17] & 0x10; flip_v = h[17] & 0x20;
    c.pos = 18;
    im = image_new(w, hgt, ch, 0);
    total = (size_t)w * hgt * ch;
    if (!rle) {
        if (c.size - c.pos < total) die("truncated tga", filename);
        memcpy(im->ldr, c.data + c.pos, total);
    } else {
        while (filled < total) {
            unsigned char head = next_byte(&c);
            int count = (head & 0x7f) + 1, i, j;
            if (filled + (size_t)count * ch > total) die("tga rle overrun", filename);
            if (head & 0x80) {
                unsigned char px[4];
                for (j = 0; j < ch; j++) px[j] = next_byte(&c);
                for (i = 0; i < count; i++) for (j = 0; j < ch; j++) im->ldr[filled++] = px[j];
            } else {
                for (i = 0; i < count * ch; i++) im->ldr[filled++] = next_byte(&c);
            }
        }
    }
    free(c.data);
    if (flip_h) {
        int r, col, k;
        for (r = 0; r < hgt; r++) for (col = 0; col < w / 2; col++) {
            unsigned char *a = im->ldr + ((size_t)r * w + col) * ch;
            unsigned char *b = im->ldr + ((size_t)r * w + (w - 1 - col)) * ch;
            for (k = 0; k < ch; k++) { unsigned char t = a[k]; a[k] = b[k]; b[k] = t; }
        }
    }
    if (flip_v) {
        int r, col, k;
        for (r = 0; r < hgt / 2; r++) for (col = 0; col < w; col++) {
            unsigned char *a = im->ldr + ((size_t)r * w + col) * ch;
            unsigned char *b = im->ldr + ((size_t)(hgt - 1 - r) * w + col) * ch;
            for (k = 0; k < ch; k++) { unsigned char t = a[k]; a[k] = b[k]; b[k] = t; }
        }
    }
    if (ch >= 3) {
        size_t i, n = (size_t)w * hgt;
        for (i = 0; i < n; i++) {
            unsigned char *p = im->ldr + i * ch, t = p[0];
            p[0] = p[2]; p[2] = t;
        }
    }
    return im;
}

/* image.c:315-329: value = mantissa * 2^(e-128) / 256; both factors are exact */
static void rgbe_decode(const unsigned char rgbe[4], float out[3]) {
    if (rgbe[3] == 0) {
        out[0] = out[1] = out[2] = 0;
    } else {
        float factor = (float)ldexp(1.0 / 256, (int)rgbe[3] - 128);
        out[0] = (float)rgbe[0] * factor;
        out[1] = (float)rgbe[1] * factor;
        out[2] = (float)rgbe[2] * factor;
    }
}

static int read_text_line(cursor_t *c, char *line, int cap) {
    int n = 0;
    if (c->pos >= c->size) return 0;
    while (c->pos < c->size && n < cap - 1) {
        char ch = (char)c->data[c->pos++];
        line[n++] = ch;
        if (ch == '\n') break;
    }
    line[n] = 0;
    return 1;
}

/* Radiance HDR: image.c:349-472.  Scanlines are stored top-down; the image is
 * kept bottom-up.  New-style RLE scanlines start with 2 2 hi lo. */
static rzh_image *load_hdr(const char *filename) {
    cursor_t c = slurp(filename);
    char line[RZ_LINE];
    int w = 0, h = 0, have_format = 0, i, x, k;
    rzh_image *im;
    unsigned char *chan;
    if (!read_text_line(&c, line, RZ_LINE) || strncmp(line, "#?", 2) != 0) die("bad hdr magic", filename);
    for (;;) {
        if (!read_text_line(&c, line, RZ_LINE)) die("bad hdr header", filename);
        if (line[0] == '\n' && line[1] == 0) break;
        if (strncmp(line, "FORMAT=", 7) == 0) {
            if (strncmp(line, "FORMAT=32-bit_rle_rgbe", 22) != 0) die("unsupported hdr format", filename);
            have_format = 1;
        } else if (strncmp(line, "GAMMA=", 6) == 0 || strncmp(line, "EXPOSURE=", 9) == 0 || line[0] == '#') {
            /* ignored */
        } else {
            die("unexpected hdr header line", filename);
        }
    }
    if (!have_format) die("hdr without FORMAT", filename);
    if (!read_text_line(&c, line, RZ_LINE) || sscanf(line, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
        die("bad hdr resolution", filename);
    im = image_new(w, h, 3, 1);
    im->rgbe = (unsigned char*)xmalloc((size_t)w * h * 4);
    chan = (unsigned char*)xmalloc((size_t)w * 4);
    for (i = 0; i < h; i++) {
        float *row = im->hdr + (size_t)(h - 1 - i) * w * 3;
        unsigned char *quads = im->rgbe + (size_t)(h - 1 - i) * w * 4;
        int rle = 0;
        if (w >= 8 && w <= 0x7fff) {
            if (c.size - c.pos < 4) die("truncated hdr", filename);
            if (c.data[c.pos] == 2 && c.data[c.pos + 1] == 2 && !(c.data[c.pos + 2] & 0x80)) {
                if (c.data[c.pos + 2] * 256 + c.data[c.pos + 3] != w) die("hdr scanline width", filename);
                c.pos += 4;
                rle = 1;
            }
        }
        if (rle) {
            for (k = 0; k < 4; k++) {
                int filled = 0;
                while (filled < w) {
                    unsigned char b = next_byte(&c);
                    int count, j;
                    if (b > 128) {
                        unsigned char v = next_byte(&c);
                        count = b - 128;
                        if (filled + count > w) die("hdr rle overrun", filename);
                        for (j = 0; j < count; j++) chan[k * w + filled++] = v;
                    } else {
                        count = b;
                        if (count == 0 || filled + count > w) die("hdr rle overrun", filename);
                        for (j = 0; j < count; j++) chan[k * w + filled++] = next_byte(&c);
                    }
                }
            }
            for (x = 0; x < w; x++) {
                unsigned char px[4];
                for (k = 0; k < 4; k++) px[k] = chan[k * w + x];
                memcpy(quads + x * 4, px, 4);
                rgbe_decode(px, row + x * 3);
            }
        } else {
            for (x = 0; x < w; x++) {
                unsigned char px[4];
                for (k = 0; k < 4; k++) px[k] = next_byte(&c);
                memcpy(quads + x * 4, px, 4);
                rgbe_decode(px, row + x * 3);
            }
        }
    }
    free(chan);
    free(c.data);
    return im;
}

rzh_image *rzh_image_load(const char *filename) {
    const char *ext = extension_of(filename);
    if (strcmp(ext, "tga") == 0) return load_tga(filename);
    if (strcmp(ext, "hdr") == 0) return load_hdr(filename);
    die("unknown image type", filename);
    return NULL;
}

/* =========================== textures from images ======================= */

/* maths.c:49-57 */
static float tone_aces(float v) {
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    v = (v * (a * v + b)) / (v * (c * v + d) + e);
    return hm_saturate(v);
}

/* texture.c:32-128: expand to float RGBA (luminance replicated, alpha 1 when
 * absent); an LDR file used as HDR colour is decoded with pow(v, 2.2); an HDR
 * file used as LDR colour is tone-mapped (ACES) and encoded with pow(v, 1/2.2). */
static void fill_texture_from_image(rz_texture *tex, const rzh_image *im, rz_usage usage) {
    size_t n = (size_t)im->width * im->height, i;
    int ch = im->channels;
    for (i = 0; i < n; i++) {
        rz_vec4 t = {0, 0, 0, 1};
        if (!im->is_hdr) {
            const unsigned char *p = im->ldr + i * ch;
            if (ch <= 2) {
                t.x = t.y = t.z = p[0] / 255.0f;
                if (ch == 2) t.w = p[1] / 255.0f;
            } else {
                t.x = p[0] / 255.0f; t.y = p[1] / 255.0f; t.z = p[2] / 255.0f;
                if (ch == 4) t.w = p[3] / 255.0f;
            }
            if (usage == RZ_USAGE_HDR_COLOR) {
                t.x = (float)pow(t.x, 2.2); t.y = (float)pow(t.y, 2.2); t.z = (float)pow(t.z, 2.2);
            }
        } else {
            const float *p = im->hdr + i * ch;
            if (ch <= 2) {
                t.x = t.y = t.z = p[0];
                if (ch == 2) t.w = p[1];
            } else {
                t.x = p[0]; t.y = p[1]; t.z = p[2];
                if (ch == 4) t.w = p[3];
            }
            if (usage == RZ_USAGE_LDR_COLOR) {
                t.x = (float)pow(tone_aces(t.x), 1 / 2.2);
                t.y = (float)pow(tone_aces(t.y), 1 / 2.2);
                t.z = (float)pow(tone_aces(t.z), 1 / 2.2);
            }
        }
        tex->buffer[i] = t;
    }
}

/* The image in one of the pipeline's 4-byte formats, straight from the file's bytes -- what
 * fill_texture_from_image computes is then a function of those bytes alone:
 *   8-bit file, data or LDR colour usage: channel = byte / 255.0f                          -> format 4
 *   8-bit file, HDR colour usage: rgb = (float)pow(byte / 255.0f, 2.2), alpha = byte / 255.0f -> format 8
 *   .hdr file, any usage but LDR colour: rgb = mantissa * 2^(e - 136) (rgbe_decode), alpha 1  -> format 16
 * (luminance replicated, absent alpha = 255).  An .hdr file used as LDR colour is tone-mapped and has
 * no such format.  Returns the format and the words, or 0. */
static int packed_image(const rzh_image *im, rz_usage usage, unsigned int **words_out) {
    size_t n = (size_t)im->width * im->height, i;
    unsigned int *words;
    int ch = im->channels;
    if (im->is_hdr && (usage == RZ_USAGE_LDR_COLOR || ch != 3 || !im->rgbe)) return 0;
    words = (unsigned int*)xmalloc(n * sizeof *words);
    if (im->is_hdr) {
        for (i = 0; i < n; i++) {
            const unsigned char *q = im->rgbe + i * 4;
            /* exponent byte 0 means black whatever the mantissas say (image.c:320-322) */
            words[i] = q[3] == 0 ? 0u : (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
        }
        *words_out = words;
        return 16;
    }
    for (i = 0; i < n; i++) {
        const unsigned char *p = im->ldr + i * ch;
        unsigned r, g, b, a = 255;
        if (ch <= 2) { r = g = b = p[0]; if (ch == 2) a = p[1]; }
        else { r = p[0]; g = p[1]; b = p[2]; if (ch == 4) a = p[3]; }
        words[i] = r | (g << 8) | (b << 16) | (a << 24);
    }
    *words_out = words;
    return usage == RZ_USAGE_HDR_COLOR ? 8 : 4;
}

/* texture.c:108 */
rz_texture *texture_from_file(const char *filename, rz_usage usage) {
    rzh_image *im = rzh_image_load(filename);
    rz_texture *tex = NULL;
    unsigned int *words = NULL;
    int format = packed_image(im, usage, &words);
    if (format) {
        /* a pipeline that keeps such images takes the bytes as they are; one that works on floats
         * (the CPU oracle) says no and gets the reference's float image below */
        tex = texture_create_packed(im->width, im->height, format, words);
        free(words);
        if (tex) { rzh_image_free(im); return tex; }
    }
    tex = texture_create(im->width, im->height);
    fill_texture_from_image(tex, im, usage);
    texture_invalidate(tex);
    rzh_image_free(im);
    return tex;
}

/* texture.c:184 */
rz_cubemap *cubemap_from_files(const char *px, const char *nx, const char *py,
                               const char *ny, const char *pz, const char *nz,
                               rz_usage usage) {
    rz_cubemap *cube = (rz_cubemap*)xmalloc(sizeof *cube);
    cube->faces[0] = texture_from_file(px, usage);
    cube->faces[1] = texture_from_file(nx, usage);
    cube->faces[2] = texture_from_file(py, usage);
    cube->faces[3] = texture_from_file(ny, usage);
    cube->faces[4] = texture_from_file(pz, usage);
    cube->faces[5] = texture_from_file(nz, usage);
    return cube;
}

/* texture.c:198 */
void cubemap_release(rz_cubemap *cube) {
    int i;
    for (i = 0; i < 6; i++) texture_release(cube->faces[i]);
    free(cube);
}

/* =========================== meshes ===================================== */

typedef struct { void *data; size_t count, cap, elem; } grow_t;

static void *grow_push(grow_t *g, size_t elem) {
    if (g->count == g->cap) {
        g->cap = g->cap ? g->cap * 2 : 1024;
        g->data = realloc(g->data, g->cap * elem);
        if (!g->data) die("out of memory", NULL);
    }
    g->elem = elem;
    return (char*)g->data + (g->count++) * elem;
}

/* mesh.c:84-171: "v", "vt", "vn", "f a/b/c ..." plus per-position
 * "# ext.tangent|joint|weight x y z w" comment lines; faces are de-indexed
 * into a flat vertex stream, defaults tangent (1,0,0,1), joint 0, weight 0;
 * centre = midpoint of the position bounding box (mesh.c:19-82). */
rzh_mesh *rzh_mesh_load(const char *filename) {
    grow_t pos = {0}, uv = {0}, nrm = {0}, tan = {0}, jnt = {0}, wgt = {0}, idx = {0};
    char line[RZ_LINE];
    rzh_mesh *mesh;
    rz_vec3 lo = {1e6f, 1e6f, 1e6f}, hi = {-1e6f, -1e6f, -1e6f};
    size_t i, nidx;
    FILE *f;
    if (strcmp(extension_of(filename), "obj") != 0) die("unknown mesh type", filename);
    f = rzh_open_asset(filename);
    while (fgets(line, RZ_LINE, f)) {
        if (line[0] == 'v' && line[1] == ' ') {
            rz_vec3 *p = (rz_vec3*)grow_push(&pos, sizeof *p);
            if (sscanf(line, "v %f %f %f", &p->x, &p->y, &p->z) != 3) die("bad v line", filename);
        } else if (line[0] == 'v' && line[1] == 't' && line[2] == ' ') {
            rz_vec2 *p = (rz_vec2*)grow_push(&uv, sizeof *p);
            if (sscanf(line, "vt %f %f", &p->x, &p->y) != 2) die("bad vt line", filename);
        } else if (line[0] == 'v' && line[1] == 'n' && line[2] == ' ') {
            rz_vec3 *p = (rz_vec3*)grow_push(&nrm, sizeof *p);
            if (sscanf(line, "vn %f %f %f", &p->x, &p->y, &p->z) != 3) die("bad vn line", filename);
        } else if (line[0] == 'f' && line[1] == ' ') {
            int v[9], k;
            if (sscanf(line, "f %d/%d/%d %d/%d/%d %d/%d/%d", &v[0], &v[1], &v[2],
                       &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]) != 9) die("bad f line", filename);
            for (k = 0; k < 9; k++) *(int*)grow_push(&idx, sizeof(int)) = v[k] - 1;
        } else if (strncmp(line, "# ext.tangent ", 14) == 0) {
            rz_vec4 *p = (rz_vec4*)grow_push(&tan, sizeof *p);
            if (sscanf(line, "# ext.tangent %f %f %f %f", &p->x, &p->y, &p->z, &p->w) != 4) die("bad tangent", filename);
        } else if (strncmp(line, "# ext.joint ", 12) == 0) {
            rz_vec4 *p = (rz_vec4*)grow_push(&jnt, sizeof *p);
            if (sscanf(line, "# ext.joint %f %f %f %f", &p->x, &p->y, &p->z, &p->w) != 4) die("bad joint", filename);
        } else if (strncmp(line, "# ext.weight ", 13) == 0) {
            rz_vec4 *p = (rz_vec4*)grow_push(&wgt, sizeof *p);
            if (sscanf(line, "# ext.weight %f %f %f %f", &p->x, &p->y, &p->z, &p->w) != 4) die("bad weight", filename);
        }
    }
    fclose(f);
    nidx = idx.count / 3;                   /* (position, texcoord, normal) triples */
    if (nidx == 0 || nidx % 3 != 0) die("mesh without faces", filename);
    mesh = (rzh_mesh*)xmalloc(sizeof *mesh);
    mesh->num_faces = (int)(nidx / 3);
    mesh->vertices = (rz_vertex*)xmalloc(nidx * sizeof(rz_vertex));
    for (i = 0; i < nidx; i++) {
        const int *t = (const int*)idx.data + i * 3;
        rz_vertex *v = &mesh->vertices[i];
        if (t[0] < 0 || (size_t)t[0] >= pos.count || t[1] < 0 || (size_t)t[1] >= uv.count ||
            t[2] < 0 || (size_t)t[2] >= nrm.count) die("face index out of range", filename);
        v->position = ((rz_vec3*)pos.data)[t[0]];
        v->texcoord = ((rz_vec2*)uv.data)[t[1]];
        v->normal = ((rz_vec3*)nrm.data)[t[2]];
        if (tan.count) { if ((size_t)t[0] >= tan.count) die("tangent index", filename); v->tangent = ((rz_vec4*)tan.data)[t[0]]; }
        else v->tangent = hm_v4(1, 0, 0, 1);
        if (jnt.count) { if ((size_t)t[0] >= jnt.count) die("joint index", filename); v->joint = ((rz_vec4*)jnt.data)[t[0]]; }
        else v->joint = hm_v4(0, 0, 0, 0);
        if (wgt.count) { if ((size_t)t[0] >= wgt.count) die("weight index", filename); v->weight = ((rz_vec4*)wgt.data)[t[0]]; }
        else v->weight = hm_v4(0, 0, 0, 0);
        lo = hm_v3_min(lo, v->position);
        hi = hm_v3_max(hi, v->position);
    }
    mesh->center = hm_v3_div(hm_v3_add(lo, hi), 2);
    free(pos.data); free(uv.data); free(nrm.data); free(tan.data);
    free(jnt.data); free(wgt.data); free(idx.data);
    return mesh;
}

void rzh_mesh_free(rzh_mesh *mesh) {
    if (!mesh) return;
    mesh_invalidate(mesh->vertices);
    free(mesh->vertices);
    free(mesh);
}

/* =========================== skeletons ================================== */

typedef struct {
    int count;
    int ascending;              /* key times strictly increase: the segment can be found by bisection */
    float *times;
    float *values;              /* count * width floats */
} track_t;

typedef struct {
    int parent;
    rz_mat4 inverse_bind;
    track_t translation, rotation, scale;   /* widths 3, 4, 3 */
    rz_mat4 world;                          /* this frame's joint-to-model */
} joint_t;

struct rzh_skeleton {
    float min_time, max_time;
    int num_joints;
    joint_t *joints;
    rz_mat4 *joint_matrices;
    rz_mat3 *normal_matrices;
    float cached_time;
    /* the same key tracks handed to the pipeline for evaluation on the device (rzh_skeleton_pose_source) */
    rz_pose_source *pose_source;
    int pose_source_asked;
};

static void read_track(FILE *f, const char *label, int width, track_t *t, const char *filename) {
    char fmt[64];
    int i;
    snprintf(fmt, sizeof fmt, " %s %%d:", label);
    if (fscanf(f, fmt, &t->count) != 1 || t->count < 0) die("bad ani track", filename);
    t->times = NULL; t->values = NULL;
    t->ascending = 0;
    if (t->count == 0) return;
    t->times = (float*)xmalloc(sizeof(float) * t->count);
    t->values = (float*)xmalloc(sizeof(float) * t->count * width);
    for (i = 0; i < t->count; i++) {
        float *v = t->values + (size_t)i * width;
        int got;
        if (width == 3) got = fscanf(f, " time: %f, value: [%f, %f, %f]", &t->times[i], &v[0], &v[1], &v[2]);
        else got = fscanf(f, " time: %f, value: [%f, %f, %f, %f]", &t->times[i], &v[0], &v[1], &v[2], &v[3]);
        if (got != width + 1) die("bad ani key", filename);
    }
    t->ascending = 1;
    for (i = 1; i < t->count; i++) if (!(t->times[i - 1] < t->times[i])) t->ascending = 0;
}

/* skeleton.c:142-200 */
rzh_skeleton *rzh_skeleton_load(const char *filename) {
    rzh_skeleton *sk;
    FILE *f;
    int i, r;
    if (strcmp(extension_of(filename), "ani") != 0) die("unknown skeleton type", filename);
    f = rzh_open_asset(filename);
    sk = (rzh_skeleton*)xmalloc(sizeof *sk);
    if (fscanf(f, " joint-size: %d", &sk->num_joints) != 1 || sk->num_joints <= 0) die("bad ani", filename);
    if (fscanf(f, " time-range: [%f, %f]", &sk->min_time, &sk->max_time) != 2) die("bad ani", filename);
    sk->joints = (joint_t*)xmalloc(sizeof(joint_t) * sk->num_joints);
    for (i = 0; i < sk->num_joints; i++) {
        joint_t *j = &sk->joints[i];
        char word[RZ_LINE];
        int index;
        if (fscanf(f, " joint %d:", &index) != 1 || index != i) die("bad ani joint", filename);
        if (fscanf(f, " parent-index: %d", &j->parent) != 1) die("bad ani parent", filename);
        if (fscanf(f, " %255s", word) != 1 || strcmp(word, "inverse-bind:") != 0) die("bad ani bind", filename);
        for (r = 0; r < 4; r++) {
            if (fscanf(f, " %f %f %f %f", &j->inverse_bind.m[r][0], &j->inverse_bind.m[r][1],
                       &j->inverse_bind.m[r][2], &j->inverse_bind.m[r][3]) != 4) die("bad ani bind row", filename);
        }
        read_track(f, "translations", 3, &j->translation, filename);
        read_track(f, "rotations", 4, &j->rotation, filename);
        read_track(f, "scales", 3, &j->scale, filename);
    }
    fclose(f);
    sk->joint_matrices = (rz_mat4*)xmalloc(sizeof(rz_mat4) * sk->num_joints);
    sk->normal_matrices = (rz_mat3*)xmalloc(sizeof(rz_mat3) * sk->num_joints);
    memset(sk->joint_matrices, 0, sizeof(rz_mat4) * sk->num_joints);
    memset(sk->normal_matrices, 0, sizeof(rz_mat3) * sk->num_joints);
    sk->cached_time = -1;
    sk->pose_source = NULL;
    sk->pose_source_asked = 0;
    return sk;
}

/* The skeleton as the flat description pose_source_create takes; created on first use, NULL when the
 * pipeline does not take this skeleton (the caller then poses it here, rzh_skeleton_update). */
rz_pose_source *rzh_skeleton_pose_source(rzh_skeleton *sk) {
    rz_skeleton_desc d;
    int *parent, *counts, i, k;
    rz_mat4 *bind;
    const float **times, **values;
    if (sk->pose_source_asked) return sk->pose_source;
    sk->pose_source_asked = 1;
    parent = (int*)xmalloc(sizeof(int) * sk->num_joints);
    bind = (rz_mat4*)xmalloc(sizeof(rz_mat4) * sk->num_joints);
    counts = (int*)xmalloc(sizeof(int) * 3 * sk->num_joints);
    times = (const float**)xmalloc(sizeof(float*) * 3 * sk->num_joints);
    values = (const float**)xmalloc(sizeof(float*) * 3 * sk->num_joints);
    for (i = 0; i < sk->num_joints; i++) {
        const track_t *tracks[3];
        tracks[0] = &sk->joints[i].translation; tracks[1] = &sk->joints[i].rotation; tracks[2] = &sk->joints[i].scale;
        parent[i] = sk->joints[i].parent;
        bind[i] = sk->joints[i].inverse_bind;
        for (k = 0; k < 3; k++) {
            counts[3 * i + k] = tracks[k]->count;
            times[3 * i + k] = tracks[k]->times;
            values[3 * i + k] = tracks[k]->values;
        }
    }
    d.num_joints = sk->num_joints;
    d.max_time = sk->max_time;
    d.parent = parent; d.inverse_bind = bind; d.key_count = counts; d.key_times = times; d.key_values = values;
    sk->pose_source = pose_source_create(&d);
    free(parent); free(bind); free(counts); free((void*)times); free((void*)values);
    return sk->pose_source;
}

void rzh_skeleton_free(rzh_skeleton *sk) {
    int i;
    if (!sk) return;
    if (sk->pose_source) pose_source_release(sk->pose_source);
    for (i = 0; i < sk->num_joints; i++) {
        joint_t *j = &sk->joints[i];
        free(j->translation.times); free(j->translation.values);
        free(j->rotation.times); free(j->rotation.values);
        free(j->scale.times); free(j->scale.values);
    }
    free(sk->joints); free(sk->joint_matrices); free(sk->normal_matrices);
    free(sk);
}

/* skeleton.c:231-313: clamp outside the key range, else the first segment
 * [t_i, t_i+1) containing the time, parameter (t - t_i) / (t_i+1 - t_i).
 * Returns the segment index (or -1 / -2 for the clamped ends) and the
 * parameter; vector tracks lerp, rotations slerp. */
static int track_segment(const track_t *t, float time, float *param) {
    int i;
    if (time <= t->times[0]) return -1;
    if (time >= t->times[t->count - 1]) return -2;
    if (t->ascending) {
        /* the one segment with times[i] <= time < times[i + 1] (which is what the scan below finds
         * first): animations have hundreds of keys per track and an orbit evaluates them per view */
        int lo = 0, hi = t->count - 1;              /* times[lo] <= time < times[hi] */
        while (hi - lo > 1) {
            int mid = (lo + hi) / 2;
            if (time >= t->times[mid]) lo = mid; else hi = mid;
        }
        *param = (time - t->times[lo]) / (t->times[lo + 1] - t->times[lo]);
        return lo;
    }
    for (i = 0; i < t->count - 1; i++) {
        float a = t->times[i], b = t->times[i + 1];
        if (time >= a && time < b) { *param = (time - a) / (b - a); return i; }
    }
    die("skeleton key search failed", NULL);
    return -1;
}

static rz_vec3 sample_vec3(const track_t *t, float time, rz_vec3 fallback) {
    float u = 0;
    int s;
    const float *v;
    if (t->count == 0) return fallback;
    s = track_segment(t, time, &u);
    if (s == -1) { v = t->values; return hm_v3(v[0], v[1], v[2]); }
    if (s == -2) { v = t->values + (size_t)(t->count - 1) * 3; return hm_v3(v[0], v[1], v[2]); }
    v = t->values + (size_t)s * 3;
    return hm_v3_lerp(hm_v3(v[0], v[1], v[2]), hm_v3(v[3], v[4], v[5]), u);
}

static rz_vec4 sample_quat(const track_t *t, float time) {
    float u = 0;
    int s;
    const float *v;
    if (t->count == 0) return hm_v4(0, 0, 0, 1);
    s = track_segment(t, time, &u);
    if (s == -1) { v = t->values; return hm_v4(v[0], v[1], v[2], v[3]); }
    if (s == -2) { v = t->values + (size_t)(t->count - 1) * 4; return hm_v4(v[0], v[1], v[2], v[3]); }
    v = t->values + (size_t)s * 4;
    return hm_quat_slerp(hm_v4(v[0], v[1], v[2], v[3]), hm_v4(v[4], v[5], v[6], v[7]), u);
}

/* skeleton.c:315-341: time wraps with fmod(max_time); joints are stored
 * parents-first; joint matrix = world * inverse_bind, normal matrix = its
 * upper 3x3 inverse-transposed.  Results are cached per time value. */
void rzh_skeleton_update(rzh_skeleton *sk, float frame_time) {
    int i;
    frame_time = (float)fmod(frame_time, sk->max_time);
    if (frame_time == sk->cached_time) return;
    for (i = 0; i < sk->num_joints; i++) {
        joint_t *j = &sk->joints[i];
        rz_vec3 t = sample_vec3(&j->translation, frame_time, hm_v3(0, 0, 0));
        rz_vec4 r = sample_quat(&j->rotation, frame_time);
        rz_vec3 s = sample_vec3(&j->scale, frame_time, hm_v3(1, 1, 1));
        rz_mat4 jm;
        j->world = hm_m4_from_trs(t, r, s);
        if (j->parent >= 0) j->world = hm_m4_mul(sk->joints[j->parent].world, j->world);
        jm = hm_m4_mul(j->world, j->inverse_bind);
        sk->joint_matrices[i] = jm;
        sk->normal_matrices[i] = hm_m3_inverse_transpose(hm_m3_from_m4(jm));
    }
    sk->cached_time = frame_time;
}

void rzh_skeleton_use_bisection(rzh_skeleton *sk, int on) {
    int i, k;
    for (i = 0; i < sk->num_joints; i++) {
        track_t *tracks[3];
        tracks[0] = &sk->joints[i].translation; tracks[1] = &sk->joints[i].rotation; tracks[2] = &sk->joints[i].scale;
        for (k = 0; k < 3; k++) {
            track_t *t = tracks[k];
            int j;
            t->ascending = on && t->count > 0;
            for (j = 1; j < t->count; j++) if (!(t->times[j - 1] < t->times[j])) t->ascending = 0;
        }
    }
    sk->cached_time = -1;
}

rz_mat4 *rzh_skeleton_joint_matrices(rzh_skeleton *sk) { return sk->joint_matrices; }
rz_mat3 *rzh_skeleton_normal_matrices(rzh_skeleton *sk) { return sk->normal_matrices; }
int rzh_skeleton_num_joints(rzh_skeleton *sk) { return sk->num_joints; }

/* =========================== caches ===================================== */

enum { KIND_MESH, KIND_SKELETON, KIND_TEXTURE, KIND_SKYBOX, KIND_IBL };

typedef struct {
    int kind, param, refs;
    char *key;
    void *object;
} slot_t;

static slot_t *g_slots;
static int g_num_slots, g_cap_slots;

static slot_t *cache_find(int kind, const char *key, int param) {
    int i;
    for (i = 0; i < g_num_slots; i++) {
        slot_t *s = &g_slots[i];
        if (s->refs > 0 && s->kind == kind && s->param == param && strcmp(s->key, key) == 0) return s;
    }
    return NULL;
}

static slot_t *cache_find_object(int kind, void *object) {
    int i;
    for (i = 0; i < g_num_slots; i++) {
        if (g_slots[i].refs > 0 && g_slots[i].kind == kind && g_slots[i].object == object) return &g_slots[i];
    }
    return NULL;
}

static slot_t *cache_add(int kind, const char *key, int param, void *object) {
    slot_t *s = NULL;
    int i;
    for (i = 0; i < g_num_slots; i++) if (g_slots[i].refs == 0) { s = &g_slots[i]; break; }
    if (!s) {
        if (g_num_slots == g_cap_slots) {
            g_cap_slots = g_cap_slots ? g_cap_slots * 2 : 64;
            g_slots = (slot_t*)realloc(g_slots, sizeof(slot_t) * g_cap_slots);
            if (!g_slots) die("out of memory", NULL);
        }
        s = &g_slots[g_num_slots++];
        s->key = NULL;
    }
    free(s->key);
    s->key = (char*)xmalloc(strlen(key) + 1);
    strcpy(s->key, key);
    s->kind = kind; s->param = param; s->refs = 1; s->object = object;
    return s;
}

rzh_mesh *rzh_acquire_mesh(const char *filename) {
    slot_t *s;
    if (!filename) return NULL;
    s = cache_find(KIND_MESH, filename, 0);
    if (s) { s->refs++; return (rzh_mesh*)s->object; }
    return (rzh_mesh*)cache_add(KIND_MESH, filename, 0, rzh_mesh_load(filename))->object;
}

void rzh_release_mesh(rzh_mesh *mesh) {
    slot_t *s;
    if (!mesh) return;
    s = cache_find_object(KIND_MESH, mesh);
    if (!s) die("release of unknown mesh", NULL);
    if (--s->refs == 0) rzh_mesh_free(mesh);
}

rzh_skeleton *rzh_acquire_skeleton(const char *filename) {
    slot_t *s;
    if (!filename) return NULL;
    s = cache_find(KIND_SKELETON, filename, 0);
    if (s) { s->refs++; return (rzh_skeleton*)s->object; }
    return (rzh_skeleton*)cache_add(KIND_SKELETON, filename, 0, rzh_skeleton_load(filename))->object;
}

void rzh_release_skeleton(rzh_skeleton *sk) {
    slot_t *s;
    if (!sk) return;
    s = cache_find_object(KIND_SKELETON, sk);
    if (!s) die("release of unknown skeleton", NULL);
    if (--s->refs == 0) rzh_skeleton_free(sk);
}

/* cache_helper.c:144: keyed by (file, usage) */
rz_texture *rzh_acquire_texture(const char *filename, rz_usage usage) {
    slot_t *s;
    if (!filename) return NULL;
    s = cache_find(KIND_TEXTURE, filename, (int)usage);
    if (s) { s->refs++; return (rz_texture*)s->object; }
    return (rz_texture*)cache_add(KIND_TEXTURE, filename, (int)usage, texture_from_file(filename, usage))->object;
}

void rzh_release_texture(rz_texture *tex) {
    slot_t *s;
    if (!tex) return;
    s = cache_find_object(KIND_TEXTURE, tex);
    if (!s) die("release of unknown texture", NULL);
    if (--s->refs == 0) texture_release(tex);
}

static rz_cubemap *load_cube(const char *env, const char *prefix, rz_usage usage) {
    static const char *face[6] = {"px", "nx", "py", "ny", "pz", "nz"};
    char path[6][RZ_LINE];
    int i;
    for (i = 0; i < 6; i++) snprintf(path[i], RZ_LINE, "%s/%s_%s.hdr", env, prefix, face[i]);
    return cubemap_from_files(path[0], path[1], path[2], path[3], path[4], path[5], usage);
}

/* cache_helper.c:208-231: the skybox shows the irradiance map (-1), the
 * sharpest (0) or the second (1) prefiltered map, tone-mapped at load */
rz_cubemap *rzh_acquire_skybox(const char *env_name, int blur_level) {
    slot_t *s;
    const char *prefix;
    if (!env_name) return NULL;
    s = cache_find(KIND_SKYBOX, env_name, blur_level);
    if (s) { s->refs++; return (rz_cubemap*)s->object; }
    prefix = blur_level == -1 ? "i" : (blur_level == 1 ? "m1" : "m0");
    return (rz_cubemap*)cache_add(KIND_SKYBOX, env_name, blur_level,
                                  load_cube(env_name, prefix, RZ_USAGE_LDR_COLOR))->object;
}

void rzh_release_skybox(rz_cubemap *cube) {
    slot_t *s;
    if (!cube) return;
    s = cache_find_object(KIND_SKYBOX, cube);
    if (!s) die("release of unknown skybox", NULL);
    if (--s->refs == 0) cubemap_release(cube);
}

/* cache_helper.c:299-339: every shipped environment has 10 prefiltered
 * levels m0..m9, an irradiance map i_*, and shares common/brdf_lut.hdr */
rz_ibldata *rzh_acquire_ibldata(const char *env_name) {
    slot_t *s;
    rz_ibldata *ibl;
    int i;
    if (!env_name) return NULL;
    s = cache_find(KIND_IBL, env_name, 0);
    if (s) { s->refs++; return (rz_ibldata*)s->object; }
    ibl = (rz_ibldata*)xmalloc(sizeof *ibl);
    memset(ibl, 0, sizeof *ibl);
    ibl->mip_levels = 10;
    ibl->diffuse_map = load_cube(env_name, "i", RZ_USAGE_HDR_COLOR);
    for (i = 0; i < ibl->mip_levels; i++) {
        char prefix[8];
        snprintf(prefix, sizeof prefix, "m%d", i);
        ibl->specular_maps[i] = load_cube(env_name, prefix, RZ_USAGE_HDR_COLOR);
    }
    ibl->brdf_lut = rzh_acquire_texture("common/brdf_lut.hdr", RZ_USAGE_HDR_DATA);
    return (rz_ibldata*)cache_add(KIND_IBL, env_name, 0, ibl)->object;
}

void rzh_release_ibldata(rz_ibldata *ibl) {
    slot_t *s;
    int i;
    if (!ibl) return;
    s = cache_find_object(KIND_IBL, ibl);
    if (!s) die("release of unknown ibldata", NULL);
    if (--s->refs > 0) return;
    cubemap_release(ibl->diffuse_map);
    for (i = 0; i < ibl->mip_levels; i++) cubemap_release(ibl->specular_maps[i]);
    rzh_release_texture(ibl->brdf_lut);
    free(ibl);
}
