/*
 * TEST INFRASTRUCTURE -- vector maths of the CPU oracle.
 *
 * Restates the subset of renderer/core/maths.c that runs per vertex and per
 * fragment.  Each helper keeps the reference's association order (cited), so a
 * strict-IEEE build (-ffp-contract=off, no -ffast-math) reproduces the
 * reference bit for bit.  pow/sqrt/floor/fabs go through double, as C89 does.
 */
#ifndef ORACLE_OMATH_H
#define ORACLE_OMATH_H

#include <math.h>
#include "../include/mi355r.h"

#define O_EPS 1e-5f             /* macro.h:4 */
#define O_PI 3.1415927f         /* macro.h:5 */

static inline float o_max(float a, float b) { return a > b ? a : b; }                    /* maths.c:13 */
static inline float o_min(float a, float b) { return a < b ? a : b; }                    /* maths.c:9 */
static inline float o_mix(float a, float b, float t) { return a + (b - a) * t; }         /* maths.c:17 */
static inline float o_sat(float f) { return f < 0 ? 0 : (f > 1 ? 1 : f); }               /* maths.c:25 */
static inline unsigned char o_to_u8(float v) { return (unsigned char)(v * 255); }        /* maths.c:33 */
static inline float o_from_u8(unsigned char v) { return v / 255.0f; }                    /* maths.c:29 */
static inline float o_encode_srgb(float v) { return (float)pow(v, 1 / 2.2); }            /* maths.c:41 */
static inline float o_aces(float v) {                                                    /* maths.c:49 */
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    v = (v * (a * v + b)) / (v * (c * v + d) + e);
    return o_sat(v);
}

static inline rz_vec2 o_v2(float x, float y) { rz_vec2 v; v.x = x; v.y = y; return v; }
static inline rz_vec3 o_v3(float x, float y, float z) { rz_vec3 v; v.x = x; v.y = y; v.z = z; return v; }
static inline rz_vec4 o_v4(float x, float y, float z, float w) { rz_vec4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline rz_vec3 o_xyz(rz_vec4 v) { return o_v3(v.x, v.y, v.z); }
static inline rz_vec3 o_add3(rz_vec3 a, rz_vec3 b) { return o_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline rz_vec3 o_sub3(rz_vec3 a, rz_vec3 b) { return o_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline rz_vec3 o_scale3(rz_vec3 v, float f) { return o_v3(v.x * f, v.y * f, v.z * f); }
static inline rz_vec3 o_div3(rz_vec3 v, float d) { return o_scale3(v, 1 / d); }          /* maths.c:157 */
static inline rz_vec3 o_neg3(rz_vec3 v) { return o_v3(-v.x, -v.y, -v.z); }
static inline rz_vec3 o_mul3(rz_vec3 a, rz_vec3 b) { return o_v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float o_dot3(rz_vec3 a, rz_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* maths.c:173 */
static inline rz_vec3 o_unit3(rz_vec3 v) { return o_div3(v, (float)sqrt(o_dot3(v, v))); }      /* maths.c:165-171 */
static inline rz_vec3 o_cross3(rz_vec3 a, rz_vec3 b) {                                          /* maths.c:177 */
    return o_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline rz_vec3 o_mix3(rz_vec3 a, rz_vec3 b, float t) {
    return o_v3(o_mix(a.x, b.x, t), o_mix(a.y, b.y, t), o_mix(a.z, b.z, t));
}
static inline rz_vec4 o_sat4(rz_vec4 v) { return o_v4(o_sat(v.x), o_sat(v.y), o_sat(v.z), o_sat(v.w)); }

/* maths.c:551 */
static inline rz_vec4 o_m4v4(const rz_mat4 *m, rz_vec4 v) {
    float p[4];
    int r;
    for (r = 0; r < 4; r++) {
        float a = m->m[r][0] * v.x, b = m->m[r][1] * v.y, c = m->m[r][2] * v.z, d = m->m[r][3] * v.w;
        p[r] = a + b + c + d;
    }
    return o_v4(p[0], p[1], p[2], p[3]);
}
/* maths.c:387 */
static inline rz_vec3 o_m3v3(const rz_mat3 *m, rz_vec3 v) {
    float p[3];
    int r;
    for (r = 0; r < 3; r++) {
        float a = m->m[r][0] * v.x, b = m->m[r][1] * v.y, c = m->m[r][2] * v.z;
        p[r] = a + b + c;
    }
    return o_v3(p[0], p[1], p[2]);
}
/* maths.c:564 / 399: accumulator starts at zero, k ascending */
static inline rz_mat4 o_m4m4(const rz_mat4 *a, const rz_mat4 *b) {
    rz_mat4 o;
    int r, c, k;
    for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) {
        float acc = 0;
        for (k = 0; k < 4; k++) acc += a->m[r][k] * b->m[k][c];
        o.m[r][c] = acc;
    }
    return o;
}
static inline rz_mat3 o_m3m3(const rz_mat3 *a, const rz_mat3 *b) {
    rz_mat3 o;
    int r, c, k;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) {
        float acc = 0;
        for (k = 0; k < 3; k++) acc += a->m[r][k] * b->m[k][c];
        o.m[r][c] = acc;
    }
    return o;
}
/* maths.c:526 / 362: weighted sum of up to four matrices; weights <= 0 skipped */
static inline rz_mat4 o_blend4(const rz_mat4 *pool, rz_vec4 joint, rz_vec4 weight) {
    rz_mat4 o;
    float w[4];
    int idx[4], i, r, c;
    w[0] = weight.x; w[1] = weight.y; w[2] = weight.z; w[3] = weight.w;
    idx[0] = (int)joint.x; idx[1] = (int)joint.y; idx[2] = (int)joint.z; idx[3] = (int)joint.w;
    for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) o.m[r][c] = 0;
    for (i = 0; i < 4; i++) {
        if (w[i] > 0) {
            const rz_mat4 *s = &pool[idx[i]];
            for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) o.m[r][c] += w[i] * s->m[r][c];
        }
    }
    return o;
}
static inline rz_mat3 o_blend3(const rz_mat3 *pool, rz_vec4 joint, rz_vec4 weight) {
    rz_mat3 o;
    float w[4];
    int idx[4], i, r, c;
    w[0] = weight.x; w[1] = weight.y; w[2] = weight.z; w[3] = weight.w;
    idx[0] = (int)joint.x; idx[1] = (int)joint.y; idx[2] = (int)joint.z; idx[3] = (int)joint.w;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) o.m[r][c] = 0;
    for (i = 0; i < 4; i++) {
        if (w[i] > 0) {
            const rz_mat3 *s = &pool[idx[i]];
            for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) o.m[r][c] += w[i] * s->m[r][c];
        }
    }
    return o;
}
static inline rz_mat3 o_upper3(const rz_mat4 *m) {                    /* maths.c:348 */
    rz_mat3 n;
    int r, c;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) n.m[r][c] = m->m[r][c];
    return n;
}

#endif
