/*
 * Host-side vector / matrix helpers for the scene layer (camera, light,
 * skeleton, per-model uniforms).  These values become shader uniforms, so
 * every function evaluates its expression in the same association order as
 * the reference routine it stands for (renderer/core/maths.c, cited per
 * function) -- compiled without FMA contraction the results are bit-identical.
 * sqrt/sin/cos/tan/acos go through double libm and are narrowed, as in C89.
 */
#ifndef RZ_HMATH_H
#define RZ_HMATH_H

#include <math.h>
#include "../../include/mi355r.h"

#define RZ_EPSILON 1e-5f                    /* macro.h:4 */
#define RZ_PI 3.1415927f                    /* macro.h:5 */
#define RZ_RADIANS(deg) ((RZ_PI / 180) * (deg))   /* macro.h:7 */

static inline float hm_lerp(float a, float b, float t) { return a + (b - a) * t; }          /* maths.c:17 */
static inline float hm_clamp(float f, float lo, float hi) { return f < lo ? lo : (f > hi ? hi : f); }  /* maths.c:21 */
static inline float hm_saturate(float f) { return f < 0 ? 0 : (f > 1 ? 1 : f); }           /* maths.c:25 */

static inline rz_vec3 hm_v3(float x, float y, float z) { rz_vec3 v; v.x = x; v.y = y; v.z = z; return v; }
static inline rz_vec4 hm_v4(float x, float y, float z, float w) { rz_vec4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline rz_vec3 hm_v3_add(rz_vec3 a, rz_vec3 b) { return hm_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline rz_vec3 hm_v3_sub(rz_vec3 a, rz_vec3 b) { return hm_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline rz_vec3 hm_v3_scale(rz_vec3 v, float f) { return hm_v3(v.x * f, v.y * f, v.z * f); }
/* maths.c:157: division is "multiply by the reciprocal" */
static inline rz_vec3 hm_v3_div(rz_vec3 v, float d) { return hm_v3_scale(v, 1 / d); }
static inline rz_vec3 hm_v3_neg(rz_vec3 v) { return hm_v3(-v.x, -v.y, -v.z); }
static inline float hm_v3_dot(rz_vec3 a, rz_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* maths.c:173 */
static inline float hm_v3_len(rz_vec3 v) { return (float)sqrt(hm_v3_dot(v, v)); }                 /* maths.c:165 */
static inline rz_vec3 hm_v3_unit(rz_vec3 v) { return hm_v3_div(v, hm_v3_len(v)); }                /* maths.c:169 */
static inline rz_vec3 hm_v3_cross(rz_vec3 a, rz_vec3 b) {                                         /* maths.c:177 */
    return hm_v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline rz_vec3 hm_v3_lerp(rz_vec3 a, rz_vec3 b, float t) {
    return hm_v3(hm_lerp(a.x, b.x, t), hm_lerp(a.y, b.y, t), hm_lerp(a.z, b.z, t));
}
static inline rz_vec3 hm_v3_min(rz_vec3 a, rz_vec3 b) {
    return hm_v3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
}
static inline rz_vec3 hm_v3_max(rz_vec3 a, rz_vec3 b) {
    return hm_v3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
}

static inline rz_mat4 hm_m4_identity(void) {
    rz_mat4 m;
    int r, c;
    for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) m.m[r][c] = r == c ? 1.0f : 0.0f;
    return m;
}

/* maths.c:564: zero-initialised accumulator, k ascending */
static inline rz_mat4 hm_m4_mul(rz_mat4 a, rz_mat4 b) {
    rz_mat4 out;
    int r, c, k;
    for (r = 0; r < 4; r++) {
        for (c = 0; c < 4; c++) {
            float acc = 0;
            for (k = 0; k < 4; k++) acc += a.m[r][k] * b.m[k][c];
            out.m[r][c] = acc;
        }
    }
    return out;
}

/* maths.c:551: ((a + b) + c) + d */
static inline rz_vec4 hm_m4_mul_v4(rz_mat4 m, rz_vec4 v) {
    float p[4];
    int r;
    for (r = 0; r < 4; r++) {
        float a = m.m[r][0] * v.x, b = m.m[r][1] * v.y;
        float c = m.m[r][2] * v.z, d = m.m[r][3] * v.w;
        p[r] = a + b + c + d;
    }
    return hm_v4(p[0], p[1], p[2], p[3]);
}

static inline rz_mat3 hm_m3_from_m4(rz_mat4 m) {                  /* maths.c:348 */
    rz_mat3 n;
    int r, c;
    for (r = 0; r < 3; r++) for (c = 0; c < 3; c++) n.m[r][c] = m.m[r][c];
    return n;
}

/* maths.c:432-467: adjugate / determinant, cofactor signs as in the source */
static inline rz_mat3 hm_m3_inverse_transpose(rz_mat3 m) {
    rz_mat3 adj, out;
    float det, inv, a, b, c;
    int r, k;
    adj.m[0][0] = +(m.m[1][1] * m.m[2][2] - m.m[2][1] * m.m[1][2]);
    adj.m[0][1] = -(m.m[1][0] * m.m[2][2] - m.m[2][0] * m.m[1][2]);
    adj.m[0][2] = +(m.m[1][0] * m.m[2][1] - m.m[2][0] * m.m[1][1]);
    adj.m[1][0] = -(m.m[0][1] * m.m[2][2] - m.m[2][1] * m.m[0][2]);
    adj.m[1][1] = +(m.m[0][0] * m.m[2][2] - m.m[2][0] * m.m[0][2]);
    adj.m[1][2] = -(m.m[0][0] * m.m[2][1] - m.m[2][0] * m.m[0][1]);
    adj.m[2][0] = +(m.m[0][1] * m.m[1][2] - m.m[1][1] * m.m[0][2]);
    adj.m[2][1] = -(m.m[0][0] * m.m[1][2] - m.m[1][0] * m.m[0][2]);
    adj.m[2][2] = +(m.m[0][0] * m.m[1][1] - m.m[1][0] * m.m[0][1]);
    a = +m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]);
    b = -m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]);
    c = +m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
    det = a + b + c;
    inv = 1 / det;
    for (r = 0; r < 3; r++) for (k = 0; k < 3; k++) out.m[r][k] = adj.m[r][k] * inv;
    return out;
}

static inline rz_mat4 hm_m4_translate(float tx, float ty, float tz) {     /* maths.c:669 */
    rz_mat4 m = hm_m4_identity();
    m.m[0][3] = tx; m.m[1][3] = ty; m.m[2][3] = tz;
    return m;
}
static inline rz_mat4 hm_m4_scale(float sx, float sy, float sz) {         /* maths.c:687 */
    rz_mat4 m = hm_m4_identity();
    m.m[0][0] = sx; m.m[1][1] = sy; m.m[2][2] = sz;
    return m;
}
static inline rz_mat4 hm_m4_rotate_x(float angle) {                       /* maths.c:741 */
    float c = (float)cos(angle), s = (float)sin(angle);
    rz_mat4 m = hm_m4_identity();
    m.m[1][1] = c; m.m[1][2] = -s; m.m[2][1] = s; m.m[2][2] = c;
    return m;
}
static inline rz_mat4 hm_m4_rotate_y(float angle) {                       /* maths.c:762 */
    float c = (float)cos(angle), s = (float)sin(angle);
    rz_mat4 m = hm_m4_identity();
    m.m[0][0] = c; m.m[0][2] = s; m.m[2][0] = -s; m.m[2][2] = c;
    return m;
}
static inline rz_mat4 hm_m4_rotate_z(float angle) {                       /* maths.c:783 */
    float c = (float)cos(angle), s = (float)sin(angle);
    rz_mat4 m = hm_m4_identity();
    m.m[0][0] = c; m.m[0][1] = -s; m.m[1][0] = s; m.m[1][1] = c;
    return m;
}

/* maths.c:810 */
static inline rz_mat4 hm_m4_lookat(rz_vec3 eye, rz_vec3 target, rz_vec3 up) {
    rz_vec3 zax = hm_v3_unit(hm_v3_sub(eye, target));
    rz_vec3 xax = hm_v3_unit(hm_v3_cross(up, zax));
    rz_vec3 yax = hm_v3_cross(zax, xax);
    rz_mat4 m = hm_m4_identity();
    m.m[0][0] = xax.x; m.m[0][1] = xax.y; m.m[0][2] = xax.z;
    m.m[1][0] = yax.x; m.m[1][1] = yax.y; m.m[1][2] = yax.z;
    m.m[2][0] = zax.x; m.m[2][1] = zax.y; m.m[2][2] = zax.z;
    m.m[0][3] = -hm_v3_dot(xax, eye);
    m.m[1][3] = -hm_v3_dot(yax, eye);
    m.m[2][3] = -hm_v3_dot(zax, eye);
    return m;
}

/* maths.c:911 */
static inline rz_mat4 hm_m4_orthographic(float right, float top, float near_, float far_) {
    float zr = far_ - near_;
    rz_mat4 m = hm_m4_identity();
    m.m[0][0] = 1 / right;
    m.m[1][1] = 1 / top;
    m.m[2][2] = -2 / zr;
    m.m[2][3] = -(near_ + far_) / zr;
    return m;
}

/* maths.c:939 */
static inline rz_mat4 hm_m4_perspective(float fovy, float aspect, float near_, float far_) {
    float zr = far_ - near_;
    rz_mat4 m = hm_m4_identity();
    m.m[1][1] = 1 / (float)tan(fovy / 2);
    m.m[0][0] = m.m[1][1] / aspect;
    m.m[2][2] = -(near_ + far_) / zr;
    m.m[2][3] = -2 * near_ * far_ / zr;
    m.m[3][2] = -1;
    m.m[3][3] = 0;
    return m;
}

/* maths.c:492: rotation matrix of a unit quaternion (x, y, z, w) */
static inline rz_mat4 hm_m4_from_quat(rz_vec4 q) {
    rz_mat4 m = hm_m4_identity();
    float xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
    float yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w;
    float zz = q.z * q.z, zw = q.z * q.w;
    m.m[0][0] = 1 - 2 * (yy + zz); m.m[0][1] = 2 * (xy - zw);     m.m[0][2] = 2 * (xz + yw);
    m.m[1][0] = 2 * (xy + zw);     m.m[1][1] = 1 - 2 * (xx + zz); m.m[1][2] = 2 * (yz - xw);
    m.m[2][0] = 2 * (xz - yw);     m.m[2][1] = 2 * (yz + xw);     m.m[2][2] = 1 - 2 * (xx + yy);
    return m;
}

/* maths.c:519: T * (R * S) */
static inline rz_mat4 hm_m4_from_trs(rz_vec3 t, rz_vec4 r, rz_vec3 s) {
    rz_mat4 tm = hm_m4_translate(t.x, t.y, t.z);
    rz_mat4 rm = hm_m4_from_quat(r);
    rz_mat4 sm = hm_m4_scale(s.x, s.y, s.z);
    return hm_m4_mul(tm, hm_m4_mul(rm, sm));
}

/* maths.c:291 */
static inline rz_vec4 hm_quat_slerp(rz_vec4 a, rz_vec4 b, float t) {
    float cosv = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (cosv < 0) {
        b = hm_v4(-b.x, -b.y, -b.z, -b.w);
        cosv = -cosv;
    }
    if (cosv > 1 - RZ_EPSILON) {
        return hm_v4(hm_lerp(a.x, b.x, t), hm_lerp(a.y, b.y, t),
                     hm_lerp(a.z, b.z, t), hm_lerp(a.w, b.w, t));
    } else {
        float angle = (float)acos(cosv);
        float sinv = (float)sin(angle);
        float fa = (float)sin((1 - t) * angle) / sinv;
        float fb = (float)sin(t * angle) / sinv;
        return hm_v4(fa * a.x + fb * b.x, fa * a.y + fb * b.y,
                     fa * a.z + fb * b.z, fa * a.w + fb * b.w);
    }
}

#endif /* RZ_HMATH_H */
