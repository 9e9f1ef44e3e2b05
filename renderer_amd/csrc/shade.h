/*
 * Device maths, texture sampling and the fragment shaders of the three shader
 * families (gfx950).  Every expression keeps the association order of the
 * reference routine it replaces (cited per function) and the translation unit
 * is compiled with -ffp-contract=off, correctly rounded f32 divide/sqrt and
 * denormals on, so everything except the double-precision pow() calls is
 * bit-identical to the strict CPU path.
 *
 * A fragment shader here runs for lanes of one wave whose fragments belong to
 * ONE draw (whatever triangle each lane's fragment comes from): the draw's
 * uniforms are wave-uniform (scalar loads), varyings are per lane.
 */
#ifndef RZ_SHADE_H
#define RZ_SHADE_H

#include <hip/hip_runtime.h>
#include "rz_device.h"

#define RZ_EPS 1e-5f                /* macro.h:4 */
#define RZ_PI_F 3.1415927f          /* macro.h:5 */

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

#define RZ_DEV static __device__ __forceinline__

RZ_DEV float d_max(float a, float b) { return a > b ? a : b; }                 /* maths.c:13 */
RZ_DEV float d_min(float a, float b) { return a < b ? a : b; }                 /* maths.c:9 */
RZ_DEV float d_mix(float a, float b, float t) { return a + (b - a) * t; }      /* maths.c:17 */
RZ_DEV float d_sat(float f) { return f < 0 ? 0 : (f > 1 ? 1 : f); }            /* maths.c:25 */

/* ---- pow ------------------------------------------------------------------
 * The reference calls the C library's double pow and narrows the result
 * (maths.c:37-43, pbr_shader.c:357, blinn_shader.c:207).  The device library's
 * general double pow costs ~600 instructions per call; here the base is a
 * positive float, so log2 by the atanh series and exp2 by a Taylor polynomial,
 * both in double with explicit fma, give a relative error below 1e-14 --
 * thirty bits beyond float, which makes a different float after narrowing a
 * once-in-10^7 event -- in about a tenth of the instructions. */
RZ_DEV double d_log2_pos(double x) {
    int e;
    double m = frexp(x, &e);                        /* [0.5, 1) */
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    double s = (m - 1.0) / (m + 1.0);               /* |s| <= 0.1716 */
    double z = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    /* log2(m) = 2 atanh(s) / ln 2 */
    return fma(s * p, 2.8853900817779268147, (double)e);
}
RZ_DEV double d_exp2(double y) {
    double k = rint(y);
    double t = (y - k) * 0.69314718055994530942;    /* |t| <= 0.3466 */
    double p = 1.0 / 6227020800.0;                  /* 1/13! */
    p = fma(p, t, 1.0 / 479001600.0);
    p = fma(p, t, 1.0 / 39916800.0);
    p = fma(p, t, 1.0 / 3628800.0);
    p = fma(p, t, 1.0 / 362880.0);
    p = fma(p, t, 1.0 / 40320.0);
    p = fma(p, t, 1.0 / 5040.0);
    p = fma(p, t, 1.0 / 720.0);
    p = fma(p, t, 1.0 / 120.0);
    p = fma(p, t, 1.0 / 24.0);
    p = fma(p, t, 1.0 / 6.0);
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    return ldexp(p, (int)k);
}
/* pow(base, exponent) narrowed to float, for base >= 0 (or NaN): the only
 * cases the shaders produce (a saturated colour, a positive cosine) */
RZ_DEV float d_powf(float base, double exponent) {
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    if (base > 0.0f) {
        if (base == inf) return exponent > 0 ? inf : (exponent == 0 ? 1.0f : 0.0f);
        double y = exponent * d_log2_pos((double)base);
        if (!(y > -1070.0)) return y != y ? nan : 0.0f;        /* underflows float long before */
        if (y > 1020.0) return inf;
        return (float)d_exp2(y);
    }
    if (base == 0.0f) return exponent > 0 ? 0.0f : (exponent == 0 ? 1.0f : inf);
    return nan;                                                 /* NaN in, or a negative base */
}
/* one copy of the general routine for the rare inputs outside the table's range */
static __device__ __noinline__ float d_powf_outlined(float base, double exponent) { return d_powf(base, exponent); }

/* float_linear2srgb (maths.c:41): (float)pow(v, 1 / 2.2), the one pow on the pbr path (three per
 * fragment).  For 0 < v <= 1 split v = 2^e * m0 * (1 + r) with m0 = 1 + i/32 the mantissa's top five
 * bits: v^p = (2^e)^p * m0^p * (1 + r)^p, the first two from tables filled by the host's own pow
 * (150 + 32 doubles), the last by its binomial series (r < 1/32: the x^10 term is below 1e-17).
 * Three roundings at 1.1e-16 each: the same accuracy class as d_powf at a twentieth of the work. */
struct SrgbTables {
    double pow2[150];       /* (2^e)^(1/2.2), e = -149 .. 0 */
    double m0[32];          /* (1 + i/32)^(1/2.2) */
    double inv[32];         /* 1 / (1 + i/32) */
    float step[256];        /* step[k], k >= 1: the smallest v in (0, 1] whose byte, (unsigned char)((float)pow(v, 1 / 2.2) * 255)
                             * (maths.c:41 with maths.c:33), is at least k; found by the host with its own pow */
};
__device__ SrgbTables rz_srgb_tables;
/* (float)pow(k / 255.0f, 2.2) for every byte k (texture.c:32-62 with maths.c:37-39), filled by the host */
__device__ float rz_srgb8_table[256];

/* Where a shader reads the small tables from.  A lookup is a memory round trip on a fragment's
 * dependency chain (texel -> table -> colour, colour -> table -> byte): from global memory it costs
 * what a texel fetch costs, from LDS a tenth.  A kernel that runs long enough to pay for the copy
 * (the persistent shade kernels) copies the tables into LDS once per workgroup and shades with
 * LdsTables; the others read the global copies. */
struct GlobalTables {
    RZ_DEV float srgb8(uint32_t k) { return rz_srgb8_table[k]; }
    RZ_DEV double pow2(int i) { return rz_srgb_tables.pow2[i]; }
    RZ_DEV double m0(int i) { return rz_srgb_tables.m0[i]; }
    RZ_DEV double inv(int i) { return rz_srgb_tables.inv[i]; }
    RZ_DEV float step(int k) { return rz_srgb_tables.step[k]; }
    static constexpr bool bytes_only = false;   /* a blended fragment needs the colour itself */
};
__shared__ SrgbTables rz_lds_srgb_tables;
__shared__ float rz_lds_srgb8_table[256];
struct LdsTables {
    RZ_DEV float srgb8(uint32_t k) { return rz_lds_srgb8_table[k]; }
    RZ_DEV double pow2(int i) { return rz_lds_srgb_tables.pow2[i]; }
    RZ_DEV double m0(int i) { return rz_lds_srgb_tables.m0[i]; }
    RZ_DEV double inv(int i) { return rz_lds_srgb_tables.inv[i]; }
    RZ_DEV float step(int k) { return rz_lds_srgb_tables.step[k]; }
    static constexpr bool bytes_only = false;
    /* by every thread of the workgroup, before the first lookup */
    RZ_DEV void fill() {
        const uint32_t *src = (const uint32_t*)&rz_srgb_tables;
        uint32_t *dst = (uint32_t*)&rz_lds_srgb_tables;
        for (uint32_t i = threadIdx.x; i < sizeof(SrgbTables) / 4; i += blockDim.x) dst[i] = src[i];
        for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) rz_lds_srgb8_table[i] = rz_srgb8_table[i];
        __syncthreads();
    }
};
/* ... for a kernel all of whose fragments go to the framebuffer's bytes at once, unblended (d_srgb_for_bytes) */
struct LdsTablesBytes : LdsTables {
    static constexpr bool bytes_only = true;
};

template <class TABLES = GlobalTables>
RZ_DEV float d_srgb(float v) {
    if (v == 0.0f) return 0.0f;
    if (!(v > 0.0f && v <= 1.0f)) return d_powf_outlined(v, 1 / 2.2);       /* NaN, out of the display range */
    constexpr double p = 1 / 2.2;
    constexpr double c1 = p, c2 = c1 * (p - 1) / 2, c3 = c2 * (p - 2) / 3, c4 = c3 * (p - 3) / 4, c5 = c4 * (p - 4) / 5,
                     c6 = c5 * (p - 5) / 6, c7 = c6 * (p - 6) / 7, c8 = c7 * (p - 7) / 8, c9 = c8 * (p - 8) / 9;
    const unsigned long long bits = (unsigned long long)__double_as_longlong((double)v);
    const int e = (int)(bits >> 52) - 1023;                     /* -149 .. 0 (the sign bit is clear) */
    const unsigned long long mant = bits & 0x000fffffffffffffull;
    const int i = (int)(mant >> 47);
    const double m = __longlong_as_double((long long)(0x3ff0000000000000ull | mant));
    const double m0 = __longlong_as_double((long long)(0x3ff0000000000000ull | ((unsigned long long)i << 47)));
    const double r = (m - m0) * TABLES::inv(i);
    double q = c9;
    q = fma(q, r, c8); q = fma(q, r, c7); q = fma(q, r, c6); q = fma(q, r, c5);
    q = fma(q, r, c4); q = fma(q, r, c3); q = fma(q, r, c2); q = fma(q, r, c1);
    q = fma(q, r, 1.0);
    return (float)((TABLES::pow2(e + 149) * TABLES::m0(i)) * q);
}
/* The same for a fragment whose colour goes straight to the framebuffer's bytes (no blending): all that
 * counts is which of 256 bytes (float)pow(v, 1 / 2.2) turns into, and that is a step function of v -- the
 * transfer, the conversion to float, the scaling and the truncation are all monotone -- whose 255 steps the
 * host finds with its own pow.  A fast power gives the byte within one, two table values settle it, and
 * what is returned is the middle of that byte's interval: no double arithmetic at all. */
template <class TABLES = GlobalTables>
RZ_DEV float d_srgb_for_bytes(float v) {
    if (!(v > 0.0f && v <= 1.0f)) return d_srgb<TABLES>(v);                 /* 0, NaN, out of the display range */
    int k = (int)(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (float)(1 / 2.2)) * 255.0f);     /* v_log_f32, v_exp_f32: base 2 */
    k = max(0, min(k, 255));
    if (k < 255 && v >= TABLES::step(k + 1)) k++;
    else if (k > 0 && v < TABLES::step(k)) k--;
    return ((float)k + 0.5f) * (1.0f / 255.0f);
}
RZ_DEV float d_aces(float v) {                                                 /* maths.c:49 */
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    v = (v * (a * v + b)) / (v * (c * v + d) + e);
    return d_sat(v);
}
RZ_DEV f3 mk3(float x, float y, float z) { f3 v; v.x = x; v.y = y; v.z = z; return v; }
RZ_DEV f4 mk4(float x, float y, float z, float w) { f4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
RZ_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RZ_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RZ_DEV f3 scale3(f3 v, float f) { return mk3(v.x * f, v.y * f, v.z * f); }
RZ_DEV f3 div3(f3 v, float d) { return scale3(v, 1 / d); }                     /* maths.c:157 */
RZ_DEV f3 neg3(f3 v) { return mk3(-v.x, -v.y, -v.z); }
RZ_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RZ_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    /* maths.c:173 */
RZ_DEV f3 unit3(f3 v) { return div3(v, sqrtf(dot3(v, v))); }                   /* maths.c:165-171 */
RZ_DEV f3 cross3(f3 a, f3 b) {                                                 /* maths.c:177 */
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RZ_DEV f3 mix3(f3 a, f3 b, float t) { return mk3(d_mix(a.x, b.x, t), d_mix(a.y, b.y, t), d_mix(a.z, b.z, t)); }
RZ_DEV f3 xyz(f4 v) { return mk3(v.x, v.y, v.z); }

/* row-major 4x4 * column vector, ((a + b) + c) + d per row (maths.c:551) */
RZ_DEV f4 m4v4(const float *m, f4 v) {
    f4 o;
    o.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w;
    o.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * v.w;
    o.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * v.w;
    o.w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15] * v.w;
    return o;
}
RZ_DEV f3 m3v3(const float *m, f3 v) {                                          /* maths.c:387 */
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z,
               m[3] * v.x + m[4] * v.y + m[5] * v.z,
               m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

/* ---- texture addressing -------------------------------------------------- */

/* Images whose sides are multiples of 8 are stored as 8x8-texel tiles (one
 * 128-byte line per tile row of float4 texels, 1 KiB per tile), tiles in
 * row-major order: the 2-D locality of nearest-texel gathers becomes address
 * locality for a handful of integer operations.  Other sizes stay row-major. */
RZ_DEV uint32_t texel_index(int w, int flags, int c, int r) {
    if (flags & RZ_TEX_TILED) {
        uint32_t tile = __umul24((uint32_t)r >> 3, (uint32_t)w >> 3) + ((uint32_t)c >> 3);      /* sides stay below 2^16: a full-rate multiply */
        return (tile << 6) | (((uint32_t)r & 7u) << 3) | ((uint32_t)c & 7u);
    }
    return (uint32_t)r * (uint32_t)w + (uint32_t)c;
}

/* (float)k / 255.0f for a byte k without a division: one Newton step on k * (1/255) is the
 * correctly rounded quotient for all 256 inputs (checked against the division when the library
 * starts and in the tests) */
RZ_DEV float d_unorm8(uint32_t k) {
    const float r = 1.0f / 255.0f, f = (float)k;
    float q = f * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, f), r, q);
}

/* A sample is taken in two steps so that a shader can put all the loads of
 * one dependency level in flight before it waits for the first: load_*()
 * computes the address and issues ONE 16-byte load in straight-line code
 * (an absent map reads a dummy texel, a 4-byte format reads the aligned 16
 * bytes around its texel), decode_texel() turns the bits into the float RGBA
 * the reference's texture holds. */
/* NARROW = the caller knows that every image it samples has a 4-byte format (or is absent): a
 * pending texel is then one register instead of five and the load moves 4 bytes instead of 16. */
template <bool NARROW> struct TexelT;
template <> struct TexelT<false> { float4 q; uint32_t sel; };
template <> struct TexelT<true> { uint32_t word; };
typedef TexelT<false> Texel;

__device__ float4 rz_zero_texel;        /* what an absent map reads */

#define RZ_TEX_NARROW (RZ_TEX_R32F | RZ_TEX_UNORM8 | RZ_TEX_SRGB8 | RZ_TEX_RGBE8)

/* where a coordinate pair lands in an image of a given size and layout (texture.c:160-176) */
RZ_DEV uint32_t texel_cell(int w, int h, int flags, float u, float v) {
    int c = (int)((float)(w - 1) * u);
    int r = (int)((float)(h - 1) * v);
    c = max(0, min(c, w - 1));                          /* never needed for finite coordinates */
    r = max(0, min(r, h - 1));
    return texel_index(w, flags, c, r);
}

template <bool NARROW>
RZ_DEV TexelT<NARROW> load_cell(const void *data, int flags, uint32_t i) {
    const char *base = data ? (const char*)data : (const char*)&rz_zero_texel;
    TexelT<NARROW> x;
    if constexpr (NARROW) {
        x.word = *(const uint32_t*)(base + ((size_t)i << 2));
    } else {
        const size_t offset = (flags & RZ_TEX_NARROW) ? (((size_t)i << 2) & ~(size_t)15) : ((size_t)i << 4);
        x.q = *(const float4*)(base + offset);
        x.sel = i & 3u;
    }
    return x;
}

template <bool NARROW>
RZ_DEV TexelT<NARROW> load_texel(const DevTex &t, float u, float v) {
    const bool present = t.data != nullptr;
    const int w = present ? t.w : 1, h = present ? t.h : 1, flags = present ? t.flags : 0;
    return load_cell<NARROW>(t.data, flags, texel_cell(w, h, flags, u, v));
}

/* A material's maps usually share one size and layout, and all are sampled at the same
 * coordinates: then the texel's place is worked out once (the comparison is wave-uniform). */
struct MapCell { int w, h, tiled; uint32_t index; float u, v; };
RZ_DEV MapCell first_cell(const DevTex &t, float x, float y) {
    MapCell m;
    m.u = x - floorf(x); m.v = y - floorf(y);                   /* repeat addressing, texture.c:160-167 */
    m.w = t.data ? t.w : 1; m.h = t.data ? t.h : 1; m.tiled = t.data ? (t.flags & RZ_TEX_LAYOUT) : 0;
    m.index = texel_cell(m.w, m.h, m.tiled, m.u, m.v);
    return m;
}
template <bool NARROW>
RZ_DEV TexelT<NARROW> load_like(const DevTex &t, const MapCell &m) {
    if (!t.data) return load_cell<NARROW>(nullptr, 0, 0u);
    if (t.w == m.w && t.h == m.h && (t.flags & RZ_TEX_LAYOUT) == m.tiled) return load_cell<NARROW>(t.data, t.flags, m.index);
    return load_cell<NARROW>(t.data, t.flags, texel_cell(t.w, t.h, t.flags, m.u, m.v));
}

template <bool NARROW>
RZ_DEV TexelT<NARROW> load_repeat(const DevTex &t, float x, float y) { return load_texel<NARROW>(t, x - floorf(x), y - floorf(y)); }
template <bool NARROW>
RZ_DEV TexelT<NARROW> load_clamp(const DevTex &t, float x, float y) { return load_texel<NARROW>(t, d_sat(x), d_sat(y)); }

template <class TABLES = GlobalTables>
RZ_DEV f4 decode_word(int flags, float word) {
    if (flags & RZ_TEX_R32F) return mk4(word, word, word, 1.0f);
    const uint32_t p = __float_as_uint(word);
    const uint32_t b0 = p & 0xffu, b1 = (p >> 8) & 0xffu, b2 = (p >> 16) & 0xffu, b3 = p >> 24;
    if (flags & RZ_TEX_UNORM8) return mk4(d_unorm8(b0), d_unorm8(b1), d_unorm8(b2), d_unorm8(b3));
    if (flags & RZ_TEX_SRGB8) return mk4(TABLES::srgb8(b0), TABLES::srgb8(b1), TABLES::srgb8(b2), d_unorm8(b3));
    const float factor = ldexpf(1.0f, (int)b3 - 136);           /* image.c:329-347; a zero texel is stored as four zero bytes */
    return mk4((float)b0 * factor, (float)b1 * factor, (float)b2 * factor, 1.0f);
}

template <bool NARROW, class TABLES = GlobalTables>
RZ_DEV f4 decode_texel(const DevTex &t, const TexelT<NARROW> &x) {
    if constexpr (NARROW) {
        return decode_word<TABLES>(t.flags, __uint_as_float(x.word));
    } else {
        if (t.flags & RZ_TEX_NARROW)
            return decode_word<TABLES>(t.flags, x.sel == 0 ? x.q.x : (x.sel == 1 ? x.q.y : (x.sel == 2 ? x.q.z : x.q.w)));
        return mk4(x.q.x, x.q.y, x.q.z, x.q.w);
    }
}

/* nearest texel, index = (int)((size - 1) * coordinate) (texture.c:160-176) */
RZ_DEV f4 sample_repeat(const DevTex &t, float x, float y) { return decode_texel<false>(t, load_repeat<false>(t, x, y)); }
RZ_DEV f4 sample_clamp(const DevTex &t, float x, float y) { return decode_texel<false>(t, load_clamp<false>(t, x, y)); }

/* GL ES 2.0 3.7.5 face selection with strict comparisons (texture.c:210-255);
 * the face index differs per lane, so the six descriptors are read per lane */
RZ_DEV int cube_face(f3 d, float *s, float *t) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    /* selects only, stage by stage (z, then y over it, then x over that): every value below is one of +-d.x,
     * +-d.y, +-d.z, so nothing is rounded, and the compiler has no nest of conditions to turn into branches */
    const bool x_major = ax > ay && ax > az, y_major = ay > az;
    const bool px = d.x > 0, py = d.y > 0, pz = d.z > 0;
    float ma = az, sc = pz ? d.x : -d.x, tc = -d.y;
    int face = pz ? 4 : 5;
    if (y_major) { ma = ay; sc = d.x; tc = py ? d.z : -d.z; face = py ? 2 : 3; }
    if (x_major) { ma = ax; sc = px ? -d.z : d.z; tc = -d.y; face = px ? 0 : 1; }
    *s = (sc / ma + 1) / 2;
    *t = (tc / ma + 1) / 2;
    return face;
}
RZ_DEV f4 cube_repeat(const DevTex *faces, f3 dir) {                  /* texture.c:257 */
    float s, t;
    int face = cube_face(dir, &s, &t);
    DevTex f = faces[face];
    return sample_repeat(f, s, 1 - t);
}
RZ_DEV f4 cube_clamp(const DevTex *faces, f3 dir) {                   /* texture.c:264 */
    float s, t;
    int face = cube_face(dir, &s, &t);
    DevTex f = faces[face];
    return sample_clamp(f, s, 1 - t);
}

/* ---- shared by blinn and pbr --------------------------------------------- */

/* blinn_shader.c:178-195 == pbr_shader.c:311-328; the map's texel is loaded ahead of need */
template <bool NARROW>
RZ_DEV TexelT<NARROW> load_shadow(const DrawRec &D, f3 depth_position) {
    if (!D.shadow_map.data) {       /* uniform over the draw: no address arithmetic, no load of a texel nobody looks at */
        TexelT<NARROW> none;
        if constexpr (NARROW) none.word = 0u; else { none.q = make_float4(0, 0, 0, 0); none.sel = 0u; }
        return none;
    }
    float u = (depth_position.x + 1) * 0.5f;
    float v = (depth_position.y + 1) * 0.5f;
    return load_repeat<NARROW>(D.shadow_map, u, v);
}
template <bool NARROW>
RZ_DEV bool in_shadow(const DrawRec &D, const TexelT<NARROW> &texel, f3 depth_position, float n_dot_l) {
    if (D.shadow_map.data) {
        float d = (depth_position.z + 1) * 0.5f;
        float bias = d_max(0.05f * (1 - n_dot_l), 0.005f);
        float current = d - bias;
        float closest = decode_texel<NARROW>(D.shadow_map, texel).x;
        return current > closest;
    }
    return false;
}

/* ---- blinn: varyings = world_position[3] depth_position[3] texcoord[2] normal[3]
 *      (blinn_shader.h:16-21) --------------------------------------------- */

template <bool NARROW, class TABLES = GlobalTables>
RZ_DEV f4 fs_blinn(const DrawRec &D, const float *v, bool backface, bool *discard) {
    float tu = v[6], tv = v[7];
    if (D.flags & RZ_DRAW_SHADOW_PASS) {                        /* blinn_shader.c:101-115 */
        if (D.alpha_cutoff > 0) {
            float a = D.base[3];
            if (D.maps[0].data) a *= sample_repeat(D.maps[0], tu, tv).w;
            if (a < D.alpha_cutoff) *discard = true;
        }
        return mk4(0, 0, 0, 0);
    }
    /* every texel this fragment can need, in flight together */
    const MapCell cell = first_cell(D.maps[0].data ? D.maps[0] : D.maps[1].data ? D.maps[1] : D.maps[2], tu, tv);
    TexelT<NARROW> t0, t1, t2;
    bool bundled = false;
    if constexpr (NARROW) {
        if (D.bundle) {             /* (uniform) */
            const uint4 q = *(const uint4 RZ_GLOBAL*)(D.bundle.p + ((size_t)cell.index << 2));
            t0.word = q.x; t1.word = q.y; t2.word = q.z;
            bundled = true;
        }
    }
    if (!bundled) { t0 = load_like<NARROW>(D.maps[0], cell); t1 = load_like<NARROW>(D.maps[1], cell); t2 = load_like<NARROW>(D.maps[2], cell); }
    const TexelT<NARROW> tshadow = load_shadow<NARROW>(D, mk3(v[3], v[4], v[5]));
    /* material (:126-169) */
    f3 diffuse = mk3(D.base[0], D.base[1], D.base[2]);
    float alpha = D.base[3];
    if (D.maps[0].data) {
        f4 s = decode_texel<NARROW, TABLES>(D.maps[0], t0);
        diffuse = mul3(diffuse, xyz(s));
        alpha *= s.w;
    }
    f3 specular = mk3(0, 0, 0);
    if (D.maps[1].data) specular = xyz(decode_texel<NARROW, TABLES>(D.maps[1], t1));
    f3 normal = unit3(mk3(v[8], v[9], v[10]));
    if (backface) normal = neg3(normal);
    f3 emission = mk3(0, 0, 0);
    if (D.maps[2].data) emission = xyz(decode_texel<NARROW, TABLES>(D.maps[2], t2));

    if (D.alpha_cutoff > 0 && alpha < D.alpha_cutoff) {         /* :220-222 */
        *discard = true;
        return mk4(0, 0, 0, 0);
    }
    f3 color = emission;                                        /* :224-245 */
    if (D.ambient > 0) color = add3(color, scale3(diffuse, D.ambient));
    if (D.punctual > 0) {
        f3 light = neg3(mk3(D.light_dir[0], D.light_dir[1], D.light_dir[2]));
        float n_dot_l = dot3(normal, light);
        if (n_dot_l > 0 && !in_shadow<NARROW>(D, tshadow, mk3(v[3], v[4], v[5]), n_dot_l)) {
            f3 cam = mk3(D.camera_pos[0], D.camera_pos[1], D.camera_pos[2]);
            f3 view = unit3(sub3(cam, mk3(v[RZ_PV_WORLD], v[RZ_PV_WORLD + 1], v[RZ_PV_WORLD + 2])));
            f3 spec = mk3(0, 0, 0);
            if (!(specular.x == 0 && specular.y == 0 && specular.z == 0)) {      /* :201-212 */
                f3 half = unit3(add3(light, view));
                float n_dot_h = dot3(normal, half);
                if (n_dot_h > 0) spec = scale3(specular, d_powf(n_dot_h, (double)D.shininess));
            }
            f3 lit = add3(scale3(diffuse, n_dot_l), spec);
            color = add3(color, scale3(lit, D.punctual));
        }
    }
    return mk4(color.x, color.y, color.z, alpha);
}

/* ---- pbr: varyings = world_position[3] depth_position[3] clip_position[4]
 *      world_normal[3] world_tangent[3] world_bitangent[3] texcoord[2]
 *      (pbr_shader.h:29-37) ----------------------------------------------- */

struct Surface {
    f3 diffuse, specular;
    float alpha, roughness;
    f3 normal;
    float occlusion;
    f3 emission;
};

RZ_DEV float largest3(f3 v) { return v.x > v.y && v.x > v.z ? v.x : (v.y > v.z ? v.y : v.z); }   /* pbr_shader.c:184 */

template <bool NARROW, class TABLES = GlobalTables>
RZ_DEV Surface pbr_surface(const DrawRec &D, const float *v, bool backface) {
    Surface s;
    float tu = v[RZ_PV_TEXCOORD], tv = v[RZ_PV_TEXCOORD + 1];
    /* the six material texels in flight together */
    const MapCell cell = first_cell(D.maps[0].data ? D.maps[0] : D.maps[3], tu, tv);
    TexelT<NARROW> t0, t1, t2, t3, t4, t5;
    bool bundled = false;
    if constexpr (NARROW) {
        if (D.bundle) {             /* (uniform) */
            const uint32_t RZ_GLOBAL *at = D.bundle.p + ((size_t)cell.index << 3);
            const uint4 lo = *(const uint4 RZ_GLOBAL*)at;
            const uint2 hi = *(const uint2 RZ_GLOBAL*)(at + 4);
            t0.word = lo.x; t1.word = lo.y; t2.word = lo.z; t3.word = lo.w; t4.word = hi.x; t5.word = hi.y;
            bundled = true;
        }
    }
    if (!bundled) {
        t0 = load_like<NARROW>(D.maps[0], cell); t1 = load_like<NARROW>(D.maps[1], cell); t2 = load_like<NARROW>(D.maps[2], cell);
        t3 = load_like<NARROW>(D.maps[3], cell); t4 = load_like<NARROW>(D.maps[4], cell); t5 = load_like<NARROW>(D.maps[5], cell);
    }
    if (D.workflow == 0) {                                      /* metal-rough, :148-182 */
        f3 base = mk3(D.base[0], D.base[1], D.base[2]);
        float metal = D.metalness, rough = D.roughness;
        s.alpha = D.base[3];
        if (D.maps[0].data) {
            f4 t = decode_texel<NARROW, TABLES>(D.maps[0], t0);
            base = mul3(base, xyz(t));
            s.alpha *= t.w;
        }
        if (D.maps[1].data) metal *= decode_texel<NARROW, TABLES>(D.maps[1], t1).x;
        if (D.maps[2].data) rough *= decode_texel<NARROW, TABLES>(D.maps[2], t2).x;
        s.diffuse = scale3(base, (1 - 0.04f) * (1 - metal));
        s.specular = mix3(mk3(0.04f, 0.04f, 0.04f), base, metal);
        s.roughness = rough;
    } else {                                                    /* spec-gloss, :188-222 */
        f3 diff = mk3(D.base[0], D.base[1], D.base[2]);
        f3 spec = mk3(D.specular[0], D.specular[1], D.specular[2]);
        float gloss = D.glossiness;
        s.alpha = D.base[3];
        if (D.maps[0].data) {
            f4 t = decode_texel<NARROW, TABLES>(D.maps[0], t0);
            diff = mul3(diff, xyz(t));
            s.alpha *= t.w;
        }
        if (D.maps[1].data) spec = mul3(spec, xyz(decode_texel<NARROW, TABLES>(D.maps[1], t1)));
        if (D.maps[2].data) gloss *= decode_texel<NARROW, TABLES>(D.maps[2], t2).x;
        s.diffuse = scale3(diff, 1 - largest3(spec));
        s.specular = spec;
        s.roughness = 1 - gloss;
    }
    if (D.maps[3].data) {                                       /* normal map, :224-243 */
        f4 t = decode_texel<NARROW, TABLES>(D.maps[3], t3);
        f3 tn = mk3(t.x * 2 - 1, t.y * 2 - 1, t.z * 2 - 1);
        /* TBN has tangent, bitangent, normal as COLUMNS */
        const float *tg = v + RZ_PV_TANGENT, *bt = v + RZ_PV_BITANGENT, *nm = v + RZ_PV_NORMAL;
        f3 wn = mk3(tg[0] * tn.x + bt[0] * tn.y + nm[0] * tn.z,
                    tg[1] * tn.x + bt[1] * tn.y + nm[1] * tn.z,
                    tg[2] * tn.x + bt[2] * tn.y + nm[2] * tn.z);
        s.normal = unit3(wn);
    } else {
        s.normal = unit3(mk3(v[RZ_PV_NORMAL], v[RZ_PV_NORMAL + 1], v[RZ_PV_NORMAL + 2]));
    }
    if (backface) s.normal = neg3(s.normal);
    s.occlusion = D.maps[4].data ? decode_texel<NARROW, TABLES>(D.maps[4], t4).x : 1.0f;           /* :259-264 */
    s.emission = D.maps[5].data ? xyz(decode_texel<NARROW, TABLES>(D.maps[5], t5)) : mk3(0, 0, 0);
    return s;
}

/* image-based lighting (:281-309); face and prefiltered level differ per lane, so the two
 * face descriptors are read per lane -- together -- and then the three texels together */
template <bool NARROW>
RZ_DEV f3 pbr_ambient(const Surface &s, const DevIbl *ibl, f3 n, f3 view) {
    float n_dot_v = dot3(n, view);
    f3 reflected = sub3(scale3(n, 2 * n_dot_v), view);          /* :276-279 */
    float top = (float)(ibl->mip_levels - 1);
    int level = (int)(s.roughness * top + 0.5f);
    level = level < 0 ? 0 : (level > ibl->mip_levels - 1 ? ibl->mip_levels - 1 : level);
    float s1, t1, s2, t2;
    const int face1 = cube_face(n, &s1, &t1), face2 = cube_face(reflected, &s2, &t2);
    const DevTex d1 = ibl->diffuse[face1], d2 = ibl->specular[level][face2];
    const TexelT<NARROW> x1 = load_clamp<NARROW>(d1, s1, 1 - t1), x2 = load_clamp<NARROW>(d2, s2, 1 - t2);      /* texture.c:264 */
    const TexelT<NARROW> x3 = load_clamp<NARROW>(ibl->lut, n_dot_v, s.roughness);
    f3 diffuse_color = scale3(s.diffuse, s.occlusion);
    f3 irradiance = xyz(decode_texel<NARROW>(d1, x1));
    f3 diffuse = mul3(irradiance, diffuse_color);
    f4 lut = decode_texel<NARROW>(ibl->lut, x3);
    f3 spec_color = mk3(s.specular.x * lut.x + lut.y, s.specular.y * lut.x + lut.y, s.specular.z * lut.x + lut.y);
    f3 radiance = xyz(decode_texel<NARROW>(d2, x2));
    return add3(diffuse, mul3(radiance, spec_color));
}

/* GGX D, height-correlated Smith V, Schlick F (:334-393) */
RZ_DEV f3 pbr_light(const Surface &s, f3 light, f3 n, f3 view) {
    float n_dot_l = dot3(n, light);
    float n_dot_v = dot3(n, view);
    if (n_dot_l > 0 && n_dot_v > 0) {
        f3 half = unit3(add3(light, view));
        float n_dot_h = d_max(dot3(n, half), 0);
        float v_dot_h = d_max(dot3(view, half), 0);
        float a1 = s.roughness * s.roughness;
        float a2 = a1 * a1;
        float ndh2 = n_dot_h * n_dot_h;
        float fd = ndh2 * (a2 - 1) + 1;
        float d_term = a2 / (RZ_PI_F * fd * fd);
        float ndv2 = n_dot_v * n_dot_v, ndl2 = n_dot_l * n_dot_l;
        float gv = n_dot_l * sqrtf(ndv2 * (1 - a2) + a2);
        float gl = n_dot_v * sqrtf(ndl2 * (1 - a2) + a2);
        float v_term = 0.5f / (gv + gl);
        /* pow(x, 5) in double: x*x and (x*x)*(x*x) are exact or rounded once at 2^-53 */
        double fb = (double)(1 - v_dot_h), fb2 = fb * fb;
        float ff = (float)((fb2 * fb2) * fb);
        float f90 = d_sat(largest3(s.specular) * 50);
        f3 f = mk3(s.specular.x + (f90 - s.specular.x) * ff,
                   s.specular.y + (f90 - s.specular.y) * ff,
                   s.specular.z + (f90 - s.specular.z) * ff);
        f3 dl = div3(s.diffuse, RZ_PI_F);
        f3 sl = scale3(f, v_term * d_term);
        f3 lobe = mk3((1 - f.x) * dl.x + sl.x, (1 - f.y) * dl.y + sl.y, (1 - f.z) * dl.z + sl.z);
        return scale3(lobe, n_dot_l);
    }
    return mk3(0, 0, 0);
}

template <class TABLES = GlobalTables>
RZ_DEV f4 display_encode(f3 c, float alpha) {                    /* :401-406 */
    if constexpr (TABLES::bytes_only)
        return mk4(d_srgb_for_bytes<TABLES>(d_aces(c.x)), d_srgb_for_bytes<TABLES>(d_aces(c.y)), d_srgb_for_bytes<TABLES>(d_aces(c.z)), alpha);
    else
        return mk4(d_srgb<TABLES>(d_aces(c.x)), d_srgb<TABLES>(d_aces(c.y)), d_srgb<TABLES>(d_aces(c.z)), alpha);
}

/* layer inspector (:408-438) */
RZ_DEV bool right_of_band_edge(int edge, float px, float py) {
    float space = 0.15f;
    float offset = space * (float)edge;
    float sx = (1 - space * 0.5f) - offset, ex = (space * (5 - 0.5f)) - offset;
    return (px - sx) * (1.0f - 0.0f) - (py - 0.0f) * (ex - sx) > 0;
}
template <class TABLES = GlobalTables>
RZ_DEV f4 layer_color(int layer, const Surface &s) {
    if (layer == 1) return display_encode<TABLES>(s.diffuse, s.alpha);
    if (layer == 2) return display_encode<TABLES>(s.specular, s.alpha);
    if (layer == 3) return mk4(s.roughness, s.roughness, s.roughness, s.alpha);
    if (layer == 4) return mk4(s.occlusion, s.occlusion, s.occlusion, s.alpha);
    return mk4(s.normal.x * 0.5f + 0.5f, s.normal.y * 0.5f + 0.5f, s.normal.z * 0.5f + 0.5f, s.alpha);
}

/* does a fragment of this draw read the depth position (shadow map) or the clip position (layer bands)? */
RZ_DEV bool pbr_reads_positions(const DrawRec &D) {
    return (D.flags & RZ_DRAW_SHADOW_PASS) == 0 && (D.shadow_map.data != nullptr || D.layer_view == 0);
}

template <bool NARROW, class TABLES = GlobalTables>
RZ_DEV f4 fs_pbr(const DrawRec &D, const float *v, bool backface, bool *discard) {
    if (D.flags & RZ_DRAW_SHADOW_PASS) {                        /* pbr_shader.c:113-136 */
        if (D.alpha_cutoff > 0) {
            float a = D.base[3];    /* basecolor_factor.w or diffuse_factor.w */
            if (D.maps[0].data) a *= sample_repeat(D.maps[0], v[RZ_PV_TEXCOORD], v[RZ_PV_TEXCOORD + 1]).w;
            if (a < D.alpha_cutoff) *discard = true;
        }
        return mk4(0, 0, 0, 0);
    }
    const f3 depth_position = mk3(v[RZ_PV_DEPTH], v[RZ_PV_DEPTH + 1], v[RZ_PV_DEPTH + 2]);
    const TexelT<NARROW> tshadow = load_shadow<NARROW>(D, depth_position);
    Surface s = pbr_surface<NARROW, TABLES>(D, v, backface);                    /* :446-490 */
    float sxn = v[RZ_PV_CLIP] / v[RZ_PV_CLIP + 3] * 0.5f + 0.5f;
    float syn = v[RZ_PV_CLIP + 1] / v[RZ_PV_CLIP + 3] * 0.5f + 0.5f;
    if (D.alpha_cutoff > 0 && s.alpha < D.alpha_cutoff) {
        *discard = true;
        return mk4(0, 0, 0, 0);
    } else if (D.layer_view > 0) {
        return layer_color<TABLES>(D.layer_view, s);
    } else if (D.layer_view == 0 && !right_of_band_edge(0, sxn, syn)) {
        int edge;
        for (edge = 1; edge < 5; edge++) if (right_of_band_edge(edge, sxn, syn)) break;
        return layer_color<TABLES>(edge, s);
    } else {
        f3 cam = mk3(D.camera_pos[0], D.camera_pos[1], D.camera_pos[2]);
        f3 view = unit3(sub3(cam, mk3(v[RZ_PV_WORLD], v[RZ_PV_WORLD + 1], v[RZ_PV_WORLD + 2])));
        f3 light = neg3(mk3(D.light_dir[0], D.light_dir[1], D.light_dir[2]));
        float n_dot_l = dot3(s.normal, light);
        f3 color = s.emission;
        if (D.ambient > 0 && D.ibl) color = add3(color, scale3(pbr_ambient<NARROW>(s, D.ibl, s.normal, view), D.ambient));
        if (D.punctual > 0 && n_dot_l > 0) {
            if (!in_shadow<NARROW>(D, tshadow, depth_position, n_dot_l))
                color = add3(color, scale3(pbr_light(s, light, s.normal, view), D.punctual));
        }
        return display_encode<TABLES>(color, s.alpha);
    }
}

/* ---- skybox: varyings = direction[3] (skybox_shader.c:21-29); the six face descriptors sit in
 *      the draw's own maps[] (a skybox has no material maps), so the face a lane needs is one load
 *      away; when all faces share size and format only the image address differs per lane -------- */
RZ_DEV f4 fs_skybox(const DrawRec &D, const float *v) {
    float s, t;
    const int face = cube_face(mk3(v[0], v[1], v[2]), &s, &t);
    if (D.flags & RZ_DRAW_UNIFORM_CUBE) {
        DevTex f = D.maps[0];
        f.data = D.maps[face].data;
        return sample_repeat(f, s, 1 - t);                      /* texture.c:257 */
    }
    const DevTex f = D.maps[face];
    return sample_repeat(f, s, 1 - t);
}

#endif
