// orc_math.h — TEST INFRASTRUCTURE (parity oracle), never part of the shipped path.
//
// Scalar restatement of the third-party arithmetic the reference's hot path runs on:
//   * glam 0.28.0 (Cargo.lock:297-298) Vec2/Vec3/Vec4/Mat4 — the crate is not vendored under
//     /root/reference, so its published formulas are restated here and anchored on the reference's
//     call sites (camera.rs:91-93,117-123,143-148; every utils/*.rs update()/intersect()).
//   * Rust std f32 -> platform libm (powf, atan): not pinned by the reference.  The oracle uses a
//     fully specified pair (orc_powf, orc_atanf) built from IEEE-754 basic operations only; the
//     non-GPU tests check them against this host's libm.  Compile with -DORC_HOST_LIBM to route
//     both to the host libm instead (used to show the two agree within the parity tolerance).
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and cannot be built
// here (no Rust toolchain); nothing in this file is checked against reference outputs.
//
// All arithmetic is f32 with no FMA contraction and no reassociation (build with
// -ffp-contract=off, no -ffast-math), matching Rust's semantics.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------------------------------------
// scalar helpers with Rust semantics
// ---------------------------------------------------------------------------------------------
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; std::memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; std::memcpy(&f, &u, 8); return f; }

// f32::max / f32::min: a NaN operand is ignored.
static inline float fmax_(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float fmin_(float a, float b) { return (a < b || b != b) ? a : b; }
// f32::clamp: NaN stays NaN.
static inline float fclamp(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
// f32::signum: +1 for +0 and positives, -1 for -0 and negatives, NaN for NaN.
static inline float fsignum(float x) { if (x != x) return x; return (f2u(x) >> 31) ? -1.0f : 1.0f; }
// `as u8` / `as u32`: truncate toward zero, saturate, NaN -> 0.
static inline uint8_t as_u8(float f) { if (!(f > 0.0f)) return 0; if (f >= 255.0f) return 255; return (uint8_t)(int)f; }
static inline uint32_t as_u32(float f) { if (!(f > 0.0f)) return 0; if (f >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)f; }

// ---------------------------------------------------------------------------------------------
// deterministic libm pair
// ---------------------------------------------------------------------------------------------
#define RT_POW_TABLE_QUAL const
#define RT_POW_TABLE_NAME(n) orc_pow_##n
#include "orc_pow_tables.inc"
#undef RT_POW_TABLE_QUAL
#undef RT_POW_TABLE_NAME

// Fused multiply-add in binary64.  The result is defined by IEEE 754 (one rounding), so hardware FMA and the
// libm fallback agree bit for bit; target_clones only picks the fast one where the CPU has it.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_FMA_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_FMA_CLONES
#endif

// powf(ax, y) for ax > 0 finite: exp2(y * log2(ax)) in binary64 (algorithm: tools/gen_pow_tables.py).
ORC_FMA_CLONES __attribute__((noinline)) static float pow_core(float ax, float y) {
    // ---- log2(ax): range reduction on the binary32 bits (subnormals are scaled up first)
    int e_adj = 0;
    if (ax < 1.17549435e-38f) { ax = ax * 16777216.0f; e_adj = -24; }
    uint32_t ix = f2u(ax);
    int e = (int)(ix >> 23) - 127 + e_adj;
    float zf = u2f((ix & 0x007FFFFFu) | 0x3F800000u);  // [1, 2)
    if (zf >= 1.4609375f) { zf = zf * 0.5f; e += 1; }   // [187/256, 187/128)
    int i = (int)(zf * 64.0f - 46.5f);                  // both operations exact
    double z = (double)zf;
    double r = __builtin_fma(z, orc_pow_logtab[2 * i], -1.0);
    double p = orc_pow_logpoly[4];
    p = __builtin_fma(p, r, orc_pow_logpoly[3]);
    p = __builtin_fma(p, r, orc_pow_logpoly[2]);
    p = __builtin_fma(p, r, orc_pow_logpoly[1]);
    p = __builtin_fma(p, r, orc_pow_logpoly[0]);
    double lg = __builtin_fma(p, r, (double)e + orc_pow_logtab[2 * i + 1]);
    // ---- exp2(y * log2(ax))
    double E = (double)y * lg;
    if (!(std::fabs(E) < 160.0)) E = (E < 0.0) ? -160.0 : 160.0;  // beyond binary32's range either way (NaN cannot occur here)
    const double SHIFT = 6755399441055744.0;  // 1.5 * 2^52: adding it rounds to an integer (round-to-nearest-even)
    double t = E * 32.0;
    double kd = (t + SHIFT) - SHIFT;
    int k = (int)kd;
    double rr = t - kd;
    double q = orc_pow_exppoly[3];
    q = __builtin_fma(q, rr, orc_pow_exppoly[2]);
    q = __builtin_fma(q, rr, orc_pow_exppoly[1]);
    q = __builtin_fma(q, rr, orc_pow_exppoly[0]);
    q = q * rr;
    double s = u2d(orc_pow_exp2bits[k & 31] + ((uint64_t)(int64_t)(k >> 5) << 52));
    return (float)__builtin_fma(s, q, s);
}

// 0: not an integer, 1: odd integer, 2: even integer (y finite, non-zero)
static inline int pow_yint(float y) {
    float ay = std::fabs(y);
    if (ay >= 16777216.0f) return 2;
    if (ay < 1.0f) return 0;
    int64_t iy = (int64_t)ay;
    if ((float)iy != ay) return 0;
    return (iy & 1) ? 1 : 2;
}

static inline float orc_powf_spec(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    const float inf = INFINITY;
    if (y == inf || y == -inf) {
        float ax = std::fabs(x);
        if (ax == 1.0f) return 1.0f;
        return ((ax > 1.0f) == (y > 0.0f)) ? inf : 0.0f;
    }
    if (x > 0.0f && x != inf) return pow_core(x, y);
    int yi = pow_yint(y);
    if (x == 0.0f) {  // +-0
        if (y < 0.0f) return (yi == 1) ? std::copysign(inf, x) : inf;
        return (yi == 1) ? x : 0.0f;
    }
    if (x == inf) return (y < 0.0f) ? 0.0f : inf;
    if (x == -inf) {
        if (y < 0.0f) return (yi == 1) ? -0.0f : 0.0f;
        return (yi == 1) ? -inf : inf;
    }
    // x < 0, finite
    if (yi == 0) return std::nanf("");
    float r = pow_core(-x, y);
    return (yi == 1) ? -r : r;
}

// atan in pure f32 arithmetic: the classic fdlibm / musl single-precision algorithm
// (argument reduction to [0, 7/16) by four break points, odd/even split polynomial).
static inline float orc_atanf_spec(float x) {
    static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                                 9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                                 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    uint32_t hx = f2u(x);
    uint32_t ix = hx & 0x7fffffffu;
    bool neg = (hx >> 31) != 0;
    int id;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;
        float z = atanhi[3] + atanlo[3];
        return neg ? -z : z;
    }
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = std::fabs(x);
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0; x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                 // 11/16 <= |x| < 19/16
                id = 1; x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {
                id = 3; x = -1.0f / x;
            }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return neg ? -r : r;
}

#ifdef ORC_HOST_LIBM
static inline float orc_powf(float x, float y) { return ::powf(x, y); }
static inline float orc_atanf(float x) { return ::atanf(x); }
#else
static inline float orc_powf(float x, float y) { return orc_powf_spec(x, y); }
static inline float orc_atanf(float x) { return orc_atanf_spec(x); }
#endif

// ---------------------------------------------------------------------------------------------
// glam 0.28 vectors (Vec3 is the scalar struct; Vec4/Mat4 are SSE2 on x86-64 but every operation
// used here is lane-wise, so the scalar statement below is value-identical).
// ---------------------------------------------------------------------------------------------
struct Vec2 { float x, y; };
struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };

static inline Vec2 v2(float x, float y) { return Vec2{x, y}; }
static inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
static inline Vec2 operator+(Vec2 a, float s) { return {a.x + s, a.y + s}; }
static inline Vec2 operator*(float s, Vec2 a) { return {s * a.x, s * a.y}; }
static inline Vec2 operator*(Vec2 a, float s) { return {a.x * s, a.y * s}; }
static inline float dot(Vec2 a, Vec2 b) { return (a.x * b.x) + (a.y * b.y); }
static inline float length(Vec2 a) { return std::sqrt(dot(a, a)); }
static inline Vec2 vmax(Vec2 a, Vec2 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y)}; }

static inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
static inline Vec3 splat3(float s) { return Vec3{s, s, s}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline Vec3 operator/(float s, Vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
static inline Vec3 operator+(Vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline Vec3 &operator+=(Vec3 &a, Vec3 b) { a = a + b; return a; }
static inline Vec3 &operator*=(Vec3 &a, float s) { a = a * s; return a; }
static inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// Vec3::normalize = self * length().recip()
static inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
static inline Vec3 vabs(Vec3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
static inline Vec3 vmin(Vec3 a, Vec3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
static inline Vec3 vmax(Vec3 a, Vec3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
static inline Vec3 vsignum(Vec3 a) { return {fsignum(a.x), fsignum(a.y), fsignum(a.z)}; }
static inline Vec3 vpowf(Vec3 a, float n) { return {orc_powf(a.x, n), orc_powf(a.y, n), orc_powf(a.z, n)}; }

static inline Vec4 v4(float x, float y, float z, float w) { return Vec4{x, y, z, w}; }
static inline Vec4 v4(Vec3 a, float w) { return Vec4{a.x, a.y, a.z, w}; }
static inline Vec3 xyz(Vec4 a) { return {a.x, a.y, a.z}; }
static inline Vec4 operator+(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline Vec4 operator-(Vec4 a, Vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline Vec4 operator*(Vec4 a, Vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline Vec4 operator*(Vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline Vec4 operator/(Vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
// Vec4::clamp on SSE2 = _mm_min_ps(_mm_max_ps(self, min), max): a NaN lane takes the bound.
static inline float sse_max(float a, float b) { return a > b ? a : b; }
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline Vec4 clamp01(Vec4 a) {
    return {sse_min(sse_max(a.x, 0.0f), 1.0f), sse_min(sse_max(a.y, 0.0f), 1.0f),
            sse_min(sse_max(a.z, 0.0f), 1.0f), sse_min(sse_max(a.w, 0.0f), 1.0f)};
}

// Mat4: column-major, c[j] is column j.
struct Mat4 { Vec4 c[4]; };
static inline Mat4 mat4_from_cols_array(const float *m) {
    Mat4 r;
    for (int j = 0; j < 4; j++) r.c[j] = v4(m[4 * j], m[4 * j + 1], m[4 * j + 2], m[4 * j + 3]);
    return r;
}
static inline void mat4_to_cols_array(const Mat4 &m, float *o) {
    for (int j = 0; j < 4; j++) { o[4 * j] = m.c[j].x; o[4 * j + 1] = m.c[j].y; o[4 * j + 2] = m.c[j].z; o[4 * j + 3] = m.c[j].w; }
}
static inline Mat4 mat4_identity() {
    Mat4 r; r.c[0] = v4(1, 0, 0, 0); r.c[1] = v4(0, 1, 0, 0); r.c[2] = v4(0, 0, 1, 0); r.c[3] = v4(0, 0, 0, 1); return r;
}
// Mat4::mul_vec4: ((x_axis*x + y_axis*y) + z_axis*z) + w_axis*w
static inline Vec4 operator*(const Mat4 &m, Vec4 v) {
    Vec4 res = m.c[0] * v4(v.x, v.x, v.x, v.x);
    res = res + m.c[1] * v4(v.y, v.y, v.y, v.y);
    res = res + m.c[2] * v4(v.z, v.z, v.z, v.z);
    res = res + m.c[3] * v4(v.w, v.w, v.w, v.w);
    return res;
}
static inline Mat4 operator*(const Mat4 &a, const Mat4 &b) {
    Mat4 r; for (int j = 0; j < 4; j++) r.c[j] = a * b.c[j]; return r;
}
static inline Mat4 mat4_from_translation(Vec3 t) {
    Mat4 r = mat4_identity(); r.c[3] = v4(t.x, t.y, t.z, 1.0f); return r;
}
static inline Mat4 mat4_from_scale(Vec3 s) {
    Mat4 r; r.c[0] = v4(s.x, 0, 0, 0); r.c[1] = v4(0, s.y, 0, 0); r.c[2] = v4(0, 0, s.z, 0); r.c[3] = v4(0, 0, 0, 1); return r;
}
static inline Mat4 mat4_from_rotation_x(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r; r.c[0] = v4(1, 0, 0, 0); r.c[1] = v4(0, c, s, 0); r.c[2] = v4(0, -s, c, 0); r.c[3] = v4(0, 0, 0, 1); return r;
}
static inline Mat4 mat4_from_rotation_y(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r; r.c[0] = v4(c, 0, -s, 0); r.c[1] = v4(0, 1, 0, 0); r.c[2] = v4(s, 0, c, 0); r.c[3] = v4(0, 0, 0, 1); return r;
}
static inline Mat4 mat4_from_rotation_z(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r; r.c[0] = v4(c, s, 0, 0); r.c[1] = v4(-s, c, 0, 0); r.c[2] = v4(0, 0, 1, 0); r.c[3] = v4(0, 0, 0, 1); return r;
}
// Mat4::perspective_rh (depth 0..1); no argument validation in release builds.
static inline Mat4 mat4_perspective_rh(float fov_y, float aspect, float z_near, float z_far) {
    float sf = std::sin(0.5f * fov_y), cf = std::cos(0.5f * fov_y);
    float h = cf / sf;
    float w = h / aspect;
    float r = z_far / (z_near - z_far);
    Mat4 m; m.c[0] = v4(w, 0, 0, 0); m.c[1] = v4(0, h, 0, 0); m.c[2] = v4(0, 0, r, -1.0f); m.c[3] = v4(0, 0, r * z_near, 0); return m;
}
static inline Mat4 mat4_look_to_rh(Vec3 eye, Vec3 dir, Vec3 up) {
    Vec3 f = normalize(dir);
    Vec3 s = normalize(cross(f, up));
    Vec3 u = cross(s, f);
    Mat4 m;
    m.c[0] = v4(s.x, u.x, -f.x, 0);
    m.c[1] = v4(s.y, u.y, -f.y, 0);
    m.c[2] = v4(s.z, u.z, -f.z, 0);
    m.c[3] = v4(-dot(eye, s), -dot(eye, u), dot(eye, f), 1.0f);
    return m;
}
// look_at_lh(eye, center, up) = look_to_lh(eye, center - eye, up) = look_to_rh(eye, -(center - eye), up)
static inline Mat4 mat4_look_at_lh(Vec3 eye, Vec3 center, Vec3 up) { return mat4_look_to_rh(eye, -(center - eye), up); }

// Mat4::inverse: cofactor expansion in glam's term order, times 1/det.
static inline Mat4 mat4_inverse(const Mat4 &m) {
    float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z, m03 = m.c[0].w;
    float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z, m13 = m.c[1].w;
    float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z, m23 = m.c[2].w;
    float m30 = m.c[3].x, m31 = m.c[3].y, m32 = m.c[3].z, m33 = m.c[3].w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    Vec4 fac0 = v4(coef00, coef00, coef02, coef03), fac1 = v4(coef04, coef04, coef06, coef07);
    Vec4 fac2 = v4(coef08, coef08, coef10, coef11), fac3 = v4(coef12, coef12, coef14, coef15);
    Vec4 fac4 = v4(coef16, coef16, coef18, coef19), fac5 = v4(coef20, coef20, coef22, coef23);
    Vec4 vec0 = v4(m10, m00, m00, m00), vec1 = v4(m11, m01, m01, m01);
    Vec4 vec2 = v4(m12, m02, m02, m02), vec3 = v4(m13, m03, m03, m03);
    Vec4 inv0 = (vec1 * fac0 - vec2 * fac1) + vec3 * fac2;
    Vec4 inv1 = (vec0 * fac0 - vec2 * fac3) + vec3 * fac4;
    Vec4 inv2 = (vec0 * fac1 - vec1 * fac3) + vec3 * fac5;
    Vec4 inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5;
    Vec4 sign_a = v4(1.0f, -1.0f, 1.0f, -1.0f), sign_b = v4(-1.0f, 1.0f, -1.0f, 1.0f);
    Mat4 inv; inv.c[0] = inv0 * sign_a; inv.c[1] = inv1 * sign_b; inv.c[2] = inv2 * sign_a; inv.c[3] = inv3 * sign_b;
    Vec4 col0 = v4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    Vec4 dot0 = m.c[0] * col0;
    float dot1 = dot0.x + dot0.y + dot0.z + dot0.w;
    float rcp_det = 1.0f / dot1;
    Mat4 r; for (int j = 0; j < 4; j++) r.c[j] = inv.c[j] * rcp_det; return r;
}

static const float DEGREES = 3.14159274101257324f / 180.0f;  // geometry.rs:5
static const float EPSILON = 0.0001f;                        // ray.rs:9
static const float INV_PI = 1.0f / 3.14159274101257324f;     // sphere.rs:13

}  // namespace orc
