// rt_dev_math.h — device-side f32 vector math and the two transcendental functions of the hot path.
//
// Arithmetic contract (what makes the kernels reproduce the CPU reference bit for bit wherever only
// IEEE-754 basic operations are involved): every expression keeps the reference's operation order,
// nothing is contracted into an FMA, division and square root are correctly rounded, subnormals are
// kept.  Build with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt and no fast-math flag.
//
// rt_powf / rt_atanf are fully specified algorithms (binary64 table-driven exp2(y*log2 x) with explicit FMAs; the classic
// single-precision fdlibm arctangent) rather than calls into the device libm, so that the values the
// discontinuous parts of the renderer depend on (texel indices from atan-derived UVs) do not depend on
// which libm a platform ships.  Reference call sites: f32::powf src/ray.rs:81, src/ray_tracing/ray_tracing.rs:55,
// src/ray_marching/ray_marching.rs:178, src/utils/geometry.rs:65; f32::atan src/utils/sphere.rs:136-137,
// src/utils/cylinder.rs:119, src/utils/cone.rs:105.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define RT_DEV __device__ __forceinline__

struct f2 { float x, y; };
// Three-component vector as a compiler vector type: +, -, *, / act per component (each an IEEE operation, in the
// written order), a scalar operand is splat, and the x/y halves map onto packed v_pk_* instructions without the operand
// shuffles the automatic vectorizer needs for scalars.
typedef float f3 __attribute__((ext_vector_type(3)));
struct f4 { float x, y, z, w; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
RT_DEV f3 splat3(float s) { return mk3(s, s, s); }
RT_DEV float dot(f3 a, f3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
RT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// f32::sqrt, correctly rounded.  For positive normal numbers from 2^-96 on this is the compiler's own expansion of an IEEE
// square root -- v_sqrt_f32 is within one unit in the last place, and the residuals x - s * (s -+ ulp), one FMA each, pick the
// correctly rounded neighbour -- without the input scaling and the special-value select that only the other inputs need
// (zeros, subnormals, numbers below 2^-96, infinities, NaNs, negatives).  The short form runs when every lane of the
// wavefront holds such a number (a wave-uniform branch: no exec-mask bookkeeping), the full expansion otherwise; 9 instead
// of 17 VALU instructions.  tests/test_gpu_math.py compares it with the correctly rounded root on every positive binary32 number.
RT_DEV float rt_sqrt(float x) {
#ifndef RT_PLAIN_SQRT
    const bool usual = __float_as_uint(x) - 0x0F800000u < 0x7F800000u - 0x0F800000u;
    if (__builtin_amdgcn_ballot_w64(!usual) == 0ull) {
        const float s = __builtin_amdgcn_sqrtf(x);
        const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
        const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
        float r = r_dn <= 0.0f ? s_dn : s;
        r = r_up > 0.0f ? s_up : r;
        return r;
    }
#endif
    return __builtin_sqrtf(x);
}
RT_DEV float length(f3 a) { return rt_sqrt(dot(a, a)); }
RT_DEV f3 normalize(f3 a) { return a * (1.0f / length(a)); }  // glam: self * length().recip()
RT_DEV float rt_abs(float x) { return __builtin_fabsf(x); }
RT_DEV f3 vabs(f3 a) { return mk3(rt_abs(a.x), rt_abs(a.y), rt_abs(a.z)); }
// f32::max / f32::min (a NaN operand is ignored)
RT_DEV float rt_max(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float rt_min(float a, float b) { return __builtin_fminf(a, b); }
RT_DEV f3 vmax(f3 a, f3 b) { return mk3(rt_max(a.x, b.x), rt_max(a.y, b.y), rt_max(a.z, b.z)); }
RT_DEV f3 vmin(f3 a, f3 b) { return mk3(rt_min(a.x, b.x), rt_min(a.y, b.y), rt_min(a.z, b.z)); }
// f32::clamp (NaN stays NaN)
RT_DEV float rt_clamp(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
// f32::signum
RT_DEV float rt_signum(float x) { if (x != x) return x; return (__float_as_uint(x) >> 31) ? -1.0f : 1.0f; }
// Vec4::clamp(0,1) as SSE2 min(max(x,0),1): a NaN lane becomes 0
RT_DEV float rt_clamp01_sse(float x) { float m = x > 0.0f ? x : 0.0f; return m < 1.0f ? m : 1.0f; }
// `as u8` / `as u32`: truncate, saturate, NaN -> 0
// Rust `f as u8`: NaN and negatives -> 0, above 255 -> 255, else truncation.  Branch-free: fmaxf drops a NaN operand, so the
// value is in [0, 255] before the conversion.
RT_DEV uint32_t rt_as_u8(float f) { return (uint32_t)__builtin_fminf(__builtin_fmaxf(f, 0.0f), 255.0f); }
RT_DEV uint32_t rt_as_u32(float f) { if (!(f > 0.0f)) return 0u; if (f >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)f; }

// ------------------------------------------------------------------------------------------------
// powf: exp2(y * log2 x) in binary64; constants generated by tools/gen_pow_tables.py
// ------------------------------------------------------------------------------------------------
#define RT_POW_TABLE_QUAL __device__ __constant__ __attribute__((aligned(16))) const
#define RT_POW_TABLE_NAME(n) rt_pow_##n
#include "rt_pow_tables.inc"
#undef RT_POW_TABLE_QUAL
#undef RT_POW_TABLE_NAME

// Every evaluation has two DEPENDENT table lookups (interval -> 1/c, log2 c; then k -> 2^(k/32)).  The 1 KB of tables
// is copied into LDS by every workgroup (RT_POW_LDS 1, the default: each kernel declares the array, fills it with
// rt_pow_lds_init() before its first barrier and hands it down as a pointer; layout [0, 94) (1/c_i, log2 c_i) pairs,
// [94, 126) 2^(i/32)) or read from constant memory (RT_POW_LDS 0: an L1/L2 hit per lookup).  The LDS copy cuts the time
// inside powf by two thirds; at 6 waves per SIMD most of that latency was already hidden, so frames gain about 1 %.
// The staged log2 c_i are multiplied by 32 (exact): rt_pow_core evaluates 32 log2 x directly.
#ifndef RT_POW_LDS
#define RT_POW_LDS 1
#endif
#define RT_POW_LDS_DOUBLES (RT_POW_LDS ? 47 * 2 + 32 : 2)
typedef const double *PowTab;
RT_DEV void rt_pow_lds_init(double *tab) {
#if RT_POW_LDS
    // one load per staged entry (the address is selected, not the value: the two source tables cost one memory round trip)
    for (uint32_t i = threadIdx.x; i < (uint32_t)RT_POW_LDS_DOUBLES; i += blockDim.x) {
        const unsigned long long *src = i < 94u ? reinterpret_cast<const unsigned long long *>(rt_pow_logtab) + i
                                                : reinterpret_cast<const unsigned long long *>(rt_pow_exp2bits) + (i - 94u);
        const double v = __longlong_as_double((long long)*src);
        tab[i] = (i < 94u && (i & 1u)) ? v * 32.0 : v;
    }
#else
    (void)tab;
#endif
}
RT_DEV double2 rt_pow_log_entry(PowTab tab, int i) {
#if RT_POW_LDS
    return reinterpret_cast<const double2 *>(tab)[i];
#else
    (void)tab;
    double2 v = reinterpret_cast<const double2 *>(rt_pow_logtab)[i];
    v.y = v.y * 32.0;
    return v;
#endif
}
RT_DEV long long rt_pow_exp2_entry(PowTab tab, int i) {
#if RT_POW_LDS
    return __double_as_longlong(tab[94 + i]);
#else
    (void)tab;
    return (long long)rt_pow_exp2bits[i];
#endif
}

// powf(ax, y) for ax > 0 finite: exp2(y * log2(ax)) in binary64 with fused multiply-adds
// (algorithm and constants: tools/gen_pow_tables.py).
RT_DEV float rt_pow_core(float ax, float y, PowTab tab_lds) {
    // ---- log2(ax): range reduction on the binary32 bits (subnormals are scaled up first)
    int e_adj = 0;
    if (ax < 1.17549435e-38f) { ax = ax * 16777216.0f; e_adj = -24; }
    const uint32_t ix = __float_as_uint(ax);
    int e = (int)(ix >> 23) - 127 + e_adj;
    float zf = __uint_as_float((ix & 0x007FFFFFu) | 0x3F800000u);  // [1, 2)
    if (zf >= 1.4609375f) { zf = zf * 0.5f; e += 1; }               // [187/256, 187/128)
    const int i = (int)(zf * 64.0f - 46.5f);                        // both operations exact
    const double2 tab = rt_pow_log_entry(tab_lds, i);                 // (1/c_i, 32 log2 c_i)
    const double r = __builtin_fma((double)zf, tab.x, -1.0);
    // The specified algorithm computes lg = fma(P(r), r, e + log2 c_i), E = y * lg clamped to +-160, t = 32 E and
    // kd = (t + 1.5 * 2^52) - 1.5 * 2^52.  Below, every term of lg carries the factor 32 from the start: scaling by 2^5
    // commutes with each rounding, so t has the same bits, with one multiplication less; the clamp is a max/min pair and
    // the rounding a single round-to-nearest-even instruction, both with the same results.
    // (pinning the coefficients to SGPRs with an empty asm removes four scratch reloads per evaluation but measured 12 % slower:
    // the asm statements stop the scheduler from interleaving the three evaluations of a Vec3::powf)
    double p = RT_POW_L5 * 32.0;
    p = __builtin_fma(p, r, RT_POW_L4 * 32.0);
    p = __builtin_fma(p, r, RT_POW_L3 * 32.0);
    p = __builtin_fma(p, r, RT_POW_L2 * 32.0);
    p = __builtin_fma(p, r, RT_POW_L1 * 32.0);
    const double lg32 = __builtin_fma(p, r, __builtin_fma((double)e, 32.0, tab.y));
    // ---- exp2(y * log2(ax))
    double t = (double)y * lg32;
    t = __builtin_fmin(__builtin_fmax(t, -5120.0), 5120.0);  // |y log2 x| >= 160 is beyond binary32's range either way
    const double kd = __builtin_rint(t);
    const int k = (int)kd;
    const double rr = t - kd;
    double q = RT_POW_C4;
    q = __builtin_fma(q, rr, RT_POW_C3);
    q = __builtin_fma(q, rr, RT_POW_C2);
    q = __builtin_fma(q, rr, RT_POW_C1);
    q = q * rr;
    const double s = __longlong_as_double(rt_pow_exp2_entry(tab_lds, k & 31) + (long long)((uint64_t)(long long)(k >> 5) << 52));
    return (float)__builtin_fma(s, q, s);
}

// 0: not an integer, 1: odd integer, 2: even integer (y finite, non-zero)
RT_DEV int rt_pow_yint(float y) {
    float ay = rt_abs(y);
    if (ay >= 16777216.0f) return 2;
    if (ay < 1.0f) return 0;
    long long iy = (long long)ay;
    if ((float)iy != ay) return 0;
    return (iy & 1) ? 1 : 2;
}

__device__ __noinline__ float rt_powf_special(float x, float y, PowTab tab_lds) {
    const float inf = __builtin_inff();
    if (x != x || y != y) return x + y;
    if (y == inf || y == -inf) {
        float ax = rt_abs(x);
        if (ax == 1.0f) return 1.0f;
        return ((ax > 1.0f) == (y > 0.0f)) ? inf : 0.0f;
    }
    int yi = rt_pow_yint(y);
    if (x == 0.0f) {
        if (y < 0.0f) return (yi == 1) ? __builtin_copysignf(inf, x) : inf;
        return (yi == 1) ? x : 0.0f;
    }
    if (x == inf) return (y < 0.0f) ? 0.0f : inf;
    if (x == -inf) {
        if (y < 0.0f) return (yi == 1) ? -0.0f : 0.0f;
        return (yi == 1) ? -inf : inf;
    }
    if (yi == 0) return __builtin_nanf("");
    float r = rt_pow_core(-x, y, tab_lds);
    return (yi == 1) ? -r : r;
}

RT_DEV float rt_powf(float x, float y, PowTab tab_lds) {
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (__float_as_uint(x) == 0u && y > 0.0f) return 0.0f;  // pow(+0, y > 0) = +0: clamped dot products are often exactly 0
    if (x > 0.0f && x != __builtin_inff() && y == y && rt_abs(y) != __builtin_inff()) return rt_pow_core(x, y, tab_lds);
    return rt_powf_special(x, y, tab_lds);
}
// Vec3::powf.  When all three bases are positive and finite (the usual case for an accumulated light colour) the three
// evaluations run as straight-line code behind one branch instead of three.  rt_pow_core(1, y) is exactly 1, so x == 1 needs no special case there.
RT_DEV f3 vpowf(f3 a, float n, PowTab tab_lds) {
    const float inf = __builtin_inff();
    const bool fast = a.x > 0.0f && a.x < inf && a.y > 0.0f && a.y < inf && a.z > 0.0f && a.z < inf && n != 0.0f && rt_abs(n) < inf;
    if (fast) {
        // three independent chains: the scheduler interleaves them, so the table lookups of one hide behind the arithmetic of the others
        const float x = rt_pow_core(a.x, n, tab_lds);
        const float y = rt_pow_core(a.y, n, tab_lds);
        const float z = rt_pow_core(a.z, n, tab_lds);
        return mk3(x, y, z);
    }
    return mk3(rt_powf(a.x, n, tab_lds), rt_powf(a.y, n, tab_lds), rt_powf(a.z, n, tab_lds));
}

// ------------------------------------------------------------------------------------------------
// atanf: fdlibm single-precision algorithm in f32 arithmetic
// ------------------------------------------------------------------------------------------------
RT_DEV float rt_atanf(float x) {
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    uint32_t hx = __float_as_uint(x);
    uint32_t ix = hx & 0x7fffffffu;
    bool neg = (hx >> 31) != 0;
    float hi = 0.0f, lo = 0.0f;
    bool reduced = true;
    if (ix >= 0x4c000000u) {
        if (ix > 0x7f800000u) return x + x;
        float z = 1.5707962513e+00f + 7.5497894159e-08f;
        return neg ? -z : z;
    }
    if (ix < 0x3ee00000u) {
        if (ix < 0x31000000u) return x;
        reduced = false;
    } else {
        x = rt_abs(x);
        if (ix < 0x3f980000u) {
            if (ix < 0x3f300000u) { hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000u) { hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (!reduced) return x - x * (s1 + s2);
    float r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
}

// ------------------------------------------------------------------------------------------------
// counter-based generator standing in for rand's ThreadRng (reference: src/renderer.rs:112, src/ray.rs:114-127);
// u32 -> f32 mapping as rand 0.8.5 UniformFloat<f32>: (bits >> 9) * 2^-23 * (high - low) + low
// ------------------------------------------------------------------------------------------------
RT_DEV uint32_t rt_pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
RT_DEV uint32_t rt_rng_path_key(uint64_t seed, uint32_t pixel, uint32_t frame) {
    uint32_t k = rt_pcg_hash((uint32_t)seed ^ rt_pcg_hash((uint32_t)(seed >> 32)));
    k = rt_pcg_hash(k + pixel);
    k = rt_pcg_hash(k + frame);
    return k;
}
RT_DEV uint32_t rt_rng_child_key(uint32_t key, uint32_t branch) { return rt_pcg_hash(key + 0x9E3779B9u * (branch + 1u)); }
RT_DEV float rt_rng_uniform(uint32_t key, uint32_t draw, float low, float high) {
    uint32_t bits = rt_pcg_hash(key ^ (0x85EBCA6Bu * (draw + 1u)));
    float value0_1 = (float)(bits >> 9) * 1.1920928955078125e-7f;
    return value0_1 * (high - low) + low;
}
