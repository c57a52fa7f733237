// Exhaustive check on the device: fused sqrt + reciprocal from one v_rsq_f32 seed against the IEEE sequence
// fl(1 / fl(sqrt(d))) over every binary32 bit pattern.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt -o norm_check norm_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline void fast_sqrt_recip(float d, int steps, float &sq, float &rc) {
    float r = __builtin_amdgcn_rsqf(d);
    float g = d * r;
    float h = 0.5f * r;
    float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    float dd = __builtin_fmaf(-g, g, d);
    g = __builtin_fmaf(dd, h, g);
    sq = g;
    float y = h + h;
    // steps: 1 = newton, newton; 2 = newton, cubic; 3 = cubic; (cubic: y * (1 + e + e^2), which resolves the near-tie of
    // an all-ones significand that plain Newton rounds the wrong way)
    if (steps != 3) {
        float e2 = __builtin_fmaf(-g, y, 1.0f);
        y = __builtin_fmaf(e2, y, y);
    }
    float e3 = __builtin_fmaf(-g, y, 1.0f);
    if (steps != 1) e3 = __builtin_fmaf(e3, e3, e3);
    y = __builtin_fmaf(e3, y, y);
    rc = y;
}

// counters: [0] sqrt mismatches, [1..3] reciprocal mismatches of the three refinement variants
__global__ void check(unsigned long long *cnt, uint32_t lo_bits, uint32_t hi_bits, unsigned int *first_bad) {
    const uint64_t n = (uint64_t)hi_bits - lo_bits;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = lo_bits + (uint32_t)i;
        const float d = __uint_as_float(u);
        const float sq_ref = __builtin_sqrtf(d);
        const float rc_ref = 1.0f / sq_ref;
        for (int steps = 1; steps <= 3; steps++) {
            float sq, rc;
            fast_sqrt_recip(d, steps, sq, rc);
            if (steps == 1 && __float_as_uint(sq) != __float_as_uint(sq_ref)) { atomicAdd(&cnt[0], 1ull); atomicMin(&first_bad[0], u); }
            if (__float_as_uint(rc) != __float_as_uint(rc_ref)) { atomicAdd(&cnt[steps], 1ull); atomicMin(&first_bad[steps], u); }
        }
    }
}

int main() {
    unsigned long long *cnt; unsigned int *bad;
    (void)hipMalloc(&cnt, 4 * sizeof(*cnt)); (void)hipMalloc(&bad, 4 * sizeof(*bad));
    // positive normal numbers with exponent in [-100, 100]: bits [(27) << 23, (228) << 23)
    const int ranges[][2] = {{-126, -100}, {-100, -64}, {-64, 64}, {64, 100}, {100, 128}};
    for (auto &r : ranges) {
        uint32_t lo = (uint32_t)(r[0] + 127) << 23, hi = (uint32_t)(r[1] + 127) << 23;
        (void)hipMemset(cnt, 0, 4 * sizeof(*cnt)); (void)hipMemset(bad, 0xFF, 4 * sizeof(*bad));
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, cnt, lo, hi, bad);
        unsigned long long h[4]; unsigned int b[4];
        (void)hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(b, bad, sizeof(b), hipMemcpyDeviceToHost);
        printf("exponent [%d, %d): sqrt mismatches %llu (first %08x); reciprocal mismatches newton+newton / newton+cubic / cubic: %llu (%08x) / %llu (%08x) / %llu (%08x)\n",
               r[0], r[1], h[0], b[0], h[1], b[1], h[2], b[2], h[3], b[3]);
    }
    return 0;
}
