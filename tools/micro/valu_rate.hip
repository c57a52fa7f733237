// Microbenchmark: sustained wave64 VALU issue rate per SIMD for a few instruction kinds at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {  // v_fma_f32
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (KIND == 1) {  // v_mul_f32 + v_add_f32 (no contraction)
#pragma clang fp contract(off)
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
        } else if (KIND == 2) {  // v_fma_f64
            d0 = __builtin_fma(d0, (double)a, (double)b); d1 = __builtin_fma(d1, (double)a, (double)b);
            d2 = __builtin_fma(d2, (double)a, (double)b); d3 = __builtin_fma(d3, (double)a, (double)b);
        } else if (KIND == 3) {  // v_mul_f64 / v_add_f64
#pragma clang fp contract(off)
            d0 = d0 * (double)a; d1 = d1 * (double)a; d2 = d2 + (double)b; d3 = d3 + (double)b;
        } else if (KIND == 4) {  // v_cndmask + v_cmp
            x0 = x0 > a ? x1 : x0; x2 = x2 > a ? x3 : x2; x4 = x4 > b ? x5 : x4; x6 = x6 > b ? x7 : x6;
            x1 += 1.0f; x3 += 1.0f; x5 += 1.0f; x7 += 1.0f;
        } else if (KIND == 5) {  // IEEE f32 division
            x0 = a / x0; x1 = b / x1; x2 = a / x2; x3 = b / x3;
        } else if (KIND == 6) {  // IEEE f32 sqrt
            x0 = __builtin_sqrtf(x0 + a); x1 = __builtin_sqrtf(x1 + a); x2 = __builtin_sqrtf(x2 + a); x3 = __builtin_sqrtf(x3 + a);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3);
}

template <int KIND>
void run(const char *name, int ops_per_iter, int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu, iters = 20000;
    float *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr_per_simd = (double)iters * ops_per_iter * blocks_per_cu;  // one wave of every block sits on each SIMD
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles (at 2.4 GHz) per source-level op per SIMD\n", name, blocks_per_cu, ms, cycles / wave_instr_per_simd);
    CHECK(hipFree(out));
}

int main() {
    for (int occ : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 8, occ);
        run<1>("v_mul_f32 / v_add_f32", 8, occ);
        run<2>("v_fma_f64", 4, occ);
        run<3>("v_mul_f64 / v_add_f64", 4, occ);
        run<4>("v_cmp + v_cndmask + v_add", 4, occ);
        run<5>("IEEE f32 divide", 4, occ);
        run<6>("IEEE f32 sqrt (+ add)", 4, occ);
    }
    return 0;
}
