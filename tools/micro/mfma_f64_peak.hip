// Sustained rate of v_mfma_f64_16x16x4_f64 with no memory traffic: every wave runs ITERS x 16 independent accumulators.
// Calibrates the "MFMA peak" the tangent-setup kernels (csrc/factorize.hip) are priced against: the data-sheet 78.6 TFLOP/s
// assumes the 2.4 GHz peak clock; the sustained clock under a chip-wide fp64 matrix load is what this measures.
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma_f64_peak mfma_f64_peak.hip && ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((vector_size(32)));

// RANDOM = operands differ per lane and per instruction (uniform in [-1, 1)): the data-dependent part of the power draw
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double seed) {
    f64x4 acc[NACC];
    double a[NACC], b[NACC];
    unsigned long long h = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        a[i] = RANDOM ? (double)(long long)h * (1.0 / 9223372036854775808.0) : seed + threadIdx.x * 1e-9;
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        b[i] = RANDOM ? (double)(long long)h * (1.0 / 9223372036854775808.0) : seed - threadIdx.x * 1e-9;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[(i + 1) % NACC]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;     // keeps the loop alive
}

template <int NACC, bool RANDOM>
static void run(int wgs_per_cu, int cus, int iters, double ms_target) {
    double* d;
    hipMalloc(&d, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = wgs_per_cu * cus;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, RANDOM>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * NACC * 2048.0;
        printf("%s acc %2d  %d WG/CU (%d waves/SIMD)  iters %d: %8.3f ms  %6.2f TFLOP/s\n", RANDOM ? "random operands  " : "constant operands", NACC, wgs_per_cu, wgs_per_cu, iters, ms, flop / ms / 1e9);
    }
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    run<16, false>(1, cus, 2000, 0);      // ~ 1 ms: a first launch out of idle
    run<16, false>(1, cus, 40000, 0);     // ~ 17 ms
    run<9, false>(2, cus, 40000, 0);
    run<16, true>(1, cus, 40000, 0);
    run<16, true>(1, cus, 400000, 0);     // ~ 0.2 s: long enough for the power management to settle
    run<9, true>(2, cus, 40000, 0);
    return 0;
}
