// Micro-benchmark: HBM read efficiency of a column-major n x m fp64 matrix as a function of the contiguous
// segment each wave instruction reads per column (64/CW rows * 8 B), with the access order of onepass_kernel
// (a wave visits all m columns for one row tile, then the next tile).  Pure loads + one add; no register tile.
//   hipcc --offload-arch=gfx950 -O3 -o segprobe tools/micro/segprobe.hip && ./segprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int CW, int UNR>
__global__ __launch_bounds__(256) void probe(const double* __restrict__ M, int64_t ld, int m, double* out) {
    constexpr int RW = 64 / CW, kStep = RW * 4, kTiles = 2048 / kStep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & (RW - 1), h = lane / RW;
    const double* base = M + (int64_t)blockIdx.x * 2048 + wave * RW + r + (int64_t)h * ld;
    double acc = 0.0;
    const int groups = m / CW;
    for (int k = 0; k < kTiles; ++k) {
        const double* p = base + (int64_t)k * kStep;
        for (int c = 0; c < groups; c += UNR) {
            double v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(c + u) * CW * ld);
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc += v[u];
        }
    }
    if (acc == 123.456) out[0] = acc;
}

// persistent variant: `grid` workgroups, each a contiguous, balanced span of 64-row tile rounds
template <int CW, int UNR>
__global__ __launch_bounds__(256) void probe_persist(const double* __restrict__ M, int64_t ld, int m, int64_t rounds, double* out) {
    constexpr int RW = 64 / CW, kStep = RW * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & (RW - 1), h = lane / RW;
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem), t1 = t0 + q + (blockIdx.x < rem ? 1 : 0);
    const double* base = M + wave * RW + r + (int64_t)h * ld;
    double acc = 0.0;
    const int groups = m / CW;
    for (int64_t k = t0; k < t1; ++k) {
        const double* p = base + k * kStep;
        for (int c = 0; c < groups; c += UNR) {
            double v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(c + u) * CW * ld);
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc += v[u];
        }
    }
    if (acc == 123.456) out[0] = acc;
}
template <int CW, int UNR>
static void run_persist(const double* M, int64_t ld, int64_t n, int m, double* out, int wg_per_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = 256 * wg_per_cu;
    const int64_t rounds = n / (64 / CW * 4);
    probe_persist<CW, UNR><<<grid, 256>>>(M, ld, m, rounds, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe_persist<CW, UNR><<<grid, 256>>>(M, ld, m, rounds, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("persistent CW=%d unroll=%2d grid=256x%d : %.3f ms  %.0f GB/s\n", CW, UNR, wg_per_cu, ms, 8.0 * n * m / (ms * 1e-3) / 1e9);
}

template <int CW, int UNR>
static void run(const double* M, int64_t ld, int64_t n, int m, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(n / 2048);
    probe<CW, UNR><<<grid, 256>>>(M, ld, m, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<CW, UNR><<<grid, 256>>>(M, ld, m, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("CW=%d segment=%4d B  unroll=%2d : %.3f ms  %.0f GB/s\n", CW, 64 / CW * 8, UNR, ms, 8.0 * n * m / (ms * 1e-3) / 1e9);
}

int main() {
    const int64_t n = 10000384, ld = n;   // multiple of 2048
    const int m = 128;
    double *M, *out;
    hipMalloc(&M, sizeof(double) * ld * m);
    hipMalloc(&out, 64);
    hipMemset(M, 0, sizeof(double) * ld * m);
    run<1, 16>(M, ld, n, m, out);
    run<2, 16>(M, ld, n, m, out);
    run<4, 16>(M, ld, n, m, out);
    run<8, 16>(M, ld, n, m, out);
    run<1, 32>(M, ld, n, m, out);
    run<2, 32>(M, ld, n, m, out);
    run<4, 32>(M, ld, n, m, out);
    run<4, 8>(M, ld, n, m, out);
    run<2, 8>(M, ld, n, m, out);
    for (int w : {2, 3, 4, 6, 8}) run_persist<4, 16>(M, ld, n, m, out, w);
    for (int w : {2, 4, 8}) run_persist<1, 16>(M, ld, n, m, out, w);
    for (int w : {2, 3, 4}) run_persist<4, 32>(M, ld, n, m, out, w);
    return 0;
}
