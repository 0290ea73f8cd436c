// Micro-benchmark for the open question of DESIGN.md 6: is the allocation-pair sensitivity of the fused projected-CG kernel (and its 0.81
// UTCL1 misses per 128-byte request) a property of the ACCESS PATTERN of a 16-row x 128-column register tile over a column-major matrix --
// 128 pages touched per tile -- or of the memory system regardless of pattern?  Two kernels with F's structure (persistent grid, one wave
// = one 16-row tile held in registers between two products, three row-vector loads and one in-place store per tile, NT matrix loads)
// differ ONLY in the matrix layout they walk:
//   colmajor : M[row + col * ld]                      -- what the library does (4 x 128-byte pieces per wave instruction)
//   panel    : M[(row/16) * 16*m + col * 16 + row%16] -- row panels of 16 rows, a tile = 16 KB contiguous (512 bytes per instruction)
// on KZ allocations of the matrix x KW allocations of the vectors, every pair timed.
//   hipcc --offload-arch=gfx950 -O3 -o layoutprobe tools/micro/layoutprobe.hip && ./layoutprobe [KZ] [KW]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int M_COLS = 128, CPL = 32;      // 32 column registers per lane x 4 lane groups

template <bool PANEL>
__global__ __launch_bounds__(256) void tile_kernel(const double* __restrict__ M, int64_t ld, int64_t rounds, const double* __restrict__ g_in,
                                                   double* __restrict__ g_out, const double* __restrict__ d, const double* __restrict__ a,
                                                   double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;       // F's lane layout: row r of the tile, column group h
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem);
    const int cnt = (int)(q + (blockIdx.x < rem ? 1 : 0));
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    double s_red = 0.0;
    for (int k = 0; k < cnt; ++k) {
        const int64_t row0 = (t0 + k) * 64 + wave * 16;                       // this wave's tile
        double av[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int col = 4 * c + h;
            const double* p = PANEL ? M + (row0 / 16) * (16 * (int64_t)M_COLS) + (int64_t)col * 16 + r : M + row0 + r + (int64_t)col * ld;
            av[c] = __builtin_nontemporal_load(p);
        }
        const int64_t row = row0 + r;
        const double gg = g_in[row], dd = d[row], aa = a[row];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1), s);       // first product
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;                  // row update
        if (h == 0) g_out[row] = gp;
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);   // second product (no cross-lane reduction: traffic only)
    }
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red;
    if (tot == 123.456) part[blockIdx.x] = tot;
}

int main(int argc, char** argv) {
    const int KZ = argc > 1 ? atoi(argv[1]) : 3, KW = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t n = 10000000, nround = (n + 63) / 64, npad = nround * 64, ld = npad + 16;
    const size_t mbytes = sizeof(double) * (size_t)ld * M_COLS;
    std::vector<double*> Z(KZ), G(KW), D(KW), A(KW);
    for (int k = 0; k < KZ; ++k) { hipMalloc((void**)&Z[k], mbytes); hipMemset(Z[k], 0, mbytes); }
    for (int k = 0; k < KW; ++k) {
        double* slab;
        hipMalloc((void**)&slab, sizeof(double) * (size_t)npad * 3);
        hipMemset(slab, 0, sizeof(double) * (size_t)npad * 3);
        G[k] = slab; D[k] = slab + npad; A[k] = slab + 2 * npad;
    }
    double* part;
    hipMalloc((void**)&part, 8 * 4096);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tile_kernel<false>, 256, 0);
    const unsigned grid = 256u * (unsigned)(nb > 0 ? nb : 1);
    printf("n=%lld m=%d, %d workgroups (%d per CU); matrix %.2f GB; F-like traffic %.3f GB per launch\n", (long long)n, M_COLS, grid, nb, mbytes / 1e9,
           (8.0 * n * M_COLS + 32.0 * n) / 1e9);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 300; ++w) tile_kernel<false><<<grid, 256>>>(Z[0], ld, nround, G[0], G[0], D[0], A[0], part);   // warm the device
    hipDeviceSynchronize();
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int iz = 0; iz < KZ; ++iz)
            for (int iw = 0; iw < KW; ++iw) {
                float ms[2];
                for (int layout = 0; layout < 2; ++layout) {
                    for (int rep = 0; rep < 5; ++rep) {
                        if (rep == 1) hipEventRecord(e0);
                        if (layout == 0) tile_kernel<false><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                        else tile_kernel<true><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms[layout], e0, e1);
                    ms[layout] /= 4;
                }
                printf("round %d Z%d W%d: colmajor %.4f ms   panel %.4f ms\n", rnd, iz, iw, ms[0], ms[1]);
            }
    return 0;
}
