// Micro-benchmark for the open question of FINDINGS.md 6: is the allocation-pair sensitivity of the fused projected-CG kernel (and its 0.81
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

template <bool PANEL, bool LOADV = true, bool STOREV = true>
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
        const double gg = LOADV ? g_in[row] : 1.0, dd = LOADV ? d[row] : 2.0, aa = LOADV ? a[row] : 3.0;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1), s);       // first product
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;                  // row update
        if (STOREV && h == 0) g_out[row] = gp;
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);   // second product (no cross-lane reduction: traffic only)
    }
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red;
    if (tot == 123.456) part[blockIdx.x] = tot;
}

// ---- how the 80 MB store stream is issued (column-major walk, vector loads on) ----------------------------------------------------------
//   SMODE 0: one 128-byte store per tile by the 16 owner lanes (what the library's kernels do)
//   SMODE 1: lane group h keeps the value of tile k = h (mod 4); every fourth tile all 64 lanes store: one instruction, four 128-byte pieces
//   SMODE 2: the four waves of a workgroup meet in LDS once per 64-row round (one barrier); wave (k & 3) stores the round's 512 contiguous bytes
//   SMODE 3: every wave walks its OWN contiguous quarter of the workgroup's span, 512 contiguous bytes stored every fourth tile
//   SMODE 4: stores go to a 64 KB window (cache-resident): what the kernel costs when the store stream never reaches memory
//   SMODE 8 / 32 / 128: that many rounds staged in LDS, then stored in one burst by the whole workgroup (workgroups run in near lockstep: bursts align)
template <int SMODE>
__global__ __launch_bounds__(256) void store_kernel(const double* __restrict__ M, int64_t ld, int64_t rounds, const double* __restrict__ g_in,
                                                    double* __restrict__ g_out, const double* __restrict__ d, const double* __restrict__ a,
                                                    double* __restrict__ part) {
    constexpr int BURST = SMODE >= 5 ? SMODE : 2;      // SMODE >= 5: rounds staged in LDS before one burst of stores
    __shared__ double stage[BURST][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem);
    const int cnt = (int)(q + (blockIdx.x < rem ? 1 : 0));
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    double s_red = 0.0, keep = 0.0;
    // SMODE 3: this wave's tiles are the 16-row tiles [wt0, wt0 + wcnt) of the span (4 * cnt tiles in all)
    const int64_t tiles = (int64_t)cnt * 4, wq = tiles / 4;
    const int64_t wt0 = t0 * 4 + wave * wq;
    const int wcnt = (int)(wave == 3 ? tiles - 3 * wq : wq);
    const int iters = SMODE == 3 ? wcnt : cnt;
    for (int k = 0; k < iters; ++k) {
        const int64_t row0 = SMODE == 3 ? (wt0 + k) * 16 : (t0 + k) * 64 + wave * 16;
        double av[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) av[c] = __builtin_nontemporal_load(M + row0 + r + (int64_t)(4 * c + h) * ld);
        const int64_t row = row0 + r;
        const double gg = g_in[row], dd = d[row], aa = a[row];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1), s);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;
        if (SMODE == 0) { if (h == 0) g_out[row] = gp; }
        if (SMODE == 4) { if (h == 0) g_out[row & 8191] = gp; }
        if (SMODE == 1) {
            if ((k & 3) == h) keep = gp;
            if ((k & 3) == 3) g_out[(t0 + (k - 3 + h)) * 64 + wave * 16 + r] = keep;          // tile k-3+h of this wave
        }
        if (SMODE == 3) {
            if ((k & 3) == h) keep = gp;
            if ((k & 3) == 3) g_out[(wt0 + (k - 3 + h)) * 16 + r] = keep;                      // 64 contiguous rows
        }
        if (SMODE >= 5) {                              // stage BURST rounds (BURST * 512 bytes) in LDS, then the whole workgroup stores them at once
            if (h == 0) stage[k % BURST][wave * 16 + r] = gp;
            if (k % BURST == BURST - 1 || k == iters - 1) {
                const int nr = k % BURST + 1;                                    // rounds staged (the last burst of a span may be short)
                __syncthreads();
                for (int e = threadIdx.x; e < nr * 64; e += 256) g_out[(t0 + k - (nr - 1)) * 64 + e] = stage[e / 64][e % 64];
                __syncthreads();
            }
        }
        if (SMODE == 2) {
            if (h == 0) stage[k & 1][wave * 16 + r] = gp;
            __syncthreads();
            if (wave == (k & 3)) g_out[(t0 + k) * 64 + lane] = stage[k & 1][lane];
        }
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);
    }
    // (the last < 4 tiles of SMODE 1 / 3 are not flushed: a timing probe)
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red + keep;
    if (tot == 123.456) part[blockIdx.x] = tot;
}

// A fifth wave does ALL the stores: the four tile waves hand their values over through an LDS ring (two batches of R rounds, one
// barrier per batch) and never issue a store themselves.  If what hurts is a store sitting in a tile wave's in-order memory queue
// (loads issued behind it cannot be waited for without waiting for the store's acknowledgement as well), this is free.
template <int R>
__global__ __launch_bounds__(320) void writer_kernel(const double* __restrict__ M, int64_t ld, int64_t rounds, const double* __restrict__ g_in,
                                                     double* __restrict__ g_out, const double* __restrict__ d, const double* __restrict__ a,
                                                     double* __restrict__ part) {
    __shared__ double ring[2][R][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;
    const int64_t q = rounds / gridDim.x, rem = rounds % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem);
    const int cnt = (int)(q + (blockIdx.x < rem ? 1 : 0));
    const int nb = (cnt + R - 1) / R;
    if (wave == 4) {
        for (int b = 0; b < nb; ++b) {
            __syncthreads();                                   // batch b is complete
            const int nr = (cnt - b * R) < R ? (cnt - b * R) : R;
            for (int e = lane; e < nr * 64; e += 64) g_out[(t0 + (int64_t)b * R) * 64 + e] = ring[b & 1][e / 64][e % 64];
        }
        return;
    }
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    double s_red = 0.0;
    for (int k = 0; k < cnt; ++k) {
        const int64_t row0 = (t0 + k) * 64 + wave * 16;
        double av[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) av[c] = __builtin_nontemporal_load(M + row0 + r + (int64_t)(4 * c + h) * ld);
        const int64_t row = row0 + r;
        const double gg = g_in[row], dd = d[row], aa = a[row];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1), s);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;
        if (h == 0) ring[(k / R) & 1][k % R][wave * 16 + r] = gp;
        if (k % R == R - 1 || k == cnt - 1) __syncthreads();   // hand the batch to the writer wave
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);
    }
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red;
    if (tot == 123.456) part[blockIdx.x] = tot;
}

// The four waves of a workgroup all load the SAME 16-row tile (each would then do the row update and the second product of a different trial
// point of a batched Newton step): does the matrix still come from memory once?  SYNC: a barrier every 8 tiles keeps the waves together.
template <bool SYNC>
__global__ __launch_bounds__(256) void redundant_kernel(const double* __restrict__ M, int64_t ld, int64_t rounds16, const double* __restrict__ g_in,
                                                        const double* __restrict__ d, const double* __restrict__ a, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;
    const int64_t q = rounds16 / gridDim.x, rem = rounds16 % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem);
    const int cnt = (int)(q + (blockIdx.x < rem ? 1 : 0));
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    double s_red = 0.0;
    for (int k = 0; k < cnt; ++k) {
        const int64_t row0 = (t0 + k) * 16;
        double av[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) av[c] = __builtin_nontemporal_load(M + row0 + r + (int64_t)(4 * c + h) * ld);
        const int64_t row = row0 + r;
        const double gg = g_in[row], dd = d[row], aa = a[row];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1 + wave), s);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);
        if (SYNC && (k & 7) == 7) __syncthreads();
    }
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red;
    if (tot == 123.456) part[blockIdx.x * 4 + wave] = tot;
}

// ... or through LDS: each wave loads a QUARTER of the 16-row tile (8 of the 32 column registers), the four quarters meet in LDS (two
// buffers, one barrier per tile), and every wave reads the whole tile back into registers for its own trial point.
__global__ __launch_bounds__(256) void shared_tile_kernel(const double* __restrict__ M, int64_t ld, int64_t rounds16, const double* __restrict__ g_in,
                                                          const double* __restrict__ d, const double* __restrict__ a, double* __restrict__ part) {
    __shared__ double tile[2][CPL][64];                       // [buffer][column register c][lane slot (r, h)]: 16 KB each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;
    const int64_t q = rounds16 / gridDim.x, rem = rounds16 % gridDim.x;
    const int64_t t0 = blockIdx.x * q + (blockIdx.x < rem ? blockIdx.x : rem);
    const int cnt = (int)(q + (blockIdx.x < rem ? 1 : 0));
    constexpr int CQ = CPL / 4;                               // column registers a wave fetches
    double acc2[4] = {0.0, 0.0, 0.0, 0.0};
    double s_red = 0.0;
    double nx[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) nx[c] = __builtin_nontemporal_load(M + t0 * 16 + r + (int64_t)(4 * (wave * CQ + c) + h) * ld);
    for (int k = 0; k < cnt; ++k) {
        const int b = k & 1;
#pragma unroll
        for (int c = 0; c < CQ; ++c) tile[b][wave * CQ + c][lane] = nx[c];
        if (k + 1 < cnt) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) nx[c] = __builtin_nontemporal_load(M + (t0 + k + 1) * 16 + r + (int64_t)(4 * (wave * CQ + c) + h) * ld);
        }
        __syncthreads();
        double av[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) av[c] = tile[b][c][lane];
        const int64_t row = (t0 + k) * 16 + r;
        const double gg = g_in[row], dd = d[row], aa = a[row];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) s = fma(av[c], 1e-3 * (c + 1 + wave), s);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        const double gp = fma(1e-9, aa * dd, gg) - 1e-12 * s;
        s_red += gp * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc2[c & 3] = fma(av[c], gp, acc2[c & 3]);
    }
    const double tot = (acc2[0] + acc2[1]) + (acc2[2] + acc2[3]) + s_red;
    if (tot == 123.456) part[blockIdx.x * 4 + wave] = tot;
}

// pseudo-random doubles in [0.5, 1.5) (every bit pattern different from its neighbours'), or zeros
__global__ void fill_kernel(double* p, int64_t n, unsigned long long seed, int zero) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
        p[i] = zero ? 0.0 : 0.5 + (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
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
    hipMalloc((void**)&part, 8 * 16384);
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
    // which component of the kernel makes a slow pair slow?  column-major walk; all four combinations of {vector loads, vector store}
    printf("\ncomponents (column-major): full | loads only (no store) | store only (no vector loads) | matrix stream alone\n");
    for (int iz = 0; iz < KZ; ++iz)
        for (int iw = 0; iw < KW; ++iw) {
            float ms[4];
            for (int v = 0; v < 4; ++v) {
                for (int rep = 0; rep < 5; ++rep) {
                    if (rep == 1) hipEventRecord(e0);
                    if (v == 0) tile_kernel<false, true, true><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 1) tile_kernel<false, true, false><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 2) tile_kernel<false, false, true><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 3) tile_kernel<false, false, false><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[v], e0, e1);
                ms[v] /= 4;
            }
            printf("Z%d W%d: %.4f | %.4f | %.4f | %.4f ms\n", iz, iw, ms[0], ms[1], ms[2], ms[3]);
        }
    printf("\nstore modes (column-major): per tile 128 B | every 4th tile 4 x 128 B | LDS + barrier, 512 B per round | own quarter span, 512 B | stores to a 64 KB window\n");
    for (int iz = 0; iz < KZ; ++iz)
        for (int iw = 0; iw < KW; iw += (KW > 1 ? KW - 1 : 1)) {
            float ms[13];
            for (int v = 0; v < 13; ++v) {
                for (int rep = 0; rep < 5; ++rep) {
                    if (rep == 1) hipEventRecord(e0);
                    if (v == 0) store_kernel<0><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 1) store_kernel<1><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 2) store_kernel<2><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 3) store_kernel<3><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 4) store_kernel<4><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 5) store_kernel<32><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 6) store_kernel<64><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 7) store_kernel<128><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    // what the fused kernel could afford: 3 workgroups per CU (register sums, no LDS sums), 52 KB of staging each, 2 bursts per span
                    if (v == 8) store_kernel<102><<<768, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 9) store_kernel<0><<<768, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 10) writer_kernel<4><<<grid, 320>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 11) writer_kernel<16><<<grid, 320>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                    if (v == 12) writer_kernel<16><<<768, 320>>>(Z[iz], ld, nround, G[iw], G[iw], D[iw], A[iw], part);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[v], e0, e1);
                ms[v] /= 4;
            }
            printf("Z%d W%d: %.4f | %.4f | %.4f | %.4f | %.4f | bursts of 32 / 64 / 128 (= the whole span) rounds: %.4f %.4f %.4f ms\n", iz, iw, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7]);
            printf("        768 workgroups: 2 bursts of 102 rounds %.4f | per tile %.4f ms\n", ms[8], ms[9]);
            printf("        a fifth wave stores (tile waves never do): batches of 4 rounds %.4f | 16 rounds %.4f | 16 rounds, 768 workgroups %.4f ms\n", ms[10], ms[11], ms[12]);
        }
    printf("\nfour waves of a workgroup load the SAME tile (a trial point each): free-running | a barrier every 8 tiles | four different tiles (no store)\n");
    for (int iz = 0; iz < KZ; ++iz) {
        float ms[4];
        for (int v = 0; v < 4; ++v) {
            for (int rep = 0; rep < 5; ++rep) {
                if (rep == 1) hipEventRecord(e0);
                if (v == 0) redundant_kernel<false><<<grid, 256>>>(Z[iz], ld, nround * 4, G[0], D[0], A[0], part);
                if (v == 1) redundant_kernel<true><<<grid, 256>>>(Z[iz], ld, nround * 4, G[0], D[0], A[0], part);
                if (v == 2) tile_kernel<false, true, false><<<grid, 256>>>(Z[iz], ld, nround, G[0], G[0], D[0], A[0], part);
                if (v == 3) shared_tile_kernel<<<grid, 256>>>(Z[iz], ld, nround * 4, G[0], D[0], A[0], part);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
            ms[v] /= 4;
        }
        printf("Z%d: %.4f | %.4f | %.4f ms   | the tile shared through LDS (a quarter loaded per wave): %.4f ms\n", iz, ms[0], ms[1], ms[2], ms[3]);
    }
    // Does the CONTENT of the vectors matter (the library's placement trials run on zeros and come out faster than the real kernel on slow
    // pairs)?  Same kernels, the three vectors zero or pseudo-random: g alone (what is loaded AND what is stored changes), d and a alone (the
    // stored values become non-zero, the loaded g stays zero), all three.
    printf("\nvector contents (Z0 W0): per-tile store | two bursts, 768 workgroups | loads only\n");
    {
        const char* names[4] = {"all zero", "g random", "d, a random (stores non-zero)", "g, d, a random"};
        for (int c = 0; c < 4; ++c) {
            fill_kernel<<<1024, 256>>>(G[0], npad, 1, !(c == 1 || c == 3));
            fill_kernel<<<1024, 256>>>(D[0], npad, 2, !(c >= 2));
            fill_kernel<<<1024, 256>>>(A[0], npad, 3, !(c >= 2));
            float ms[3];
            for (int v = 0; v < 3; ++v) {
                for (int rep = 0; rep < 5; ++rep) {
                    if (rep == 1) hipEventRecord(e0);
                    if (c == 1 || c == 3) { if (rep == 0) {} }
                    if (v == 0) store_kernel<0><<<grid, 256>>>(Z[0], ld, nround, G[0], G[0], D[0], A[0], part);
                    if (v == 1) store_kernel<102><<<768, 256>>>(Z[0], ld, nround, G[0], G[0], D[0], A[0], part);
                    if (v == 2) tile_kernel<false, true, false><<<grid, 256>>>(Z[0], ld, nround, G[0], G[0], D[0], A[0], part);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[v], e0, e1);
                ms[v] /= 4;
            }
            printf("%-32s %.4f | %.4f | %.4f ms\n", names[c], ms[0], ms[1], ms[2]);
        }
        fill_kernel<<<1024, 256>>>(G[0], npad, 1, 1); fill_kernel<<<1024, 256>>>(D[0], npad, 2, 1); fill_kernel<<<1024, 256>>>(A[0], npad, 3, 1);
    }
    // ... and does it follow the STORED vector alone?  g from set iw, d and a from set (iw + 1) % KW
    printf("\nresidual (loaded + stored) from set W, direction / diagonal from the NEXT set:\n");
    for (int iz = 0; iz < KZ; ++iz)
        for (int iw = 0; iw < KW; ++iw) {
            const int jw = (iw + 1) % KW;
            float t;
            for (int rep = 0; rep < 5; ++rep) {
                if (rep == 1) hipEventRecord(e0);
                tile_kernel<false, true, true><<<grid, 256>>>(Z[iz], ld, nround, G[iw], G[iw], D[jw], A[jw], part);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1);
            printf("Z%d g:W%d d,a:W%d: %.4f ms\n", iz, iw, jw, t / 4);
        }
    return 0;
}
