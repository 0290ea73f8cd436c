// Device-side building blocks of liblfpsqp_hip (gfx950 / CDNA4, wave64).
//
// Every big operation of the hot path is one of four streaming shapes over the
// tall-skinny column-major layout (FINDINGS.md §4, §5):
//   onepass_kernel: y = M t, row-local update, then sums of M' v  -- ONE pass, tile held in registers
//                   (the projected-CG iteration, the Newton-retraction step, the pcg! iteration)
//   gemv_t_kernel : t = M' v      -- v produced on the fly by a functor (fused vector updates)
//   gemv_n_kernel : y = M t       -- consumed on the fly by a functor (fused updates + dot partials)
//   vec_kernel    : elementwise map + up to 8 sum/max reductions
// plus reduce_rows_kernel, the fixed-order second stage of every reduction.
//
// Two-pass kernels: a thread owns KS double2 row-pairs of a 512*KS-row tile and streams them over all
// columns: lanes read consecutive 16-byte pieces (global_load_dwordx4, 1 KiB per wave
// instruction), >= 8 loads are in flight per lane, nothing is staged through
// LDS (each matrix byte is used exactly once -- guide: "GEMV / M<=16: load straight to
// VGPRs, deep unroll").  All reductions are two-stage and atomics-free so results are
// bit-reproducible for a given (n_loc, m) on a given device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace lfpsqp {

// Tuning knobs are template parameters selected at run time per context (lfpsqp_ctx_set_tuning):
//   KS  = 16-byte row pairs per lane (tile = 512*KS rows), NT = non-temporal matrix loads.
// All buffers are padded to kPadRows so every variant is valid on the same allocation.
#ifndef LFPSQP_TCOLS
#define LFPSQP_TCOLS 4
#endif

constexpr int kThreads = 256;                 // 4 waves
constexpr int kWaves = kThreads / 64;
constexpr int kSlabRows = kThreads * 2;       // 512 rows: one 16-byte piece per lane
constexpr int kMaxKS = 4;
constexpr int kPadRows = kSlabRows * kMaxKS;  // 2048: padding / shard-boundary granularity
constexpr int kTC = LFPSQP_TCOLS;             // columns reduced together in gemv_t
constexpr int kColChunk = 256;                // columns reduced per LDS flush
constexpr int kMaxRed = 8;                    // scalar reductions per kernel (row stride of the scalar partials)
constexpr int kOnepassRound = 64;             // rows per tile round of onepass_kernel (4 waves x 16 rows); 16 in its wide form
constexpr int kOnepassMaxCols = 1024;         // widest matrix of the one-pass kernels (wide form: 4 waves x 64 groups x 4)

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
// matrix stream: every byte is read exactly once per pass, so it may bypass cache retention
template <bool NT>
__device__ __forceinline__ double2 ldm(const double* p) {
    if (NT) return make_double2(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1));
    return ld2(p);
}

// Solver scalars / status words are written by one kernel (a reduction's post-op) and read,
// at a wave-uniform address, by every workgroup of the following kernels in the stream.
// A plain load of such a word compiles to s_load (scalar cache), and on MI355X / ROCm 7.2
// that path was observed to return the PREVIOUS kernel generation's value in roughly half
// of all processes (FINDINGS.md §6 "stale scalar-cache reads"); agent-scope relaxed atomic
// loads (global_load ... sc1) always see the value.  Every cross-kernel scalar read goes
// through these two helpers.
__device__ __forceinline__ double ld_scal(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int64_t ld_stat(const int64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 8-byte load through a buffer descriptor: wave-uniform 64-bit base (scalar registers) + 32-bit per-lane byte
// offset.  Unlike a flat/global load with a 64-bit per-lane address this costs no address registers, which is what
// lets a kernel keep a whole matrix tile in registers (retract.hip).  NT = non-temporal (streamed once).
template <bool NT>
__device__ __forceinline__ double buf_load_f64(const void* uniform_base, uint32_t lane_byte_off) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), 0, 0xffffffff, 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_byte_off, 0, NT ? 2 : 0));
}
// a wave-uniform pointer forced into scalar registers (where the compiler, short of scalar registers, parks a uniform 64-bit base in vector
// registers, every buffer load through it becomes a readfirstlane LOOP; two v_readfirstlane instead)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
#ifdef LFPSQP_HIP_EMULATED
    return p;
#else
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
#endif
}
// compiler-only ordering point for memory operations (no instruction is emitted)
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// a value the optimiser may not trace back to where it came from (no instruction is emitted): keeps a small register array that is later
// SELECTED from by a lane-dependent index from being demoted to scratch memory as a dynamically indexed array
__device__ __forceinline__ double opaque_f64(double x) {
#ifndef LFPSQP_HIP_EMULATED
    asm volatile("" : "+v"(x));
#endif
    return x;
}
__device__ __forceinline__ void st2(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }
// streaming store: the line is not kept dirty in the L2 for a later write-back (which would then fall into the NEXT kernel's read stream)
__device__ __forceinline__ void st2_stream(double* p, double2 v) {
#ifdef LFPSQP_HIP_EMULATED
    st2(p, v);
#else
    __builtin_nontemporal_store(v.x, p);
    __builtin_nontemporal_store(v.y, p + 1);
#endif
}

// max that PROPAGATES NaN like Julia's max / norm(v, Inf) (fmax would drop it): a NaN constraint
// value must read as "not converged" (reference src/retractions.jl:135, src/optimize.jl:320).
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmax(a, b)); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o));
    return v;
}

// Block reduction of NRED per-thread values; thread 0 gets the result and writes
// dst[k].  `ismax` bit k selects max instead of sum for slot k.  Fixed order.
template <int NRED>
__device__ __forceinline__ void block_reduce_store(double (&red)[NRED], unsigned ismax, double* dst) {
    __shared__ double sm[kWaves][kMaxRed];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
        double r = ((ismax >> k) & 1u) ? wave_max(red[k]) : wave_sum(red[k]);
        if (lane == 0) sm[wave][k] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) {
            if ((ismax >> k) & 1u)
                dst[k] = nanmax(nanmax(sm[0][k], sm[1][k]), nanmax(sm[2][k], sm[3][k]));
            else
                dst[k] = (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]);
        }
    }
}

// The same for running sums SHARED by the four lane groups H = lane bits 3..2 (onepass_kernel, EP::kSplitRed): accumulator
// s of a lane in group h is logical sum 4*s + h.  Sums over the lanes of a group in the order wave_sum uses (xor 32, 16, 2,
// 1), then over the waves; dst[0 .. NLOG).
template <int NRL, int NLOG>
__device__ __forceinline__ void block_reduce_store_split(double (&red)[NRL], double* dst) {
    __shared__ double sm[kWaves][4 * NRL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NRL; ++s) {
        double r = red[s];
        r += __shfl_xor(r, 32);
        r += __shfl_xor(r, 16);
        r += __shfl_xor(r, 2);
        r += __shfl_xor(r, 1);
        if ((lane & 0x33) == 0) sm[wave][4 * s + (lane >> 2)] = r;
    }
    __syncthreads();
    if (threadIdx.x < NLOG) dst[threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------
// Matrix VIEWS (lfpsqp_mat::view, lfpsqp_mat_view): the matrix the solvers work with is
//     diag(rs) * M + u w'          (rs, u: n-vectors; w: m-vector; each part optional)
// with M constant -- the constraint gradients Jct(x) = diag(phi'(x)) A + 2 x qw' of the nonlinear class lfpsqp_elementwise, which then never
// exist in memory (the reference's jac! rewrites the whole n x m matrix every outer iteration, src/autodiff_generators.jl:60-66; here it
// rewrites rs and u).  The kernels stay as they are: the launch helpers (internal.h) wrap the row functor.  First products arrive as
// rs_i (M t)_i + u_i (w't) with the scalar w't from a one-workgroup kernel ahead of the launch; second-product vectors leave scaled by rs,
// and sum_i u_i v_i travels as one more reduction term, folded into the column sums (+ w_j times it) by a one-workgroup kernel behind the
// launch.  Cost: 8 or 16 bytes per row next to 8 m, and two tiny launches.  Rows >= n carry zeros whatever the padding of rs / u holds.
// ---------------------------------------------------------------------------
template <class T, class = void>
struct is_rowscaled : std::false_type {};
template <class T>
struct is_rowscaled<T, std::void_t<decltype(T::kRowScaled)>> : std::true_type {};
// functors that refuse a view (the launch helper reports LFPSQP_ERR_UNSUPPORTED): EP::kNoRowScale
template <class T, class = void>
struct no_rowscale : std::false_type {};
template <class T>
struct no_rowscale<T, std::void_t<decltype(T::kNoRowScale)>> : std::true_type {};
struct ViewD {
    const double* rs;    // row scales or nullptr (= 1)
    const double* u;     // rank-one term u w' or nullptr
    const double* tau;   // device scalar w't of the launch's first product (u != nullptr)
    __device__ __forceinline__ double2 s2(int64_t r) const { return rs ? *reinterpret_cast<const double2*>(rs + r) : make_double2(1.0, 1.0); }
    __device__ __forceinline__ double2 u2(int64_t r) const { return u ? *reinterpret_cast<const double2*>(u + r) : make_double2(0.0, 0.0); }
};

// ---------------------------------------------------------------------------
// GEMV-T: part[tile][j] = sum over the tile's rows of M[row, j] * v[row]
//   VP::load(row, valid0, valid1) returns (v[row], v[row+1]) and may store fused
//   side outputs; rows >= n must return 0 (the tile is padded, matrix padding is 0).
//   VP::skip() (uniform) makes the whole launch a no-op (solver already finished).
// grid.x = number of tiles.  part has leading dimension part_ld >= ncols.
// ---------------------------------------------------------------------------
template <class VP, int kS, bool NT>
__global__ __launch_bounds__(kThreads) void gemv_t_kernel(const double* __restrict__ M, int64_t ld, int ncols, int64_t n,
                                                           VP vp, double* __restrict__ part, int part_ld) {
    if (vp.skip()) return;
    __shared__ double red[kWaves][kColChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * (kSlabRows * kS) + (int64_t)threadIdx.x * 2;
    double2 v[kS];
    double xsum = 0.0;                  // a view's rank-one term: this lane's part of sum_i u_i v_i -> part[tile][ncols]
#pragma unroll
    for (int s = 0; s < kS; ++s) {
        const int64_t r = row0 + (int64_t)s * kSlabRows;
        if constexpr (is_rowscaled<VP>::value) v[s] = vp.load_view(r, r < n, r + 1 < n, xsum);
        else v[s] = vp.load(r, r < n, r + 1 < n);
    }
    const double* base = M + row0;
    double* prow = part + (int64_t)blockIdx.x * part_ld;
    if constexpr (is_rowscaled<VP>::value) {
        if (vp.vw.u) {                  // (uniform)
            xsum = wave_sum(xsum);
            if (lane == 0) red[wave][0] = xsum;
            __syncthreads();
            if (threadIdx.x == 0) prow[ncols] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            __syncthreads();
        }
    }
    for (int j0 = 0; j0 < ncols; j0 += kColChunk) {
        const int jn = (ncols - j0 < kColChunk) ? (ncols - j0) : kColChunk;
        for (int j = 0; j < jn; j += kTC) {
            double p[kTC];
#pragma unroll
            for (int c = 0; c < kTC; ++c) {
                const int jj = (j + c < jn) ? (j + c) : (jn - 1);   // clamp: ragged last group re-reads a valid column
                const double* col = base + (int64_t)(j0 + jj) * ld;
                double acc = 0.0;
#pragma unroll
                for (int s = 0; s < kS; ++s) {
                    const double2 a = ldm<NT>(col + (int64_t)s * kSlabRows);
                    acc = fma(a.x, v[s].x, acc);
                    acc = fma(a.y, v[s].y, acc);
                }
                p[c] = acc;
            }
#pragma unroll
            for (int c = 0; c < kTC; ++c) p[c] = wave_sum(p[c]);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < kTC; ++c)
                    if (j + c < jn) red[wave][j + c] = p[c];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < jn; j += kThreads)
            prow[j0 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// GEMV-N: acc[row] = sum_j M[row, j] * t[j], handed to EP::apply(row, acc0, acc1,
// valid0, valid1, red) which does the fused stores and adds its reduction terms
// into red[0..NRED).  part[tile][k] receives the tile's reduction partials.
// ---------------------------------------------------------------------------
template <class EP, int NRED, int kS, bool NT>
__global__ __launch_bounds__(kThreads) void gemv_n_kernel(const double* __restrict__ M, int64_t ld, int ncols, int64_t n,
                                                           const double* __restrict__ t, EP ep, double* __restrict__ part) {
    if (ep.skip()) return;
    __shared__ double ts[kColChunk];
    const int64_t row0 = (int64_t)blockIdx.x * (kSlabRows * kS) + (int64_t)threadIdx.x * 2;
    const double* base = M + row0;
    double2 acc[kS];
#pragma unroll
    for (int s = 0; s < kS; ++s) acc[s] = make_double2(0.0, 0.0);
    for (int j0 = 0; j0 < ncols; j0 += kColChunk) {
        const int jn = (ncols - j0 < kColChunk) ? (ncols - j0) : kColChunk;
        __syncthreads();
        for (int j = threadIdx.x; j < jn; j += kThreads) ts[j] = t[j0 + j];
        __syncthreads();
        int j = 0;
        constexpr int kCU = (kS >= 4) ? 4 : 8;     // columns in flight: 16 x 16-byte loads per lane either way
        for (; j + kCU <= jn; j += kCU) {
            double2 a[kCU][kS];
#pragma unroll
            for (int c = 0; c < kCU; ++c) {
                const double* col = base + (int64_t)(j0 + j + c) * ld;
#pragma unroll
                for (int s = 0; s < kS; ++s) a[c][s] = ldm<NT>(col + (int64_t)s * kSlabRows);
            }
#pragma unroll
            for (int c = 0; c < kCU; ++c) {
                const double tj = ts[j + c];
#pragma unroll
                for (int s = 0; s < kS; ++s) {
                    acc[s].x = fma(a[c][s].x, tj, acc[s].x);
                    acc[s].y = fma(a[c][s].y, tj, acc[s].y);
                }
            }
        }
        for (; j < jn; ++j) {
            const double* col = base + (int64_t)(j0 + j) * ld;
            const double tj = ts[j];
#pragma unroll
            for (int s = 0; s < kS; ++s) {
                const double2 a = ldm<NT>(col + (int64_t)s * kSlabRows);
                acc[s].x = fma(a.x, tj, acc[s].x);
                acc[s].y = fma(a.y, tj, acc[s].y);
            }
        }
    }
    double red[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int k = 0; k < (NRED > 0 ? NRED : 1); ++k) red[k] = 0.0;
#pragma unroll
    for (int s = 0; s < kS; ++s) {
        const int64_t r = row0 + (int64_t)s * kSlabRows;
        ep.apply(r, acc[s], r < n, r + 1 < n, red);
    }
    if (NRED > 0) block_reduce_store<(NRED > 0 ? NRED : 1)>(red, 0u, part + (int64_t)blockIdx.x * kMaxRed);
}

// ---------------------------------------------------------------------------
// GEMV-N followed by GEMV-T on the SAME rows in one launch (the Newton-retraction step,
// reference src/retractions.jl:140-149):  acc = M1[rows, :n1] * t  ->  EP::apply(row, acc, ...)
// updates the iterate in registers, stores it once, returns v[row] and adds NRED reduction
// terms; then part[tile][j] = sum_rows M2[row, j] * v[row] for j < n2 and
// part[tile][n2 .. n2+NRED) = the block's reduction terms.  The intermediate vector never
// goes back to HBM between the two products.
// ---------------------------------------------------------------------------
template <class EP, int NRED, int kS, bool NT>
__global__ __launch_bounds__(kThreads) void gemv_nt_kernel(const double* __restrict__ M1, int64_t ld1, int n1,
                                                            const double* __restrict__ t, const double* __restrict__ M2, int64_t ld2_,
                                                            int n2, int64_t n, EP ep, double* __restrict__ part, int part_ld) {
    if (ep.skip()) return;
    __shared__ double ts[kColChunk];
    __shared__ double red[kWaves][kColChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * (kSlabRows * kS) + (int64_t)threadIdx.x * 2;
    double2 acc[kS];
#pragma unroll
    for (int s = 0; s < kS; ++s) acc[s] = make_double2(0.0, 0.0);
    {
        const double* base = M1 + row0;
        for (int j0 = 0; j0 < n1; j0 += kColChunk) {
            const int jn = (n1 - j0 < kColChunk) ? (n1 - j0) : kColChunk;
            __syncthreads();
            for (int j = threadIdx.x; j < jn; j += kThreads) ts[j] = t[j0 + j];
            __syncthreads();
            constexpr int kCU = (kS >= 4) ? 4 : 8;
            int j = 0;
            for (; j + kCU <= jn; j += kCU) {
                double2 a[kCU][kS];
#pragma unroll
                for (int c = 0; c < kCU; ++c)
#pragma unroll
                    for (int s = 0; s < kS; ++s) a[c][s] = ldm<NT>(base + (int64_t)(j0 + j + c) * ld1 + (int64_t)s * kSlabRows);
#pragma unroll
                for (int c = 0; c < kCU; ++c) {
                    const double tj = ts[j + c];
#pragma unroll
                    for (int s = 0; s < kS; ++s) {
                        acc[s].x = fma(a[c][s].x, tj, acc[s].x);
                        acc[s].y = fma(a[c][s].y, tj, acc[s].y);
                    }
                }
            }
            for (; j < jn; ++j) {
                const double tj = ts[j];
#pragma unroll
                for (int s = 0; s < kS; ++s) {
                    const double2 a = ldm<NT>(base + (int64_t)(j0 + j) * ld1 + (int64_t)s * kSlabRows);
                    acc[s].x = fma(a.x, tj, acc[s].x);
                    acc[s].y = fma(a.y, tj, acc[s].y);
                }
            }
        }
    }
    double rsum[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int k = 0; k < (NRED > 0 ? NRED : 1); ++k) rsum[k] = 0.0;
    double2 v[kS];
#pragma unroll
    for (int s = 0; s < kS; ++s) {
        const int64_t r = row0 + (int64_t)s * kSlabRows;
        v[s] = ep.apply(r, acc[s], r < n, r + 1 < n, rsum);
    }
    const double* base2 = M2 + row0;
    double* prow = part + (int64_t)blockIdx.x * part_ld;
    for (int j0 = 0; j0 < n2; j0 += kColChunk) {
        const int jn = (n2 - j0 < kColChunk) ? (n2 - j0) : kColChunk;
        for (int j = 0; j < jn; j += kTC) {
            double p[kTC];
#pragma unroll
            for (int c = 0; c < kTC; ++c) {
                const int jj = (j + c < jn) ? (j + c) : (jn - 1);
                const double* col = base2 + (int64_t)(j0 + jj) * ld2_;
                double a0 = 0.0;
#pragma unroll
                for (int s = 0; s < kS; ++s) {
                    const double2 a = ldm<NT>(col + (int64_t)s * kSlabRows);
                    a0 = fma(a.x, v[s].x, a0);
                    a0 = fma(a.y, v[s].y, a0);
                }
                p[c] = a0;
            }
#pragma unroll
            for (int c = 0; c < kTC; ++c) p[c] = wave_sum(p[c]);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < kTC; ++c)
                    if (j + c < jn) red[wave][j + c] = p[c];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < jn; j += kThreads) prow[j0 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        __syncthreads();
    }
    if (NRED > 0) block_reduce_store<(NRED > 0 ? NRED : 1)>(rsum, 0u, prow + n2);
}

// ---------------------------------------------------------------------------
// One-stream GEMV-N -> GEMV-T over the SAME matrix (onepass_kernel).  For y[row] = M[row, :ncN] . t followed by
// sums over rows of M[row, j] * v_k[row] (k < NV vectors produced from y by the row functor), a wave keeps a
// 16-row x (4*CPL)-column tile of M in registers between the two products, so M is streamed from HBM once instead
// of twice.  No barrier per tile (two per burst of staged stores, STG below); the next tile's loads are issued while the current
// tile's second product runs.  Loads go through a buffer descriptor (uniform column base + one 32-bit lane
// offset), so the tile costs no address registers.
//   Lane layout (lane bits 5..0 = R R H H r r): row = 4*RR + rr of the tile, column group H; register c of a lane is
// column 4c + H.  One wave instruction reads, per column group, the 16 rows of one 128-byte line (measured: 128-byte
// segments stream at the same rate as 512-byte ones, 64-byte ones at 70 %, tools/micro/segprobe.hip).
//   The second product does NOT keep one accumulator per (lane, column): four columns at a time, the products are
// summed over the row bits RR with two transposing lane swaps (v_permlane32_swap, v_permlane16_swap -- each swap
// halves the number of live values), so a lane accumulates ceil(CPL/4) values per vector instead of CPL.  That is
// what lets 2-3 waves per SIMD (enough loads in flight to saturate HBM) coexist with a register-resident tile.
//   The grid is persistent: (resident workgroups per CU) x (CUs) workgroups, each a contiguous balanced span of 64-row
// tile rounds (interleaved over its 4 waves), each emitting ONE partial row:
//   part[wg][k*ncT + j] (k < NV, j < ncT), then part[wg][NV*ncT + r] (r < NRED).
// Users: the Newton-retraction step (retract.hip) and the fused projected-CG iteration (projcg.hip).
// Host guarantees ncN >= 4, 3*ld*8 + kPadRows*8 < 2^32, (n + kPadRows)*8 < 2^32 and 1 <= gridDim.x <= rounds = ceil(n/64).
// ---------------------------------------------------------------------------
// (lanes l, l^BIT) hold (x0, x1) each: returns, in the lanes with BIT clear, x0(l) + x0(l^BIT); with BIT set, x1(l) + x1(l^BIT)
template <int BIT>
__device__ __forceinline__ double swap_add(double x0, double x1) {
    const uint64_t u0 = __builtin_bit_cast(uint64_t, x0), u1 = __builtin_bit_cast(uint64_t, x1);
    uint32_t a_lo, a_hi, b_lo, b_hi;
    if (BIT == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)u0, (uint32_t)u1, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(u0 >> 32), (uint32_t)(u1 >> 32), false, false);
        a_lo = lo[0]; b_lo = lo[1]; a_hi = hi[0]; b_hi = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)u0, (uint32_t)u1, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(u0 >> 32), (uint32_t)(u1 >> 32), false, false);
        a_lo = lo[0]; b_lo = lo[1]; a_hi = hi[0]; b_hi = hi[1];
    }
    return __builtin_bit_cast(double, ((uint64_t)a_hi << 32) | a_lo) + __builtin_bit_cast(double, ((uint64_t)b_hi << 32) | b_lo);
}

// EXACT: ncN > 4*(CPL-1), i.e. the last column group is register CPL-1 (no run-time column-group select).
// WIDE (ncN up to 16*CPL columns): the four waves of a workgroup share ONE 16-row tile and split its columns (wave w owns
// column groups [w*CPL, (w+1)*CPL)); the first product's per-wave partial sums meet in LDS (one barrier per tile round,
// two buffers), the row update is computed redundantly by every wave (wave 0 stores), the second product stays per wave.
// NA > 1: NA coefficient vectors t[b*t_stride ...] (batched first product: NA independent right-hand sides share the pass).
// LACC: the second product's running sums live in LDS (one private 8-byte slot per lane and sum, ds_add_f64, no
// conflicts, no return value) instead of 2*NV*NQ registers -- at CPL = 32, NV = 2 that is the difference between 3 and 4
// waves per SIMD.  The adds happen in the order the register version makes them: same bits.
// PIPE: the next tile's loads of column registers 4j..4j+3 are issued right after their last use in the second product
// (group by group) instead of after the whole tile -- the registers are dead from there on, so the single-buffered tile
// gets up to a second product's worth of head start on its memory latency at no register cost.
//
// Row functor contract (all members required):
//   EP::skip()                       uniform: launch is a no-op
//   EP::Uni, EP::uniform()           wave-uniform inputs, read once per kernel (kept in scalar registers)
//   EP::Row, EP::fetch(o)            per-row inputs at byte offset o = row*8, fetched one tile ahead of their use
//   EP::kSplitRed                    the four lane groups H of a row (which all hold the same row values) SHARE the scalar
//                                    reductions: group h accumulates logical sum 4*s + h in its accumulator s, so a lane keeps
//                                    ceil(NRED/4) running sums instead of NRED
//   EP::apply(row, o, acc[NA], valid, owner, lead, uni, in, v[NV], red[NRL])
//       row update; `owner` lanes (one per row) store; `lead` = this wave's reductions count (wave 0 in the wide form)
#ifndef LFPSQP_OP_LACC
#define LFPSQP_OP_LACC 1
#endif
#ifndef LFPSQP_OP_PIPE
#define LFPSQP_OP_PIPE 1
#endif
// ROWLATE: the next tile's row inputs (g, d, a ... of the row functor) are requested AFTER the next tile's matrix loads instead of before
// them.  The vector cache returns a CU's loads in order (profiles/r03b_*: on slow allocation pairs TCP_LFIFO_STALL_CYCLES is +57 % at equal
// traffic -- head-of-line blocking), so a slow row-input request ahead of 32 matrix loads holds all of them back; behind them it holds
// nothing, and it is not needed before the first product (which waits for the matrix anyway) is done.
#ifndef LFPSQP_OP_ROWLATE
#define LFPSQP_OP_ROWLATE 0
#endif
constexpr bool kOpLacc = LFPSQP_OP_LACC != 0, kOpPipe = LFPSQP_OP_PIPE != 0, kOpRowLate = LFPSQP_OP_ROWLATE != 0;

struct NoUni {};
// a wave-uniform double forced into scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// running sum in LDS: *p += v (no return value; p is private to the lane)
__device__ __forceinline__ void lds_add_f64(double* p, double v) {
#ifdef LFPSQP_HIP_EMULATED
    *p += v;
#else
    (void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
#endif
}

// number of staged output vectors a row functor offers (EP::kStageStreams; absent = 0)
template <class EP, class = void>
struct stage_streams : std::integral_constant<int, 0> {};
template <class EP>
struct stage_streams<EP, std::void_t<decltype(EP::kStageStreams)>> : std::integral_constant<int, EP::kStageStreams> {};

// functors with batched first products (NA > 1) that want their second product's running sums in LDS (EP::kLaccAnyNA)
template <class EP, class = void>
struct lacc_any_na : std::false_type {};
template <class EP>
struct lacc_any_na<EP, std::void_t<decltype(EP::kLaccAnyNA)>> : std::true_type {};
// functors whose staged form takes several first products (EP::kStageAnyNA; the others: NA == 1 only)
template <class EP, class = void>
struct stage_any_na : std::false_type {};
template <class EP>
struct stage_any_na<EP, std::void_t<decltype(EP::kStageAnyNA)>> : std::true_type {};

// STG > 0 (functors with a staged output, EP::apply_staged / EP::stage_out): the row update's output vector is not stored tile by
// tile; up to STG rounds of it (STG * 512 bytes) wait in LDS and the whole workgroup stores them in one burst, in equal bursts
// over its span.  Why: on MI355X a thin store stream inside the matrix read stream costs far more than its bytes (an 80 MB
// stream inside 10.2 GB of reads: +0.11 ... +0.25 ms on 1.53, depending on where the buffers landed); the same stores issued
// in a few device-wide phases (the workgroups of the persistent grid advance in step) cost half of that or less
// (tools/micro/layoutprobe.hip, profiles/r03e_layoutprobe_store_modes.txt).  The staging area shares `buf` (next to LDS running sums, LACC, it is an array of its own).
// SW: waves per SIMD the kernel is compiled for (the register budget follows from it; 1 = the compiler's choice).
template <class EP, int NV, int NRED, int CPL, bool EXACT, bool WIDE, int NA = 1, bool LACC = false, int STG = 0, int SW = 1>
__global__ __launch_bounds__(kThreads, SW) void onepass_kernel(const double* __restrict__ M, int64_t ld, int ncN, int ncT, int64_t n,
                                                            int64_t rounds, const double* __restrict__ t, int t_stride, EP ep,
                                                            double* __restrict__ part, int part_ld, int stage_cap, int vspans) {
    if (ep.skip()) return;
    constexpr int CW = 4, RW = 16;                   // column groups per wave instruction, rows per wave tile
    constexpr int NW = WIDE ? kWaves : 1;            // waves sharing a row tile
    constexpr int kStep = WIDE ? RW : RW * kWaves;   // rows the workgroup advances per tile round
    constexpr int NC = CW * CPL * NW;
    constexpr int NQ = (CPL + 3) / 4;                // running sums per lane and vector
    constexpr bool kSplit = EP::kSplitRed;
    constexpr int NRL = kSplit ? (NRED + 3) / 4 : NRED;   // scalar running sums per lane
    constexpr bool kRowAhead = sizeof(typename EP::Row) <= 16 * sizeof(double);   // fetch the next tile's row inputs a tile ahead
    static_assert(!(EXACT && WIDE), "the exact variant exists for the narrow kernel only");
    // LDS: ts (first-product coefficients), accx (wide form: per-wave partial sums of the first product), and ONE buffer
    // that holds the running sums of the second product during the tile loop (LACC) and the per-wave column sums after it
    constexpr int kRedD = (WIDE ? 1 : kWaves) * NV * NC;
    constexpr int kAccD = LACC ? NV * NQ * kThreads : 0;
    constexpr int NS = stage_streams<EP>::value > 0 ? stage_streams<EP>::value : 1;   // staged output vectors (the stacked forms: two)
    constexpr int kStgD = NS * STG * kStep;
    // the staging area shares `buf` with the final column sums; next to LDS running sums (LACC) it is an array of its own
    constexpr int kBufD0 = kRedD > kAccD ? kRedD : (kAccD > 0 ? kAccD : 1);
    constexpr int kBufD = (!LACC && kStgD > kBufD0) ? kStgD : kBufD0;
    __shared__ double ts[NA][NC];
    __shared__ double buf[kBufD];
    __shared__ double stg_own[(LACC && STG > 0) ? kStgD : 1];
    double* const stg = (LACC && STG > 0) ? stg_own : buf;
    __shared__ double accx[WIDE ? 2 : 1][WIDE ? NA : 1][WIDE ? kWaves : 1][WIDE ? RW : 1];
    auto red = [&](int w, int qq, int sl) -> double& { return buf[(w * NV + qq) * NC + sl]; };
    // (the wave index as a SCALAR: in the wide form it selects the wave's column range, i.e. the base of every buffer load of the tile -- as
    // threadIdx.x >> 6 the compiler must assume a lane-dependent base and wraps each load in a readfirstlane loop: 66 such loops in the
    // 32-register wide instantiation, round 5)
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = (lane & 3) | ((lane >> 4) << 2), h = (lane >> 2) & 3;
    const int g0 = WIDE ? wave * CPL : 0;            // first column group of this wave
    // Column groups: group g < glast holds columns g*CW .. g*CW+CW-1; the last group is shifted back to the columns
    // ncN-CW .. ncN-1 (all valid; the ones it shares with group glast-1 get a zero coefficient), groups past it
    // re-read it with zero coefficients.  Slot s = g*CW + h of ts[] / red[] therefore maps to one column.
    const int glast = EXACT ? (CPL - 1) : (ncN - 1) / CW;
    const int lastc0 = ncN - CW;
    for (int j = threadIdx.x; j < NC; j += kThreads) {
        const int g = j / CW, hh = j - g * CW;
#pragma unroll
        for (int b = 0; b < NA; ++b) {
            const double* tb = t + (int64_t)b * t_stride;
            double v = 0.0;
            if (g < glast) v = ld_scal(tb + j);
            else if (g == glast && lastc0 + hh >= glast * CW) v = ld_scal(tb + lastc0 + hh);
            ts[b][j] = v;
        }
    }
    double* pl = buf + threadIdx.x;                  // LACC: this lane's running sum (qq, j) is pl[(qq*NQ + j) * kThreads]
    __syncthreads();
    const typename EP::Uni uni = ep.uniform();
    // Persistent grid (one launch fills the machine once): workgroup b owns the contiguous span of `cnt` tile rounds
    // (kStep rows each) starting at round t0, balanced to +-1 round -- no tail of half-empty scheduling waves, and only
    // gridDim.x partial rows for the second stage.  grid <= rounds, so cnt >= 1.
    // VIRTUAL SPANS (batched first products, NA > 1, vspans > 0): the rows are cut into `vspans` spans -- the grid of ANOTHER instantiation, the
    // single-trial Newton step's -- and this workgroup works through ceil(vspans / gridDim.x) consecutive ones, emitting one partial row per
    // span.  The running sums of every span are then the very sums the other kernel forms, the second stage sees the same partial rows in the
    // same order, and a trial of the batch gets bit for bit what it gets retracted alone (retract.hip, "exact batch").  Everywhere else a
    // workgroup has exactly one span (vb = blockIdx.x) and the loop below runs once.
    const int64_t nspan = (NA > 1 && vspans > 0) ? (int64_t)vspans : (int64_t)gridDim.x;
    const int64_t vper = (NA > 1) ? (nspan + gridDim.x - 1) / gridDim.x : 1;
    const int64_t vb0 = (NA > 1) ? (int64_t)blockIdx.x * vper : (int64_t)blockIdx.x;
    const int64_t vb1 = (NA > 1) ? ((vb0 + vper < nspan) ? vb0 + vper : nspan) : vb0 + 1;
    for (int64_t vb = vb0; vb < vb1; ++vb) {
    if (LACC) {                                      // (private slots: no barrier; a previous span's use of buf ended with one)
#pragma unroll
        for (int s = 0; s < NV * NQ; ++s) pl[s * kThreads] = 0.0;
    }
    const int64_t q = rounds / nspan, rem = rounds % nspan;
    // (Dealing the rounds round-robin instead -- workgroup b takes rounds b, b + grid, ..., so that the whole grid works on one
    // window of consecutive rows -- was measured 7 % SLOWER at n = 1e7, m = 128: 2.11 against 1.97 ms on the same box.)
    const int64_t t0 = vb * q + (vb < rem ? vb : rem);
    const int cnt = (int)(q + (vb < rem ? 1 : 0));
    const int64_t row0 = t0 * kStep;                                             // uniform
    const int lrow = WIDE ? r : wave * RW + r;                                   // row within the round
    const uint32_t vo = (uint32_t)(lrow * 8) + (uint32_t)((int64_t)h * ld * 8);  // lane offset: row, and column within the group
    const int64_t cs = (int64_t)CW * ld * 8;
    const char* Mb = reinterpret_cast<const char*>(M + row0);
    const int64_t first_off = (int64_t)g0 * cs;
    const int64_t last_off = (int64_t)lastc0 * ld * 8;
    // column registers [c0, c1) of tile round k
    auto load_cols = [&](double (&a)[CPL], int k, int c0, int c1) {
        const char* tb = Mb + (int64_t)k * (kStep * 8);                                  // wave-uniform
        const char* lastb = tb + last_off;
#pragma unroll
        for (int c = c0; c < c1; ++c)
            if constexpr (WIDE) a[c] = buf_load_f64<true>(uniform_ptr((g0 + c < glast) ? tb + first_off + (int64_t)c * cs : lastb), vo);
            else a[c] = buf_load_f64<true>((EXACT ? (c < CPL - 1) : (g0 + c < glast)) ? tb + first_off + (int64_t)c * cs : lastb, vo);
    };
    double a[CPL], p[LACC ? 1 : NV][LACC ? 1 : NQ];
    if (!LACC) {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int qq = 0; qq < NV; ++qq) p[LACC ? 0 : qq][LACC ? 0 : j] = 0.0;
    }
    load_cols(a, 0, 0, CPL);
    uint32_t ro = (uint32_t)((row0 + lrow) * 8);
    typename EP::Row in = ep.fetch(ro);
    double rsum[NRL > 0 ? NRL : 1];
#pragma unroll
    for (int qq = 0; qq < (NRL > 0 ? NRL : 1); ++qq) rsum[qq] = 0.0;
    // staged stores: `cnt` rounds in ceil(cnt / STG) bursts of equal length (the last one may be shorter)
    const int bmax = (stage_cap > 0 && stage_cap < STG) ? stage_cap : STG;       // (stage_cap: test hook, 0 in production)
    const int nburst = STG > 0 ? (cnt + bmax - 1) / bmax : 1;
    const int blen = (cnt + nburst - 1) / nburst;
    int sk = 0;                                      // rounds waiting in the staging area
    auto tile_step = [&](int k, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        compiler_fence();               // re-read ts[] from LDS every tile instead of pinning 2*CPL registers on it
        double acc[NA];
#pragma unroll
        for (int b = 0; b < NA; ++b) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) s = fma(a[c], ts[b][(g0 + c) * CW + h], s);
            s += __shfl_xor(s, 4);      // sum over the column groups H (commutative pairings: every lane of a row agrees)
            s += __shfl_xor(s, 8);
            acc[b] = s;
        }
        if constexpr (WIDE) {           // ... and over the four waves' column ranges (one barrier for all NA right-hand sides)
            const int pb = k & 1;
            if (h == 0) {
#pragma unroll
                for (int b = 0; b < NA; ++b) accx[pb][b][wave][r] = acc[b];
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NA; ++b) acc[b] = (accx[pb][b][0][r] + accx[pb][b][1][r]) + (accx[pb][b][2][r] + accx[pb][b][3][r]);
        }
        const int64_t row = row0 + lrow + (int64_t)k * kStep;
        typename EP::Row in_next = in;
        if (MORE && kRowAhead && !kOpRowLate) in_next = ep.fetch(ro + kStep * 8);
        double v[NV];
        const bool lead = !WIDE || wave == 0;
        if constexpr (STG > 0) ep.apply_staged(row, ro, acc, row < n, h == 0 && lead, lead, uni, in, v, rsum, stg + sk * kStep + lrow, STG * kStep);
        else ep.apply(row, ro, acc, row < n, h == 0 && lead, lead, uni, in, v, rsum);
        // second product: columns 4j .. 4j+3 of this lane's group, summed over the row bits RR by two transposing swaps;
        // afterwards the lane holds the 4-row sum of column register 4j + 2*bit4 + bit5
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int qq = 0; qq < NV; ++qq) {
                const double x0 = a[4 * j] * v[qq];
                const double x1 = (4 * j + 1 < CPL) ? a[(4 * j + 1 < CPL) ? 4 * j + 1 : 0] * v[qq] : 0.0;
                const double x2 = (4 * j + 2 < CPL) ? a[(4 * j + 2 < CPL) ? 4 * j + 2 : 0] * v[qq] : 0.0;
                const double x3 = (4 * j + 3 < CPL) ? a[(4 * j + 3 < CPL) ? 4 * j + 3 : 0] * v[qq] : 0.0;
                const double w01 = swap_add<32>(x0, x1);
                const double w23 = swap_add<32>(x2, x3);
                const double w = swap_add<16>(w01, w23);
                if (LACC) lds_add_f64(pl + (qq * NQ + j) * kThreads, w);
                else p[LACC ? 0 : qq][LACC ? 0 : j] += w;
            }
            if (kOpPipe && MORE) {      // these four column registers are dead now: start the next tile's loads of them
                compiler_fence();
                load_cols(a, k + 1, 4 * j, (4 * j + 4 < CPL) ? 4 * j + 4 : CPL);
            }
        }
        if (!kOpPipe) {
            compiler_fence();           // the next tile's loads reuse a[]: keep them below its last use (no second buffer)
            if (MORE) load_cols(a, k + 1, 0, CPL);
        }
        ro += kStep * 8;
        if (kRowAhead && kOpRowLate) { if (MORE) in = ep.fetch(ro); }
        else if (kRowAhead) in = in_next;
        else if (MORE) in = ep.fetch(ro);   // big row records (batched trials): fetched after this tile's use, no second copy live
        if constexpr (STG > 0) {
            if (++sk == blen || !MORE) {    // uniform: the workgroup stores the staged rounds (the next tile's loads are in flight meanwhile)
                __syncthreads();
                const int64_t rb = row0 + (int64_t)(k + 1 - sk) * kStep;
#pragma unroll
                for (int sv = 0; sv < NS; ++sv) {
                    double* outv = ep.stage_out(sv);            // (nullptr: a launch that stores nothing, e.g. an evaluation-only pass)
                    if (outv != nullptr)
                        for (int e = threadIdx.x; e < sk * kStep; e += kThreads)
                            if (rb + e < n) outv[rb + e] = stg[sv * (STG * kStep) + e];
                }
                __syncthreads();            // before the next round overwrites the area
                sk = 0;
            }
        }
    };
#pragma unroll 1
    for (int k = 0; k < cnt - 1; ++k) tile_step(k, std::true_type());
    tile_step(cnt - 1, std::false_type());
    const int creg = 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);     // which of the 4 column registers this lane accumulated
    double pfin[NV][NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int qq = 0; qq < NV; ++qq) pfin[qq][j] = LACC ? pl[(qq * NQ + j) * kThreads] : p[LACC ? 0 : qq][LACC ? 0 : j];
    if (LACC) __syncthreads();           // buf changes its role: running sums -> per-wave column sums (STG: the last burst ended with a barrier)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int qq = 0; qq < NV; ++qq) {
            double s = pfin[qq][j];
            s += __shfl_xor(s, 1);       // the remaining row bits rr
            s += __shfl_xor(s, 2);
            const int c = 4 * j + creg;
            if ((lane & 3) == 0 && c < CPL) red(WIDE ? 0 : wave, qq, (g0 + c) * CW + h) = s;
        }
    __syncthreads();
    double* prow = part + vb * part_ld;
    for (int j = threadIdx.x; j < NV * ncT; j += kThreads) {
        const int qq = j / ncT, col = j - qq * ncT;
        const int sl = (col < glast * CW) ? col : (glast * CW + (col - lastc0));       // slot holding column `col`
        constexpr int W1 = WIDE ? 0 : 1, W2 = WIDE ? 0 : 2, W3 = WIDE ? 0 : 3;   // (the wide form has a single row of slots)
        prow[j] = WIDE ? red(0, qq, sl) : (red(0, qq, sl) + red(W1, qq, sl)) + (red(W2, qq, sl) + red(W3, qq, sl));
    }
    if (NRED > 0) {
        if (kSplit) block_reduce_store_split<NRL, NRED>(rsum, prow + NV * ncT);
        else block_reduce_store<(NRL > 0 ? NRL : 1)>(rsum, 0u, prow + NV * ncT);
    }
    if (NA > 1) __syncthreads();         // the next span reuses buf[] and the reduction scratch
    }   // virtual spans
}

// ---- the functor wrappers of matrix views (see ViewD above) ---------------------------------------------------------------------------
template <class VP>
struct RsLoadV {      // GEMV-T producer
    static constexpr bool kRowScaled = true;
    VP vp;
    ViewD vw;
    __device__ __forceinline__ bool skip() const { return vp.skip(); }
    __device__ __forceinline__ double2 load_view(int64_t r, bool v0, bool v1, double& xsum) const {
        const double2 v = vp.load(r, v0, v1);
        const double2 s = vw.s2(r), u = vw.u2(r);
        if (v0) xsum = fma(u.x, v.x, xsum);
        if (v1) xsum = fma(u.y, v.y, xsum);
        return make_double2(v0 ? v.x * s.x : 0.0, v1 ? v.y * s.y : 0.0);
    }
};
template <class EP>
struct RsApplyE {     // GEMV-N consumer
    static constexpr bool kRowScaled = true;
    EP ep;
    ViewD vw;
    __device__ __forceinline__ bool skip() const { return ep.skip(); }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double* red) const {
        const double2 s = vw.s2(r), u = vw.u2(r);
        const double tau = vw.u ? ld_scal(vw.tau) : 0.0;
        ep.apply(r, make_double2(v0 ? fma(u.x, tau, acc.x * s.x) : 0.0, v1 ? fma(u.y, tau, acc.y * s.y) : 0.0), v0, v1, red);
    }
};
template <class EP, int NREDI>
struct RsStepE {      // GEMV-N -> GEMV-T in one launch (two matrices, each plain or a view); reduction term NREDI = sum_i u2_i v_i
    static constexpr bool kRowScaled = true;
    EP ep;
    ViewD v1w, v2w;
    bool view1, view2;
    __device__ __forceinline__ bool skip() const { return ep.skip(); }
    __device__ __forceinline__ double2 apply(int64_t r, double2 acc, bool v0, bool v1, double* red) const {
        if (view1) {
            const double2 s = v1w.s2(r), u = v1w.u2(r);
            const double tau = v1w.u ? ld_scal(v1w.tau) : 0.0;
            acc = make_double2(v0 ? fma(u.x, tau, acc.x * s.x) : 0.0, v1 ? fma(u.y, tau, acc.y * s.y) : 0.0);
        }
        double2 v = ep.apply(r, acc, v0, v1, red);
        if (view2) {
            const double2 s = v2w.s2(r), u = v2w.u2(r);
            double x = 0.0;
            if (v0) x = fma(u.x, v.x, x);
            if (v1) x = fma(u.y, v.y, x);
            red[NREDI] += x;
            v = make_double2(v0 ? v.x * s.x : 0.0, v1 ? v.y * s.y : 0.0);
        }
        return v;
    }
};
// one-stream kernel (onepass_kernel): the row's scale and rank-one entry travel with its other inputs, a tile ahead; NV more reduction terms
// (logical indices NREDI .. NREDI + NV - 1, in the functor's own convention: split over the lane groups or on the owner lanes)
template <class EP, int NV, int NREDI>
struct RsRowE {
    static constexpr bool kRowScaled = true;
    EP ep;
    ViewD vw;
    struct Uni { typename EP::Uni in; double tau; };
    struct Row { typename EP::Row in; double s, u; };
    static constexpr bool kSplitRed = EP::kSplitRed;
    static constexpr int NRLI = kSplitRed ? (NREDI + 3) / 4 : NREDI;          // the wrapped functor's running sums per lane
    __device__ __forceinline__ bool skip() const { return ep.skip(); }
    __device__ __forceinline__ Uni uniform() const { return Uni{ep.uniform(), vw.u ? uniform_f64(ld_scal(vw.tau)) : 0.0}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        return Row{ep.fetch(o), vw.rs ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(vw.rs) + o) : 1.0,
                   vw.u ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(vw.u) + o) : 0.0};
    }
    template <class RED>
    __device__ __forceinline__ void extra(const double (&v)[NV], double ui, bool valid, bool owner, bool lead, RED& red) const {
        if constexpr (kSplitRed) {
            const int h = (int)((threadIdx.x >> 2) & 3u);
#pragma unroll
            for (int q = 0; q < NV; ++q)
                if (valid && lead && h == ((NREDI + q) & 3)) red[(NREDI + q) >> 2] = fma(ui, v[q], red[(NREDI + q) >> 2]);
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q)
                if (valid && owner) red[NREDI + q] = fma(ui, v[q], red[NREDI + q]);
        }
    }
    template <class RED>
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool lead, const Uni& u, const Row& w,
                                          double (&v)[NV], RED& red) const {
        const double s = valid ? w.s : 0.0;
        const double a2[1] = {valid ? fma(w.u, u.tau, acc[0] * s) : 0.0};
        ep.apply(row, o, a2, valid, owner, lead, u.in, w.in, v, reinterpret_cast<double(&)[NRLI > 0 ? NRLI : 1]>(red));
        extra(v, w.u, valid, owner, lead, red);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] *= s;
    }
    template <class RED>
    __device__ __forceinline__ void apply_staged(int64_t row, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool lead, const Uni& u,
                                                 const Row& w, double (&v)[NV], RED& red, double* slot, int sstride) const {
        const double s = valid ? w.s : 0.0;
        const double a2[1] = {valid ? fma(w.u, u.tau, acc[0] * s) : 0.0};
        ep.apply_staged(row, o, a2, valid, owner, lead, u.in, w.in, v, reinterpret_cast<double(&)[NRLI > 0 ? NRLI : 1]>(red), slot, sstride);
        extra(v, w.u, valid, owner, lead, red);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] *= s;
    }
    __device__ __forceinline__ double* stage_out(int sv) const { return ep.stage_out(sv); }
};
// (the staged form is offered exactly when the wrapped functor offers it)
template <class EP, int NV, int NREDI>
struct stage_streams<RsRowE<EP, NV, NREDI>> : stage_streams<EP> {};

// tau = w . t[0:ncols) ahead of a first product over a view with a rank-one term; one workgroup, fixed order
template <int D = 0>
__global__ __launch_bounds__(256) void view_tau_kernel(const double* __restrict__ w, const double* __restrict__ t, int ncols, double* __restrict__ tau) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int j = threadIdx.x; j < ncols; j += 256) s = fma(w[j], t[j], s);
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *tau = sm[0];
}
// behind a second product over a view: out[q * nc + j] = in[q * nc + j] + w_j in[nv * nc + nred + q]  (j < nc, q < nv; w == nullptr: no
// rank-one term), out[nv * nc + r] = in[nv * nc + r] (r < nred)
template <int D = 0>
__global__ __launch_bounds__(256) void view_fold_kernel(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ w, int nv, int nc,
                                                        int nred) {
    for (int i = threadIdx.x; i < nv * nc + nred; i += 256) {
        double v = in[i];
        if (w && i < nv * nc) v = fma(w[i % nc], in[nv * nc + nred + i / nc], v);
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------
// Elementwise map + reductions.  F::apply(i, valid0, valid1, red) handles the
// row pair (i, i+1).  Block b handles the `tpb` consecutive 512-row tiles [b*tpb, (b+1)*tpb) in order: a static assignment
// and a fixed in-block order keep the sums reproducible.  part[block][k].
//   Why short-lived blocks (tools/micro/vecprobe.hip, profiles/r04a_vecprobe.txt): on MI355X a plain fp64 copy / triad over 3.2 GB
// vectors streams at 6.24 / 6.03 TB/s with ONE tile per block (781 250 blocks dispatched in order), 5.8 / 5.6 with two, 5.3 / 5.3
// with eight or more, and 4.7 / 4.9 TB/s with a persistent grid of 2048 blocks striding over the tiles -- whatever the number of
// 16-byte pieces a lane keeps in flight (1, 2, 4, 8: no difference) and whatever the temporal hints.  So kernels without
// reductions get one tile per block; kernels with reductions keep the number of partial rows bounded (the second stage reads them).
// ---------------------------------------------------------------------------
template <class F, int NRED>
__global__ __launch_bounds__(kThreads) void vec_kernel(F f, int64_t n, unsigned ismax, double* __restrict__ part, int tpb) {
    if (f.skip()) return;
    double red[NRED > 0 ? NRED : 1];
#pragma unroll
    for (int k = 0; k < (NRED > 0 ? NRED : 1); ++k) red[k] = 0.0;
    const int64_t ntiles = (n + kSlabRows - 1) / kSlabRows;
    int64_t tile = (int64_t)blockIdx.x * tpb;
    const int64_t tend = (tile + tpb < ntiles) ? tile + tpb : ntiles;
    for (; tile < tend; ++tile) {
        const int64_t i = tile * kSlabRows + (int64_t)threadIdx.x * 2;
        f.apply(i, i < n, i + 1 < n, red);
    }
    if (NRED > 0) block_reduce_store<(NRED > 0 ? NRED : 1)>(red, ismax, part + (int64_t)blockIdx.x * kMaxRed);
}

// ---------------------------------------------------------------------------
// Second stage: out[by][j] = sum (or max) over the rows [by*row_chunk, (by+1)*row_chunk) of
// part[row][j], fixed order.  A block of 1024 threads = cw columns x (1024/cw) row groups
// (cw = 2^cw_log2 <= 32: few columns => many row groups, so scalar reductions use the whole
// block); per-thread strided sums, then a fixed LDS tree over the groups.  grid = (ceil(ncols/cw),
// row blocks).  POST::run(out) is executed by one thread after the sums are visible when the
// launch has a single block (scalar reductions: computes alpha/beta/status ...).
// ---------------------------------------------------------------------------
template <class POST>
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const double* __restrict__ part, int64_t nrows, int ncols, int part_ld,
                                                            unsigned ismax, double* out, int out_ld, int64_t row_chunk, int cw_log2,
                                                            POST post) {
    if (post.skip()) return;
    __shared__ double sm[1024];
    const int cw = 1 << cw_log2, groups = 1024 >> cw_log2;
    const int c = threadIdx.x & (cw - 1), g = threadIdx.x >> cw_log2;
    const int col = blockIdx.x * cw + c;
    const bool mx = (ismax >> (col < 32 ? col : 31)) & 1u;
    const int64_t r_begin = (int64_t)blockIdx.y * row_chunk;
    const int64_t r_end = (r_begin + row_chunk < nrows) ? (r_begin + row_chunk) : nrows;
    double acc = 0.0;
    if (col < ncols) {
        const double* pc = part + col;
        if (mx) {
            for (int64_t r = r_begin + g; r < r_end; r += groups) acc = nanmax(acc, pc[r * part_ld]);
        } else {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int64_t r = r_begin + g;
            const int64_t st = groups;
            for (; r + 3 * st < r_end; r += 4 * st) {
                a0 += pc[r * part_ld];
                a1 += pc[(r + st) * part_ld];
                a2 += pc[(r + 2 * st) * part_ld];
                a3 += pc[(r + 3 * st) * part_ld];
            }
            for (; r < r_end; r += st) a0 += pc[r * part_ld];
            acc = (a0 + a1) + (a2 + a3);
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = groups >> 1; s > 0; s >>= 1) {
        if (g < s) {
            const double o = sm[threadIdx.x + s * cw];
            sm[threadIdx.x] = mx ? nanmax(sm[threadIdx.x], o) : (sm[threadIdx.x] + o);
        }
        __syncthreads();
    }
    if (g == 0 && col < ncols) out[(int64_t)blockIdx.y * out_ld + col] = sm[c];
    if (gridDim.x == 1 && gridDim.y == 1) {
        __syncthreads();
        if (threadIdx.x == 0) post.run(out);
    }
}

struct NoPost {
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void run(double*) const {}
};

// 1-thread launch used when an all-reduce sits between the reduction and its post-op
template <class POST>
__global__ void post_kernel(double* out, POST post) {
    if (post.skip()) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) post.run(out);
}

// Elementwise transforms of the nonlinear constraint class lfpsqp_elementwise (kind as a double code: 0: t, 1: sin t, 2: t^2)
__device__ __forceinline__ double ew_phi(double k, double t) { return k == 0.0 ? t : (k == 1.0 ? sin(t) : t * t); }
__device__ __forceinline__ double ew_phi1(double k, double t) { return k == 0.0 ? 1.0 : (k == 1.0 ? cos(t) : 2.0 * t); }
__device__ __forceinline__ double ew_phi2(double k, double t) { return k == 0.0 ? 0.0 : (k == 1.0 ? -sin(t) : 2.0); }

// splitmix64 finaliser as a hash of the flat index -> [-1, 1)   
__device__ __forceinline__ double hash_u(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 0x1.0p-52 - 1.0;
}

}  // namespace lfpsqp
