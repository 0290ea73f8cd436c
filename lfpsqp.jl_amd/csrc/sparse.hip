// Sparse constraint gradients (the reference's README to-do "sparse Jacobians", README.md:80).
//
// Jct (n_loc x m, the transposed constraint Jacobian) with a few nonzeros per ROW -- every variable takes part in a few
// constraints: banded / block-structured equality systems, the reference's own test system (test/test_retractions.jl:34-54:
// two nonzeros per constraint).  Both products the retractions make with it stream nnz*(8+4) bytes instead of 8*n*m:
//
//   SpMV-N  y = alpha * Jct * t + beta * y     row-local: ELL by rows (K = max nonzeros of a row; val[k][row], col[k][row],
//                                              coalesced over rows), t (m doubles) stays in cache
//   SpMV-T  t = Jct' * v                       CSC (nonzeros of a column = of a constraint, rows ascending), cut into chunks
//                                              of 8192 nonzeros: one workgroup per chunk, fixed-order sums -> a second kernel adds
//                                              the chunks of each column in order.  No atomics: bit-reproducible.
//
//   Gram    G = Jct' diag(w2) Jct              exact fixed-point accumulation of the K (K + 1) / 2 products of a row (sp_gram below): atomics whose
//                                              order cannot matter -- bit-reproducible, 13 x faster than the MFMA Gram of a dense copy
//   SpMM    Z = [Jct | dense extra columns] * W  the basis-forming product of the tangent setup (lfpsqp_factorize_sp): bound by writing the
//                                              dense basis, a third of the MFMA product's time
//
// Users: lfpsqp_constraints_eval (c! of linear equalities), lfpsqp_pcg (the inner solve of the default ProjPenalty retraction: per
// iteration two sparse products instead of a dense pass), lfpsqp_factorize_sp, and -- through the basis in factored form U = Jct W
// (sp_basis_small, sp_factored_gemv_t/_n below; projcg.hip, retract.hip, ineq.hip) -- the projected CG, the Newton step and the
// tangent projection, none of which then reads a dense n x m matrix.
#include <algorithm>
#include <numeric>
#include <vector>

#include "internal.h"
#include "sparse.h"

namespace lfpsqp {

constexpr int kSpChunk = 8192;

__global__ __launch_bounds__(kThreads) void spmv_t_chunk_kernel(const int64_t* __restrict__ chunk_beg, const int32_t* __restrict__ row,
                                                                 const double* __restrict__ val, const double* __restrict__ v,
                                                                 double* __restrict__ partial) {
    const int64_t e0 = chunk_beg[blockIdx.x], e1 = chunk_beg[blockIdx.x + 1];
    // Eight gathers (row index, then v[row]) in flight per lane, fixed association.  (Round 3 measured four against eight in flight: 0.130 ms
    // either way at n = 1e7, K = 4 -- the kernel is not latency-bound.  Rows ascend inside a column, so the gathers of a wave are coalesced and
    // every v[i] is read once per nonzero of its row: 12 + 8 bytes per nonzero = 800 MB in 0.125 ms, 6.4 TB/s through L2 -- the "0.53 of peak"
    // of the algorithmic-bytes accounting (12 nnz + 8 n) is the price of reading v K times, not idle bandwidth.)
    constexpr int U = 8;
    double a[U];
#pragma unroll
    for (int k = 0; k < U; ++k) a[k] = 0.0;
    int64_t e = e0 + threadIdx.x;
    for (; e + (U - 1) * kThreads < e1; e += U * kThreads) {
        int32_t r[U];
        double w[U];
#pragma unroll
        for (int k = 0; k < U; ++k) r[k] = row[e + k * kThreads];
#pragma unroll
        for (int k = 0; k < U; ++k) w[k] = val[e + k * kThreads];
#pragma unroll
        for (int k = 0; k < U; ++k) a[k] = fma(w[k], v[r[k]], a[k]);
    }
    for (; e < e1; e += kThreads) a[0] = fma(val[e], v[row[e]], a[0]);
    double red[1] = {((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))};
    block_reduce_store<1>(red, 0u, partial + blockIdx.x);
}
__global__ void spmv_t_final_kernel(const int32_t* __restrict__ col_chunk, const double* __restrict__ partial, int m, double* __restrict__ t) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int c = col_chunk[j]; c < col_chunk[j + 1]; ++c) s += partial[c];
    t[j] = s;
}

struct SpmvNF {   // y = alpha * (Jct t) + beta * y
    EllRows E;
    double* y;
    double alpha, beta;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 a = E.acc(i);
        double2 o;
        if (beta == 0.0) {
            o = make_double2(alpha * a.x, alpha * a.y);
        } else {
            const double2 yy = ld2(y + i);
            o = make_double2(fma(alpha, a.x, beta * yy.x), fma(alpha, a.y, beta * yy.y));
        }
        if (v1) st2(y + i, o);
        else if (v0) y[i] = o.x;
    }
};
// Out = [S | X] * W, the basis-forming product of the tangent setup when the constraint gradients are sparse: n*m outputs but only
// K (+nx) multiply-adds each, so the kernel is bound by WRITING the dense result.  A lane owns two consecutive rows (16-byte stores,
// lanes along the rows: every store instruction writes 1 KB of one output column), keeps their ELL entries in registers (KR of them;
// KR = 0: re-read per output column) and walks the columns of a panel of W that the workgroup holds in LDS (odd row stride: the
// gathers W[col_k, c] of a wave spread over the banks; neighbouring rows of a banded matrix hit the same word and broadcast).
constexpr int kSpmmThreads = 1024, kSpmmLds = 16384 + 1024;      // doubles of LDS for the panel of W
template <int KR>
__global__ __launch_bounds__(kSpmmThreads) void spmm_kernel(const double* __restrict__ val, const int32_t* __restrict__ col, int64_t ld, int K,
                                                             const double* __restrict__ X, int64_t ldx, int nx, int ms,
                                                             const double* __restrict__ W, int ldw, int r, int pw, double* __restrict__ Out,
                                                             int64_t ld_out, int64_t n) {
    __shared__ double Wl[kSpmmLds];
    const int kw = ms + nx, pws = pw | 1;
    const int c0 = (int)blockIdx.y * pw;
    const int pc = (r - c0 < pw) ? (r - c0) : pw;
    for (int idx = threadIdx.x; idx < kw * pc; idx += kSpmmThreads) {
        const int k = idx % kw, c = idx / kw;
        Wl[k * pws + c] = W[(int64_t)(c0 + c) * ldw + k];
    }
    __syncthreads();
    const int64_t ntiles = (n + 2 * kSpmmThreads - 1) / (2 * kSpmmThreads);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * (2 * kSpmmThreads) + 2 * (int64_t)threadIdx.x;
        if (i >= n) continue;                                   // (no barrier below)
        double2 v[KR > 0 ? KR : 1];
        int2 ci[KR > 0 ? KR : 1];
        if (KR > 0) {
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                v[k] = make_double2(0.0, 0.0);
                ci[k] = make_int2(0, 0);
                if (k < K) {                                    // (the ELL arrays are padded to whole tiles: row i + 1 exists, with value 0)
                    v[k] = ld2(val + (int64_t)k * ld + i);
                    ci[k] = *reinterpret_cast<const int2*>(col + (int64_t)k * ld + i);
                }
            }
        }
        double2 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = make_double2(0.0, 0.0);
            if (j < nx) {
                const double* px = X + (int64_t)j * ldx + i;
                x[j] = make_double2(px[0], (i + 1 < n) ? px[1] : 0.0);
            }
        }
        for (int c = 0; c < pc; ++c) {
            double2 a = make_double2(0.0, 0.0);
            if (KR > 0) {
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    a.x = fma(v[k].x, Wl[ci[k].x * pws + c], a.x);
                    a.y = fma(v[k].y, Wl[ci[k].y * pws + c], a.y);
                }
            } else {
                for (int k = 0; k < K; ++k) {
                    const double2 vv = ld2(val + (int64_t)k * ld + i);
                    const int2 cc = *reinterpret_cast<const int2*>(col + (int64_t)k * ld + i);
                    a.x = fma(vv.x, Wl[cc.x * pws + c], a.x);
                    a.y = fma(vv.y, Wl[cc.y * pws + c], a.y);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nx) {
                    const double w = Wl[(ms + j) * pws + c];
                    a.x = fma(x[j].x, w, a.x);
                    a.y = fma(x[j].y, w, a.y);
                }
            double* o = Out + (int64_t)(c0 + c) * ld_out + i;
            if (i + 1 < n) st2(o, a);
            else o[0] = a.x;
        }
    }
}

int spmm(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* X, int x0, int nx, const double* W_dev, int ldw, int r, lfpsqp_mat* Out) {
    if (r <= 0 || S->n == 0) return 0;
    const int kw = (int)S->m + nx;
    if (nx < 0 || nx > 4 || (nx > 0 && !X) || kw < 1 || kw > 16384 || Out->n != S->n || Out->m < r)
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "spmm: shape not covered (%d + %d rows of W, %d columns)", (int)S->m, nx, r);
    int pw = 16384 / kw;
    if (pw > r) pw = r;
    const int npan = (r + pw - 1) / pw;
    const int64_t ntiles = (S->n + 2 * kSpmmThreads - 1) / (2 * kSpmmThreads);
    const int64_t cus = ctx->num_cu > 0 ? ctx->num_cu : 64;
    const dim3 grid((unsigned)(ntiles < cus ? ntiles : cus), (unsigned)npan);
    const double* xp = nx > 0 ? X->p + (int64_t)x0 * X->ld : nullptr;
    const int64_t ldx = nx > 0 ? X->ld : 0;
#define LF_SPMM(KR_)                                                                                                                          \
    hipLaunchKernelGGL((spmm_kernel<KR_>), grid, dim3(kSpmmThreads), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, xp, ldx, nx, (int)S->m, \
                       W_dev, ldw, r, pw, Out->p, Out->ld, S->n)
    if (S->K <= 4) LF_SPMM(4);
    else if (S->K <= 8) LF_SPMM(8);
    else LF_SPMM(0);
#undef LF_SPMM
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

// t = W' tA (m entries; W is wm x m, column-major), and optionally u = W t (wm entries).  One workgroup; tA / t are replicated.
__global__ __launch_bounds__(256) void sp_basis_small_kernel(const double* __restrict__ W, int wm, int m, const double* tA, double* t_out,
                                                              double* u_out) {
    __shared__ double ts[kOnepassMaxCols];
    for (int j = threadIdx.x; j < m; j += 256) {
        const double* wj = W + (size_t)j * wm;
        double s = 0.0;
        for (int k = 0; k < wm; ++k) s = fma(wj[k], ld_scal(tA + k), s);
        ts[j] = s;
        t_out[j] = s;
    }
    __syncthreads();
    if (u_out)
        for (int k = threadIdx.x; k < wm; k += 256) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s = fma(W[(size_t)j * wm + k], ts[j], s);
            u_out[k] = s;
        }
}
int sp_basis_small(lfpsqp_ctx* ctx, const double* W_dev, int wm, int m, const double* tA, double* t_out, double* u_out) {
    hipLaunchKernelGGL(sp_basis_small_kernel, dim3(1), dim3(256), 0, ctx->stream, W_dev, wm, m, tA, t_out, u_out);
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

// The factored basis U = A W (A = [SA | X[:, SA.m:]], W host, wm x m) applied to a vector without a dense n x m matrix:
// t = U'v = W'(A'v)   and   y = alpha U t + beta y = alpha A (W t) + beta y.
struct SpPlainVecF {   // GEMV-T producer: the vector itself
    const double* v;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 a = ld2(v + r);
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};
struct SpFactoredNF {  // y = alpha * ([S | X] u) + beta * y
    EllRows E;
    const double* xcol;
    int64_t ldx;
    const double* ux;
    int nx;
    double* y;
    double alpha, beta;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        double2 a = E.acc(i);
        for (int j = 0; j < nx; ++j) {
            const double w = ld_scal(ux + j);
            const double2 c = ld2(xcol + (int64_t)j * ldx + i);
            a.x = fma(c.x, w, a.x);
            a.y = fma(c.y, w, a.y);
        }
        double2 o;
        if (beta == 0.0) {
            o = make_double2(alpha * a.x, alpha * a.y);
        } else {
            const double2 yy = ld2(y + i);
            o = make_double2(fma(alpha, a.x, beta * yy.x), fma(alpha, a.y, beta * yy.y));
        }
        if (v1) st2(y + i, o);
        else if (v0) y[i] = o.x;
    }
};
int factored_setup(lfpsqp_ctx* ctx, const lfpsqp_mat* A, const double* W_host, int m, double** dW, double** tA, double** uA) {
    const int wm = (int)A->m;
    LF_TRY(ensure_small(ctx, (size_t)wm * m + 2 * (size_t)wm + 64));
    *dW = ctx->small;
    *tA = *dW + (((size_t)wm * m + 1) & ~(size_t)1);
    *uA = *tA + ((wm + 1) & ~1);
    LF_HIP(ctx, hipMemcpyAsync(*dW, W_host, sizeof(double) * (size_t)wm * m, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));                 // W_host is caller-owned pageable memory
    return 0;
}
int sp_factored_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* SA, const lfpsqp_mat* A, const double* W_host, int m, const double* v, double* t_out) {
    double *dW, *tA, *uA;
    LF_TRY(factored_setup(ctx, A, W_host, m, &dW, &tA, &uA));
    const int wm = (int)A->m, nx = wm - (int)SA->m;
    LF_TRY(spmv_t(ctx, SA, v, tA));
    if (nx > 0) {
        lfpsqp_mat view = *A;
        view.p = A->p + (int64_t)SA->m * A->ld;
        view.m = nx;
        LF_TRY(run_gemv_t(ctx, &view, nx, A->n, SpPlainVecF{v}, tA + SA->m));
    }
    return sp_basis_small(ctx, dW, wm, m, tA, t_out, nullptr);
}
__global__ __launch_bounds__(256) void sp_w_times_t_kernel(const double* __restrict__ W, int wm, int m, const double* t, double* u_out) {
    for (int k = threadIdx.x; k < wm; k += 256) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) s = fma(W[(size_t)j * wm + k], ld_scal(t + j), s);
        u_out[k] = s;
    }
}
int factored_w_times_t(lfpsqp_ctx* ctx, const double* W_dev, int wm, int m, const double* t, double* u_out) {
    hipLaunchKernelGGL(sp_w_times_t_kernel, dim3(1), dim3(256), 0, ctx->stream, W_dev, wm, m, t, u_out);
    LF_LAUNCH_CHECK(ctx);
    return 0;
}
int sp_factored_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_spmat* SA, const lfpsqp_mat* A, const double* W_host, int m, double alpha, const double* t,
                       double beta, double* y) {
    double *dW, *tA, *uA;
    LF_TRY(factored_setup(ctx, A, W_host, m, &dW, &tA, &uA));
    const int wm = (int)A->m, nx = wm - (int)SA->m;
    hipLaunchKernelGGL(sp_w_times_t_kernel, dim3(1), dim3(256), 0, ctx->stream, dW, wm, m, t, uA);
    LF_LAUNCH_CHECK(ctx);
    const SpFactoredNF f{ell_rows(SA, uA), A->p + (int64_t)SA->m * A->ld, A->ld, uA + SA->m, nx, y, alpha, beta};
    return run_vec<SpFactoredNF, 0, NoPost>(ctx, A->n, f, 0u, nullptr, NoPost());
}

// dense[:, j] from the CSC arrays of column j (one entry per position: every element written once)
__global__ void sp_scatter_kernel(const int64_t* __restrict__ colptr, const int32_t* __restrict__ row, const double* __restrict__ val,
                                  double* __restrict__ dense, int64_t ld_dense) {
    const int j = blockIdx.y;
    for (int64_t e = colptr[j] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < colptr[j + 1]; e += (int64_t)gridDim.x * blockDim.x)
        dense[(int64_t)j * ld_dense + row[e]] = val[e];
}

// t_out[0:m) = S' v (global: all-reduced).  v must hold at least S->n entries.
int spmv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const double* v, double* t_out) {
    if (S->m == 0) return 0;
    if (S->nchunks > 0) {
        LF_TRY(ensure_part(ctx, (size_t)S->nchunks + 8));
        hipLaunchKernelGGL(spmv_t_chunk_kernel, dim3((unsigned)S->nchunks), dim3(kThreads), 0, ctx->stream, S->chunk_beg, S->csc_row, S->csc_val, v,
                           ctx->part);
        LF_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(spmv_t_final_kernel, dim3((unsigned)((S->m + 255) / 256)), dim3(256), 0, ctx->stream, S->col_chunk, ctx->part, (int)S->m, t_out);
    LF_LAUNCH_CHECK(ctx);
    return allreduce_dev(ctx, t_out, S->m);
}



// ---- Gram matrix from the nonzeros -----------------------------------------------------------------------------------------------------
// G = S' diag(w2) S: row i adds w2_i v_a v_b to G[c_a, c_b] for every pair of its nonzeros -- a scattered accumulation.  Floating-point
// atomics would make the result depend on the order of arrival.  Here the accumulation is EXACT, so the order cannot matter and the result is
// bit-reproducible: with the values scaled by a power of two per column (|v'| < 1: entrywise accuracy relative to the columns' own magnitudes,
// as a floating-point Gram matrix has it), |term| < 2^E (E from the measured max_i w2_i max_a v'_ia^2) and at most n terms per entry (one per row:
// duplicates are merged at creation),
// every term x is cut into two fixed-point limbs of b = min(62 - ceil(log2(n + 1)), 40) bits,
//     q1 = rint(x 2^(b-E)),      q2 = rint((x - q1 2^(E-b)) 2^(2b-E))        (power-of-two scalings and an exact remainder),
// which are summed as 64-bit integers -- LDS atomics inside a workgroup, plain integer sums over the workgroups -- without overflow or
// rounding; what is dropped is below 2^(E-2b-1) per term (b = 38 at n = 1e7: 2^-77 of the largest term).  One rounding at the end.
// A workgroup owns the rows of a slice and one 128 x 64 tile of G (two limbs = 128 KB of LDS); tiles below the diagonal are skipped, the
// mirror image is written by the reduction.
constexpr int kSgJ = 128, kSgK = 64, kSgTile = kSgJ * kSgK, kSgThreads = 1024, kSgRun = 4096, kSgLimbMax = 40;
// KR > 0 (rows of at most KR nonzeros): a lane walks the rows tid, tid + T, ... of its workgroup's contiguous slice and keeps the limb sums of
// the KR (KR + 1) / 2 pairs in REGISTERS while the column set of its rows stays the same (integer-valued doubles: exact up to 2^53, hence
// limbs of at most 40 bits and runs of at most 4096 rows); the LDS atomics happen when the column set changes.  In a banded or
// block-structured system thousands of consecutive rows share their columns, and the atomics -- 64 lanes on one address cost ~160 cycles
// of the CU's one LDS per instruction: 0.8 ms of the first version's 0.96 at n = 1e7, K = 4 -- all but vanish.  Scattered column sets
// degrade to one flush per row, the generic path (KR = 0: no registers, atomics per term).
// Can a block of kSpBox rows whose nonzeros lie in the columns [lo, hi] touch the tile (rows j0 .. j0+kSgJ-1) x (columns k0 .. k0+kSgK-1) of G?
// (rows_per_wg is a multiple of kSpBox, so a workgroup's slice is whole blocks; uniform over the workgroup)
__device__ __forceinline__ bool sg_box_hits(const int32_t* lo, const int32_t* hi, int64_t b0, int j0, int k0) {
    const int l = lo[b0 / kSpBox], h = hi[b0 / kSpBox];
    return l < j0 + kSgJ && h >= j0 && l < k0 + kSgK && h >= k0;
}
// rint(y) for |y| < 2^51 in two full-rate additions (round-to-nearest-even, like rint): the limbs are at most 40 bits wide
constexpr double kSgMagic = 6755399441055744.0;              // 1.5 * 2^52
__device__ __forceinline__ double sg_rint_scaled(double x, double s) {      // rint(x * s), s a power of two (x * s exact)
    return fma(x, s, kSgMagic) - kSgMagic;
}
template <int KR, int T>
__global__ __launch_bounds__(T) void sp_gram_kernel(const double* __restrict__ val, const int32_t* __restrict__ col, int64_t ld, int K, int64_t n,
                                                   const double* __restrict__ w2, const double* __restrict__ cs, double s1, double r1,
                                                   double s2, int nkb, int64_t rows_per_wg, const int32_t* __restrict__ box_lo,
                                                   const int32_t* __restrict__ box_hi, unsigned long long* __restrict__ partial) {
    __shared__ unsigned long long acc[2 * kSgTile];
    const int j0 = ((int)blockIdx.y / nkb) * kSgJ, k0 = ((int)blockIdx.y % nkb) * kSgK;
    if (k0 + kSgK - 1 < j0) return;                          // a tile wholly below the diagonal (uniform over the workgroup)
    for (int idx = threadIdx.x; idx < 2 * kSgTile; idx += T) acc[idx] = 0ull;
    __syncthreads();
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_wg, rend = (rbeg + rows_per_wg < n) ? rbeg + rows_per_wg : n;
    if constexpr (KR > 0) {
        constexpr int NP = KR * (KR + 1) / 2;
        double a1[NP], a2[NP];
        int cur[KR];
        double csr[KR];                                      // the scales of the run's columns: looked up when the column set changes
#pragma unroll
        for (int a = 0; a < KR; ++a) { cur[a] = -1; csr[a] = 0.0; }
#pragma unroll
        for (int p = 0; p < NP; ++p) a1[p] = a2[p] = 0.0;
        int run = 0;
        auto flush = [&]() {
            int p = 0;
#pragma unroll
            for (int a = 0; a < KR; ++a) {
                const int ca = cur[a] - j0;
#pragma unroll
                for (int b = a; b < KR; ++b, ++p) {
                    const int cb = cur[b] - k0;
                    if ((a1[p] != 0.0 || a2[p] != 0.0) && ca >= 0 && ca < kSgJ && cb >= 0 && cb < kSgK) {
                        const int idx = ca * kSgK + cb;
                        atomicAdd(&acc[idx], (unsigned long long)(long long)a1[p]);
                        atomicAdd(&acc[kSgTile + idx], (unsigned long long)(long long)a2[p]);
                    }
                    a1[p] = a2[p] = 0.0;
                }
            }
        };
        for (int64_t b0 = rbeg; b0 < rend; b0 += kSpBox) {
        if (!sg_box_hits(box_lo, box_hi, b0, j0, k0)) continue;
        const int64_t bend = (b0 + kSpBox < rend) ? b0 + kSpBox : rend;
        for (int64_t i = b0 + threadIdx.x; i < bend; i += T) {
            double v[KR];
            int c[KR];
            bool same = run < kSgRun;
#pragma unroll
            for (int a = 0; a < KR; ++a) {
                c[a] = a < K ? col[(int64_t)a * ld + i] : 0;
                v[a] = a < K ? val[(int64_t)a * ld + i] : 0.0;
                same = same && c[a] == cur[a];
            }
            if (!same) {
                flush();
                run = 0;
#pragma unroll
                for (int a = 0; a < KR; ++a) { cur[a] = c[a]; csr[a] = cs[c[a]]; }
            }
            ++run;
#pragma unroll
            for (int a = 0; a < KR; ++a) v[a] *= csr[a];     // (column scaling: an exact power of two)
            const double w = w2 ? w2[i] : 1.0;
            int p = 0;
#pragma unroll
            for (int a = 0; a < KR; ++a) {                   // the slots of a row are in ascending column order: c_a < c_b for a < b
                const double wa = w * v[a];
#pragma unroll
                for (int b = a; b < KR; ++b, ++p) {
                    const double x = wa * v[b];
                    const double q1 = sg_rint_scaled(x, s1);
                    a1[p] += q1;
                    a2[p] += sg_rint_scaled(fma(-q1, r1, x), s2);
                }
            }
        }
        }
        flush();
    } else {
        for (int64_t b0 = rbeg; b0 < rend; b0 += kSpBox) {
        if (!sg_box_hits(box_lo, box_hi, b0, j0, k0)) continue;
        const int64_t bend = (b0 + kSpBox < rend) ? b0 + kSpBox : rend;
        for (int64_t i = b0 + threadIdx.x; i < bend; i += T) {
            const double w = w2 ? w2[i] : 1.0;
            for (int a = 0; a < K; ++a) {
                const int ca0 = col[(int64_t)a * ld + i], ca = ca0 - j0;
                const double va = val[(int64_t)a * ld + i] * cs[ca0];
                if (va == 0.0 || ca < 0 || ca >= kSgJ) continue;
                const double wa = w * va;
                for (int b = a; b < K; ++b) {
                    const int cb0 = col[(int64_t)b * ld + i], cb = cb0 - k0;
                    const double vb = val[(int64_t)b * ld + i] * cs[cb0];
                    if (vb == 0.0 || cb < 0 || cb >= kSgK) continue;
                    const double x = wa * vb;
                    const double q1 = sg_rint_scaled(x, s1);
                    const double q2 = sg_rint_scaled(fma(-q1, r1, x), s2);
                    const int idx = ca * kSgK + cb;
                    atomicAdd(&acc[idx], (unsigned long long)(long long)q1);
                    atomicAdd(&acc[kSgTile + idx], (unsigned long long)(long long)q2);
                }
            }
        }
        }
    }
    __syncthreads();
    unsigned long long* out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * kSgTile);
    for (int idx = threadIdx.x; idx < 2 * kSgTile; idx += T) out[idx] = acc[idx];
}
// Rows of 9 .. 16 nonzeros: KR (KR + 1) / 2 pairs are too many limb sums for one lane, so PS lanes share a row stream -- lane q of the
// group keeps the pairs (a, b) with b = j PS + q (j < KR / PS; static register indices on both sides: every lane holds all KR values of
// the row for the a side and its own KR / PS of them for the b side), i.e. about KR^2 / (2 PS) pairs of two limbs each.  The lanes of a
// group load the same addresses (one request), the pair products cost what they cost in the one-lane form, and the LDS atomics again
// happen only when the column set of the stream changes.
template <int KR, int PS, int T, bool PF>
__global__ __launch_bounds__(T) void sp_gram_split_kernel(const double* __restrict__ val, const int32_t* __restrict__ col, int64_t ld, int K, int64_t n,
                                                         const double* __restrict__ w2, const double* __restrict__ cs, double s1, double r1,
                                                         double s2, int nkb, int64_t rows_per_wg, const int32_t* __restrict__ box_lo,
                                                         const int32_t* __restrict__ box_hi, unsigned long long* __restrict__ partial) {
    static_assert(KR % PS == 0 && T % PS == 0, "whole b-slots per lane, whole groups per workgroup");
    __shared__ unsigned long long acc[2 * kSgTile];
    const int j0 = ((int)blockIdx.y / nkb) * kSgJ, k0 = ((int)blockIdx.y % nkb) * kSgK;
    if (k0 + kSgK - 1 < j0) return;
    for (int idx = threadIdx.x; idx < 2 * kSgTile; idx += T) acc[idx] = 0ull;
    __syncthreads();
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_wg, rend = (rbeg + rows_per_wg < n) ? rbeg + rows_per_wg : n;
    constexpr int NB = KR / PS, G = T / PS;
    const int q = (int)threadIdx.x % PS, g = (int)threadIdx.x / PS;
    double a1[NB][KR], a2[NB][KR];                           // (j, a): used for a < (j + 1) PS only -- the rest is never touched
    int cur[KR], curb[NB];
    double csr[KR];
#pragma unroll
    for (int a = 0; a < KR; ++a) { cur[a] = -1; csr[a] = 0.0; }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        curb[j] = -1;
#pragma unroll
        for (int a = 0; a < (j + 1) * PS; ++a) a1[j][a] = a2[j][a] = 0.0;
    }
    int run = 0;
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int cb = curb[j] - k0;
#pragma unroll
            for (int a = 0; a < (j + 1) * PS; ++a) {
                const int ca = cur[a] - j0;
                if ((a1[j][a] != 0.0 || a2[j][a] != 0.0) && ca >= 0 && ca < kSgJ && cb >= 0 && cb < kSgK) {
                    const int idx = ca * kSgK + cb;
                    atomicAdd(&acc[idx], (unsigned long long)(long long)a1[j][a]);
                    atomicAdd(&acc[kSgTile + idx], (unsigned long long)(long long)a2[j][a]);
                }
                a1[j][a] = a2[j][a] = 0.0;
            }
        }
    };
    for (int64_t b0 = rbeg; b0 < rend; b0 += kSpBox) {
    if (!sg_box_hits(box_lo, box_hi, b0, j0, k0)) continue;
    const int64_t bend = (b0 + kSpBox < rend) ? b0 + kSpBox : rend;
    double vn[PF ? KR : 1];                                  // PF: the next row of the block is requested a row ahead (K <= 12: 3 % faster; at 16
                                                             // the 48 extra registers spill and cost 6 %)
    int cn[PF ? KR : 1];
    if (PF && b0 + g < bend) {
#pragma unroll
        for (int a = 0; a < KR; ++a) {
            cn[PF ? a : 0] = a < K ? col[(int64_t)a * ld + b0 + g] : 0;
            vn[PF ? a : 0] = a < K ? val[(int64_t)a * ld + b0 + g] : 0.0;
        }
    }
    for (int64_t i = b0 + g; i < bend; i += G) {
        double v[KR];
        int c[KR];
        bool same = run < kSgRun;
        if (PF) {
#pragma unroll
            for (int a = 0; a < KR; ++a) { c[a] = cn[PF ? a : 0]; v[a] = vn[PF ? a : 0]; same = same && c[a] == cur[a]; }
            if (i + G < bend) {
#pragma unroll
                for (int a = 0; a < KR; ++a) {
                    cn[PF ? a : 0] = a < K ? col[(int64_t)a * ld + i + G] : 0;
                    vn[PF ? a : 0] = a < K ? val[(int64_t)a * ld + i + G] : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < KR; ++a) {
                c[a] = a < K ? col[(int64_t)a * ld + i] : 0;
                v[a] = a < K ? val[(int64_t)a * ld + i] : 0.0;
                same = same && c[a] == cur[a];
            }
        }
        if (!same) {
            flush();
            run = 0;
#pragma unroll
            for (int a = 0; a < KR; ++a) { cur[a] = c[a]; csr[a] = cs[c[a]]; }
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int sl = 0; sl < PS; ++sl)
                    if (q == sl) curb[j] = c[j * PS + sl];
        }
        ++run;
        const double w = w2 ? w2[i] : 1.0;
#pragma unroll
        for (int a = 0; a < KR; ++a) v[a] *= csr[a];         // (column scaling: an exact power of two)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double vb = 0.0;
#pragma unroll
            for (int sl = 0; sl < PS; ++sl)
                if (q == sl) vb = v[j * PS + sl];
#pragma unroll
            for (int a = 0; a < (j + 1) * PS; ++a) {         // a <= b = j PS + q: the slots of a row are in ascending column order
                const double x = (a < j * PS || a - j * PS <= q) ? (w * v[a]) * vb : 0.0;
                const double q1 = sg_rint_scaled(x, s1);
                a1[j][a] += q1;
                a2[j][a] += sg_rint_scaled(fma(-q1, r1, x), s2);
            }
        }
    }
    }
    flush();
    __syncthreads();
    unsigned long long* out = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * kSgTile);
    for (int idx = threadIdx.x; idx < 2 * kSgTile; idx += T) out[idx] = acc[idx];
}
// sums the workgroups' limbs of each entry, rounds once, writes G[j, k] and its mirror image (G: m x m, column-major, leading dimension ldg)
__global__ __launch_bounds__(256) void sp_gram_reduce_kernel(const unsigned long long* __restrict__ partial, int nblk, int nsum, int nkb, int m, double r1,
                                                             double r2, const double* __restrict__ csinv, double* __restrict__ G, int ldg) {
    const int j0 = ((int)blockIdx.y / nkb) * kSgJ, k0 = ((int)blockIdx.y % nkb) * kSgK;
    if (k0 + kSgK - 1 < j0) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int j = j0 + idx / kSgK, k = k0 + idx % kSgK;
    if (j > k || k >= m) return;
    const unsigned long long* p = partial + (int64_t)blockIdx.y * nblk * (2 * kSgTile) + idx;
    long long a1 = 0, a2 = 0;
    for (int b = 0; b < nsum; ++b) {                          // (nsum = nblk, or the folded slots of sp_gram_fold_kernel)
        a1 += (long long)p[(int64_t)b * (2 * kSgTile)];
        a2 += (long long)p[(int64_t)b * (2 * kSgTile) + kSgTile];
    }
    const double hi = (double)a1;                             // |a1| <= 2^62: the conversion rounds, its error is an exact integer
    const double lo = (double)(a1 - (long long)hi);
    const double g = (hi * r1 + (lo * r1 + (double)a2 * r2)) * csinv[j] * csinv[k];      // (undoing the column scaling: exact)
    G[(int64_t)k * ldg + j] = g;
    G[(int64_t)j * ldg + k] = g;
}
// many slices: slot z of a tile's partials takes the sum of the slots z, z + nfold, z + 2 nfold, ... (integers: any order gives the same
// bits), so that the reduction above reads nfold slots per entry and this one runs on (entries x nfold) threads
__global__ __launch_bounds__(256) void sp_gram_fold_kernel(unsigned long long* __restrict__ partial, int nblk, int nfold, int nkb) {
    const int j0 = ((int)blockIdx.y / nkb) * kSgJ, k0 = ((int)blockIdx.y % nkb) * kSgK;
    if (k0 + kSgK - 1 < j0) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;              // < 2 * kSgTile: both limbs
    unsigned long long* p = partial + (int64_t)blockIdx.y * nblk * (2 * kSgTile) + idx;
    unsigned long long a = 0ull;
    for (int b = (int)blockIdx.z; b < nblk; b += nfold) a += p[(int64_t)b * (2 * kSgTile)];
    p[(int64_t)blockIdx.z * (2 * kSgTile)] = a;
}
// max over the rows of w2_i (max_a |v'_ia|)^2, v' the column-scaled values: the bound on the terms.  (The largest weight alone will not do: a row
// without nonzeros -- the slack row of a ball constraint -- may carry a weight 2^23 times the others, and every bit of slack in the bound is a
// bit lost at the bottom of the second limb.)
__global__ __launch_bounds__(1024) void sp_gram_bound_kernel(const double* __restrict__ val, const int32_t* __restrict__ col, int64_t ld, int K,
                                                             int64_t n, const double* __restrict__ w2, const double* __restrict__ cs,
                                                             unsigned long long* __restrict__ out) {
    __shared__ double red[1024];
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        double r = 0.0;
        for (int a = 0; a < K; ++a) r = fmax(r, fabs(val[(int64_t)a * ld + i] * cs[col[(int64_t)a * ld + i]]));
        const double t = fabs(w2[i]) * r * r;
        mx = (t > mx || t != t) ? t : mx;                     // (a NaN sticks)
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            const double o = red[threadIdx.x + h];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned long long bits;
        const double r0 = red[0];
        memcpy(&bits, &r0, sizeof(bits));
        atomicMax(out, bits);
    }     // non-negative doubles (and NaN above them) order like integers
}
__global__ void sp_mul_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

// G (host, mt x mt column-major, mt = S->m + nx) = A' diag(w2) A for A = [S | X[:, x0 : x0+nx)], all-reduced over the ranks.  scratch: an
// n-vector (used when there are extra columns and weights).  LFPSQP_ERR_UNSUPPORTED when the nonzeros do not allow the exact accumulation
// (rows wider than kmax nonzeros -- the register-resident kernels cover 8, beyond that the per-term LDS atomics are slower than the MFMA Gram
// of a dense copy at K = 12 already (10.4 vs 2.8 ms at n = 1e7, m = 128) --, non-finite or extreme values): the caller then forms the Gram
// matrix on a dense copy.
int sp_gram(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* X, int x0, int nx, const lfpsqp_vec* w2, double* scratch, double* G, int kmax) {
    const int ms = (int)S->m, mt = ms + nx;
    for (size_t e = 0; e < (size_t)mt * mt; ++e) G[e] = 0.0;
    if (mt == 0) return 0;
    // whether the exact accumulation is possible is a LOCAL property (row width, finite values) -- the ranks must agree on it before the first
    // collective, or one would enter the dense Gram's all-reduce while the others enter this one's
    double cannot = (S->K > kmax || !(S->amax == S->amax)) ? 1.0 : 0.0;
    if (ctx->comm_active()) {
        LF_HIP(ctx, hipMemcpyAsync(ctx->scal + 41, &cannot, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));       // (a stack variable)
        LF_TRY(allreduce_dev(ctx, ctx->scal + 41, 1, 1));
        LF_TRY(read_back(ctx, ctx->scal + 41, &cannot, 1));
    }
    if (cannot != 0.0) return LFPSQP_ERR_UNSUPPORTED;
    // |column-scaled value| < 1, so without weights every term is below 1; with weights the bound is measured (sp_gram_bound_kernel)
    double B = 1.0;
    // (collectives are entered by every rank, also by one that holds no rows: only the kernel launches depend on the local row count)
    if (w2 && ms > 0) {
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctx->scal + 40);
        LF_HIP(ctx, hipMemsetAsync(slot, 0, sizeof(unsigned long long), ctx->stream));
        if (S->n > 0) {
            hipLaunchKernelGGL(sp_gram_bound_kernel, dim3((unsigned)std::min<int64_t>((S->n + 1023) / 1024, 1024)), dim3(1024), 0, ctx->stream,
                               S->ell_val, S->ell_col, S->ld, S->K, S->n, w2->p, S->col_scale, slot);
            LF_LAUNCH_CHECK(ctx);
        }
        LF_TRY(allreduce_dev(ctx, ctx->scal + 40, 1, 1));      // (every rank on the same grid)
        LF_TRY(read_back(ctx, ctx->scal + 40, &B, 1));
    }
    if (!(B == B) || B > 1e200 || (B != 0.0 && B < 1e-200)) return LFPSQP_ERR_UNSUPPORTED;      // (B is global: every rank decides alike)
    LF_TRY(ensure_small(ctx, (size_t)ms * ms + 16));
    LF_HIP(ctx, hipMemsetAsync(ctx->small, 0, sizeof(double) * (size_t)ms * ms, ctx->stream));
    if (ms > 0 && B != 0.0) {
        int E = 0;
        (void)frexp(B, &E);                                   // B < 2^E
        E += 1;                                               // (one binade of slack for the rounding of the three-factor product)
        int hb = 0;
        while (((int64_t)1 << hb) < S->n + 1) ++hb;
        const int b = std::min(62 - hb, kSgLimbMax);
        const double s1 = ldexp(1.0, b - E), r1 = ldexp(1.0, E - b), s2 = ldexp(1.0, 2 * b - E), r2 = ldexp(1.0, E - 2 * b);
        const int njb = (ms + kSgJ - 1) / kSgJ, nkb = (ms + kSgK - 1) / kSgK;
        int active = 0;
        for (int t = 0; t < njb * nkb; ++t) active += ((t % nkb) * kSgK + kSgK - 1 >= (t / nkb) * kSgJ);
        const int64_t cus = ctx->num_cu > 0 ? ctx->num_cu : 64;
        // one workgroup per CU (LDS); at least eight slices of the rows.  The wide-row kernels are arithmetic-bound and skip the row blocks whose
        // columns cannot touch their tile (sg_box_hits): eight times as many, shorter slices, so that the workgroups with nothing to do make room
        // for the others (a banded system: every slice matters to one or two tiles only)
        const char* slices_env = getenv("LFPSQP_SPGRAM_SLICES");      // (tuning experiments and tests: slices per CU and active tile)
        const int slices_x = slices_env ? atoi(slices_env) : 0;
        int64_t nblk = std::max<int64_t>((slices_x > 0 ? slices_x : (S->K > 8 ? 8 : 1)) * cus / active, 8);
        nblk = std::min<int64_t>(nblk, std::max<int64_t>((S->n + kSpBox - 1) / kSpBox, 1));
        LF_TRY(ensure_part(ctx, (size_t)njb * nkb * nblk * 2 * kSgTile + 8));
        unsigned long long* part = reinterpret_cast<unsigned long long*>(ctx->part);
        if (S->n > 0) {
            const int64_t rows_per_wg = round_up((S->n + nblk - 1) / nblk, kSpBox);      // whole column boxes per workgroup
            const dim3 grid((unsigned)nblk, (unsigned)(njb * nkb));
            const double* w2p = w2 ? w2->p : nullptr;
            if (S->K <= 4)
                hipLaunchKernelGGL((sp_gram_kernel<4, kSgThreads>), grid, dim3(kSgThreads), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, S->n, w2p,
                                   S->col_scale, s1, r1, s2, nkb, rows_per_wg, S->box_lo, S->box_hi, part);
            else if (S->K <= 8)                                  // 72 limb sums per lane: two waves per SIMD
                hipLaunchKernelGGL((sp_gram_kernel<8, 512>), grid, dim3(512), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, S->n, w2p, S->col_scale, s1, r1,
                                   s2, nkb, rows_per_wg, S->box_lo, S->box_hi, part);
            else if (S->K <= 12 && ctx->tune_spgram >= 0)
                hipLaunchKernelGGL((sp_gram_split_kernel<12, 4, 512, true>), grid, dim3(512), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, S->n, w2p,
                                   S->col_scale, s1, r1, s2, nkb, rows_per_wg, S->box_lo, S->box_hi, part);
            else if (S->K <= 16 && ctx->tune_spgram >= 0)
                hipLaunchKernelGGL((sp_gram_split_kernel<16, 8, 512, false>), grid, dim3(512), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, S->n, w2p,
                                   S->col_scale, s1, r1, s2, nkb, rows_per_wg, S->box_lo, S->box_hi, part);
            else
                hipLaunchKernelGGL((sp_gram_kernel<0, kSgThreads>), grid, dim3(kSgThreads), 0, ctx->stream, S->ell_val, S->ell_col, S->ld, S->K, S->n, w2p,
                                   S->col_scale, s1, r1, s2, nkb, rows_per_wg, S->box_lo, S->box_hi, part);
            constexpr int kFold = 32;
            if (nblk > 2 * kFold)
                hipLaunchKernelGGL(sp_gram_fold_kernel, dim3(2 * kSgTile / 256, (unsigned)(njb * nkb), kFold), dim3(256), 0, ctx->stream, part, (int)nblk,
                                   kFold, nkb);
            // (the reduction's stride between slots stays nblk; it reads the first kFold of them after a fold)
            hipLaunchKernelGGL(sp_gram_reduce_kernel, dim3(kSgTile / 256, (unsigned)(njb * nkb)), dim3(256), 0, ctx->stream, part, (int)nblk,
                               (int)(nblk > 2 * kFold ? kFold : nblk), nkb, ms, r1, r2,
                               S->col_scale + S->m, ctx->small, ms);
            LF_LAUNCH_CHECK(ctx);
        }
        LF_TRY(allreduce_dev(ctx, ctx->small, (int64_t)ms * ms, 0));
    }
    std::vector<double> h((size_t)ms * ms + 1);
    if (ms > 0) {
        LF_HIP(ctx, hipMemcpyAsync(h.data(), ctx->small, sizeof(double) * (size_t)ms * ms, hipMemcpyDeviceToHost, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < ms; ++k)
            for (int j = 0; j < ms; ++j) G[(size_t)k * mt + j] = h[(size_t)k * ms + j];
    }
    // the dense extra columns: G[0:ms, ms + c] = S' (w2 .* x_c) streams the nonzeros, the corner is a handful of dot products
    if (nx > 0) LF_TRY(ensure_mvec(ctx, (size_t)mt + 8));
    for (int c = 0; c < nx; ++c) {
        const double* xc = X->p + (int64_t)(x0 + c) * X->ld;
        const double* wx = xc;
        if (w2) {
            hipLaunchKernelGGL(sp_mul_kernel, dim3(2048), dim3(256), 0, ctx->stream, w2->p, xc, scratch, S->n);
            LF_LAUNCH_CHECK(ctx);
            wx = scratch;
        }
        if (ms > 0) {
            LF_TRY(spmv_t(ctx, S, wx, ctx->d_m));
            LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->d_m, sizeof(double) * ms, hipMemcpyDeviceToHost, ctx->stream));
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int j = 0; j < ms; ++j) G[(size_t)(ms + c) * mt + j] = G[(size_t)j * mt + ms + c] = ctx->h_m[j];
        }
        lfpsqp_vec va, vb;
        va.p = const_cast<double*>(wx); va.n = S->n; va.cap = round_up(S->n, kPadRows);     // (scratch and matrix columns are both padded to whole tiles)
        for (int d = c; d < nx; ++d) {
            vb.p = X->p + (int64_t)(x0 + d) * X->ld; vb.n = S->n; vb.cap = round_up(S->n, kPadRows);
            double dv = 0.0;
            LF_TRY(lfpsqp_dot(ctx, &va, &vb, &dv));
            G[(size_t)(ms + d) * mt + ms + c] = G[(size_t)(ms + c) * mt + ms + d] = dv;
        }
    }
    return 0;
}

}  // namespace lfpsqp

using namespace lfpsqp;

extern "C" {

int lfpsqp_spmat_create(lfpsqp_ctx* ctx, int64_t n, int64_t m, int64_t nnz, const int64_t* rows, const int64_t* cols, const double* vals,
                        lfpsqp_spmat** out) {
    LF_ARG(ctx, ctx && out && n >= 0 && m >= 0 && nnz >= 0 && (nnz == 0 || (rows && cols && vals)) && m < (1 << 30) && n < ((int64_t)1 << 31));
    *out = nullptr;
    for (int64_t e = 0; e < nnz; ++e)
        if (rows[e] < 0 || rows[e] >= n || cols[e] < 0 || cols[e] >= m) return set_err(ctx, LFPSQP_ERR_ARG, "spmat: entry %lld out of range", (long long)e);
    // entries ordered by (column, row); duplicates (same row and column) are summed, so every format holds one entry per position
    std::vector<int64_t> order((size_t)nnz);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return cols[a] != cols[b] ? cols[a] < cols[b] : rows[a] < rows[b]; });
    std::vector<int32_t> hr;
    std::vector<int32_t> hcol;
    std::vector<double> hcv;
    hr.reserve((size_t)nnz + 1); hcol.reserve((size_t)nnz + 1); hcv.reserve((size_t)nnz + 1);
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t e = order[(size_t)k];
        if (!hr.empty() && hr.back() == (int32_t)rows[e] && hcol.back() == (int32_t)cols[e]) hcv.back() += vals[e];
        else { hr.push_back((int32_t)rows[e]); hcol.push_back((int32_t)cols[e]); hcv.push_back(vals[e]); }
    }
    nnz = (int64_t)hr.size();
    double amax = 0.0;
    for (double v : hcv) amax = std::max(amax, fabs(v));          // (a NaN entry: the comparison keeps amax, sp_gram's own test of the
    for (double v : hcv) if (!(v == v) || fabs(v) > 1e300) amax = NAN;    //  bound then refuses the exact accumulation)
    std::vector<int32_t> cnt((size_t)std::max<int64_t>(n, 1), 0);
    for (int64_t k = 0; k < nnz; ++k) cnt[(size_t)hr[(size_t)k]]++;
    int K = 0;
    for (int64_t i = 0; i < n; ++i) K = std::max(K, (int)cnt[(size_t)i]);
    if (K > 256) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "spmat: a row with %d nonzeros (> 256): keep such constraint gradients dense", K);
    lfpsqp_spmat* S = new lfpsqp_spmat();
    S->n = n; S->m = m; S->nnz = nnz; S->K = K;
    S->ld = round_up(n > 0 ? n : 1, kPadRows);
    const size_t ell = (size_t)std::max(K, 1) * (size_t)S->ld;
    std::vector<double> hv(ell, 0.0);
    std::vector<int32_t> hc(ell, 0);
    std::fill(cnt.begin(), cnt.end(), 0);
    std::vector<int64_t> colptr((size_t)m + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) {
        colptr[(size_t)hcol[(size_t)k] + 1]++;
        const int32_t r = hr[(size_t)k], slot = cnt[(size_t)r]++;   // ELL slots of a row in column order
        hv[(size_t)slot * S->ld + r] = hcv[(size_t)k];
        hc[(size_t)slot * S->ld + r] = hcol[(size_t)k];
    }
    for (int64_t j = 0; j < m; ++j) colptr[(size_t)j + 1] += colptr[(size_t)j];
    std::vector<int64_t> cb2;
    std::vector<int32_t> cchunk((size_t)m + 1, 0);
    for (int64_t j = 0; j < m; ++j) {
        cchunk[(size_t)j] = (int32_t)cb2.size();
        for (int64_t e = colptr[(size_t)j]; e < colptr[(size_t)j + 1]; e += kSpChunk) cb2.push_back(e);
    }
    cchunk[(size_t)m] = (int32_t)cb2.size();
    S->nchunks = (int64_t)cb2.size();
    cb2.push_back(nnz);
    if (hr.empty()) { hr.push_back(0); hcv.push_back(0.0); }
    // the kernel reads [chunk_beg[c], chunk_beg[c+1]): consecutive chunks are contiguous in the CSC order, also across columns
    bool ok = true;
    auto dev_copy = [&](void** dst, const void* src, size_t bytes) {
        if (!ok) return;
        ok = hipMalloc(dst, bytes > 0 ? bytes : 8) == hipSuccess;
        if (ok && bytes > 0) ok = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    };
    dev_copy((void**)&S->ell_val, hv.data(), ell * sizeof(double));
    dev_copy((void**)&S->ell_col, hc.data(), ell * sizeof(int32_t));
    dev_copy((void**)&S->csc_row, hr.data(), hr.size() * sizeof(int32_t));
    dev_copy((void**)&S->csc_val, hcv.data(), hcv.size() * sizeof(double));
    dev_copy((void**)&S->colptr, colptr.data(), colptr.size() * sizeof(int64_t));
    dev_copy((void**)&S->chunk_beg, cb2.data(), cb2.size() * sizeof(int64_t));
    dev_copy((void**)&S->col_chunk, cchunk.data(), cchunk.size() * sizeof(int32_t));
    {   // column bounding boxes of row blocks, and how often a row repeats the column set of the row 128 places before it
        const int64_t nbox = (std::max<int64_t>(n, 1) + kSpBox - 1) / kSpBox;
        std::vector<int32_t> lo((size_t)nbox, INT32_MAX), hi((size_t)nbox, -1);
        int64_t same = 0;
        for (int64_t i = 0; i < n; ++i) {
            const size_t b = (size_t)(i / kSpBox);
            bool eq = i >= 128 && cnt[(size_t)i] == cnt[(size_t)(i - 128)];
            for (int a = 0; a < cnt[(size_t)i]; ++a) {
                const int32_t c = hc[(size_t)a * S->ld + i];
                lo[b] = std::min(lo[b], c); hi[b] = std::max(hi[b], c);
                eq = eq && c == hc[(size_t)a * S->ld + i - 128];
            }
            same += eq ? 1 : 0;
        }
        S->run_frac = n > 128 ? (double)same / (double)(n - 128) : 0.0;
        dev_copy((void**)&S->box_lo, lo.data(), lo.size() * sizeof(int32_t));
        dev_copy((void**)&S->box_hi, hi.data(), hi.size() * sizeof(int32_t));
    }
    // power-of-two column scales for the exact Gram accumulation: entrywise accuracy relative to the columns' own magnitudes, whatever their scaling
    std::vector<double> cscale(2 * (size_t)std::max<int64_t>(m, 1), 1.0);
    {
        std::vector<double> cmax((size_t)std::max<int64_t>(m, 1), 0.0);
        for (size_t k = 0; k < hcv.size() && k < hcol.size(); ++k)
            if (fabs(hcv[k]) > cmax[(size_t)hcol[k]]) cmax[(size_t)hcol[k]] = fabs(hcv[k]);
        for (int64_t j = 0; j < m; ++j) {
            int e = 0;
            if (cmax[(size_t)j] > 0.0 && cmax[(size_t)j] < 1e300) (void)frexp(cmax[(size_t)j], &e);      // cmax = f 2^e, f in [0.5, 1)
            if (e > 400 || e < -400) amax = NAN;                // (sp_gram refuses; the scalings below stay finite)
            e = std::max(-400, std::min(400, e));
            cscale[(size_t)j] = ldexp(1.0, -e);
            cscale[(size_t)m + (size_t)j] = ldexp(1.0, e);
        }
    }
    dev_copy((void**)&S->col_scale, cscale.data(), cscale.size() * sizeof(double));
    S->amax = amax;
    if (ok) ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) {
        lfpsqp_spmat_free(ctx, S);
        return set_err(ctx, LFPSQP_ERR_HIP, "spmat: device allocation / upload failed");
    }
    *out = S;
    return 0;
}

int lfpsqp_spmat_free(lfpsqp_ctx* ctx, lfpsqp_spmat* S) {
    if (!S) return 0;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (void* p : {(void*)S->ell_val, (void*)S->csc_val, (void*)S->col_scale})
        if (p) (void)hipFree(p);
    if (S->owns_structure)
        for (void* p : {(void*)S->ell_col, (void*)S->csc_row, (void*)S->colptr, (void*)S->chunk_beg, (void*)S->col_chunk, (void*)S->box_lo, (void*)S->box_hi})
            if (p) (void)hipFree(p);
    delete S;
    return 0;
}

// ---- x-dependent values on a fixed structure (lfpsqp_elementwise: Jct(x) = diag(phi'(x)) A) ------------------------------------------
int lfpsqp_spmat_clone(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, lfpsqp_spmat** out) {
    LF_ARG(ctx, ctx && S && out);
    *out = nullptr;
    lfpsqp_spmat* C = new lfpsqp_spmat(*S);
    C->owns_structure = false;
    C->ell_val = nullptr; C->csc_val = nullptr; C->col_scale = nullptr;
    const size_t ell = (size_t)std::max(S->K, 1) * (size_t)S->ld, nz = (size_t)std::max<int64_t>(S->nnz, 1), cs = 2 * (size_t)std::max<int64_t>(S->m, 1);
    bool ok = hipMalloc((void**)&C->ell_val, ell * sizeof(double)) == hipSuccess && hipMalloc((void**)&C->csc_val, nz * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&C->col_scale, cs * sizeof(double)) == hipSuccess;
    ok = ok && hipMemcpyAsync(C->ell_val, S->ell_val, ell * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
         hipMemcpyAsync(C->csc_val, S->csc_val, nz * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
         hipMemcpyAsync(C->col_scale, S->col_scale, cs * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
         hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (!ok) {
        lfpsqp_spmat_free(ctx, C);
        return set_err(ctx, LFPSQP_ERR_HIP, "spmat_clone: device allocation / copy failed");
    }
    *out = C;
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void sp_rowscale_ell_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t ld, int K, int64_t n,
                                                              const double* __restrict__ v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double s = v[i];
        for (int k = 0; k < K; ++k) dst[(int64_t)k * ld + i] = src[(int64_t)k * ld + i] * s;
    }
}
// one workgroup per column: the CSC values of the column, its largest magnitude -> the power-of-two scales of sp_gram, and the global
// maximum (bit pattern of a non-negative double, NaN above everything) for the finiteness test
__global__ __launch_bounds__(256) void sp_rowscale_csc_kernel(const int64_t* __restrict__ colptr, const int32_t* __restrict__ row,
                                                              const double* __restrict__ src, double* __restrict__ dst, const double* __restrict__ v,
                                                              int m, double* __restrict__ cscale, unsigned long long* __restrict__ amax_bits) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    double mx = 0.0;
    for (int64_t e = colptr[j] + threadIdx.x; e < colptr[j + 1]; e += 256) {
        const double a = src[e] * v[row[e]];
        dst[e] = a;
        const double f = fabs(a);
        mx = (f > mx || f != f) ? f : mx;
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            const double o = red[threadIdx.x + h];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double c = red[0];
        if (!(c == c) || c > 1e300) c = NAN;
        int e = 0;
        if (c == c && c > 0.0) (void)frexp(c, &e);
        if (e > 400 || e < -400) c = NAN;
        e = e > 400 ? 400 : (e < -400 ? -400 : e);
        cscale[j] = ldexp(1.0, -e);
        cscale[m + j] = ldexp(1.0, e);
        unsigned long long bits;
        memcpy(&bits, &c, sizeof(bits));
        atomicMax(amax_bits, bits);
    }
}
}  // namespace

int lfpsqp_spmat_rowscale(lfpsqp_ctx* ctx, lfpsqp_spmat* dst, const lfpsqp_spmat* src, const lfpsqp_vec* v) {
    LF_ARG(ctx, ctx && dst && src && v && v->n >= src->n && dst->n == src->n && dst->m == src->m && dst->nnz == src->nnz && dst->K == src->K &&
                    dst->ell_col == src->ell_col && dst->csc_row == src->csc_row);
    if (src->n > 0 && src->K > 0) {
        hipLaunchKernelGGL(sp_rowscale_ell_kernel, dim3((unsigned)std::min<int64_t>((src->n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, src->ell_val,
                           dst->ell_val, src->ld, src->K, src->n, v->p);
        LF_LAUNCH_CHECK(ctx);
    }
    double amax = 0.0;
    if (src->m > 0) {
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(ctx->scal + 42);
        LF_HIP(ctx, hipMemsetAsync(slot, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(sp_rowscale_csc_kernel, dim3((unsigned)src->m), dim3(256), 0, ctx->stream, src->colptr, src->csc_row, src->csc_val, dst->csc_val,
                           v->p, (int)src->m, dst->col_scale, slot);
        LF_LAUNCH_CHECK(ctx);
        LF_TRY(read_back(ctx, ctx->scal + 42, &amax, 1));
    }
    dst->amax = amax;
    return 0;
}

int lfpsqp_spmat_info(const lfpsqp_spmat* S, int64_t* n, int64_t* m, int64_t* nnz, int64_t* ell_width) {
    if (!S) return LFPSQP_ERR_ARG;
    if (n) *n = S->n;
    if (m) *m = S->m;
    if (nnz) *nnz = S->nnz;
    if (ell_width) *ell_width = S->K;
    return 0;
}

int lfpsqp_spmv_t(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_vec* v, lfpsqp_vec* t) {
    LF_ARG(ctx, ctx && S && v && t && v->n >= S->n && t->n >= S->m);
    return spmv_t(ctx, S, v->p, t->p);
}

int lfpsqp_spmv_n(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, double alpha, const lfpsqp_vec* t, double beta, lfpsqp_vec* y) {
    LF_ARG(ctx, ctx && S && t && y && y->n >= S->n && t->n >= S->m && t->p != y->p);
    return run_vec<SpmvNF, 0, NoPost>(ctx, S->n, SpmvNF{ell_rows(S, t->p), y->p, alpha, beta}, 0u, nullptr, NoPost());
}

int lfpsqp_spmat_gram(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, double* G) {
    LF_ARG(ctx, ctx && S && G && (!w2 || w2->n == S->n) && (!Jct || (plain_mat(Jct) && Jct->n == S->n && Jct->m >= S->m && Jct->m - S->m <= 4)));
    const int nx = Jct ? (int)(Jct->m - S->m) : 0;
    double* scratch = nullptr;
    if (nx > 0 && w2) {     // an n-vector like any other: whole tiles (the vector kernels load row pairs up to the end of the last tile)
        const size_t cap = (size_t)round_up(S->n > 0 ? S->n : 1, kPadRows);
        LF_HIP(ctx, hipMalloc((void**)&scratch, sizeof(double) * cap));
        LF_HIP(ctx, hipMemsetAsync(scratch, 0, sizeof(double) * cap, ctx->stream));
    }
    const int rc = sp_gram(ctx, S, Jct, (int)S->m, nx, w2, scratch, G, 32);
    if (scratch) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(scratch); }
    if (rc == LFPSQP_ERR_UNSUPPORTED) return set_err(ctx, rc, "spmat_gram: rows wider than 32 nonzeros or values outside the exactly accumulable range");
    return rc;
}

int lfpsqp_spmat_to_dense(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, lfpsqp_mat* M) {
    LF_ARG(ctx, ctx && S && plain_mat(M) && M->n == S->n && M->m >= S->m);
    if (S->m == 0 || S->n == 0) return 0;
    LF_HIP(ctx, hipMemsetAsync(M->p, 0, sizeof(double) * (size_t)M->ld * (size_t)S->m, ctx->stream));
    hipLaunchKernelGGL(sp_scatter_kernel, dim3(64, (unsigned)S->m), dim3(256), 0, ctx->stream, S->colptr, S->csc_row, S->csc_val, M->p, M->ld);
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
