// Tangent setup on the device: weighted Gram matrix, right-multiplication by a small matrix, and
// the thin factorisation built from them (replaces the reference's per-outer-iteration LAPACK dgesvd,
// src/la_helper.jl:8-34).  These are the only genuinely contraction-shaped (compute-bound) operations
// of the hot path (AI = m/4 flop/B), so they run on the matrix cores: v_mfma_f64_16x16x4_f64 (lane l holds
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D row = (l>>4)+4*reg, col = l&15) on 128 x 128 output tiles.
// gram_kernel and rmul_kernel stage 16-deep K steps through double-buffered LDS with the operand reads one k-group
// ahead of the MFMAs and one barrier per step; rmul_resident_kernel keeps the small factor in LDS for the workgroup's
// lifetime and streams the matrix straight into the B operand.  What bounds each: FINDINGS.md 5.3.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <numeric>
#include <type_traits>
#include <vector>

#include <functional>

#include "internal.h"
#include "smallla.h"
#include "sparse.h"

namespace lfpsqp {

constexpr int kPanel = 128;
constexpr int kKStep = 16;
constexpr int kLdsLd = 144;   // row stride 288 dwords = 32 mod 64: the two k-rows a half-wave reads hit disjoint banks
typedef double f64x4 __attribute__((vector_size(32)));

// The Gram kernel stages its panels TRANSPOSED, T[column][k], with a row stride of 17 doubles (34 dwords).  The compiler pairs the
// operand reads of two k-groups into ds_read2_b64, which the LDS serves in groups of 16 consecutive lanes on (address / 4) mod 32
// banks: the 16 columns of a group start 2 banks apart and each read covers 2 -- conflict-free (a stride of 18 doubles, 16-byte
// aligned for b128 stores, is 2-way conflicted on those reads).  The staging stores (two rows of
// one column per lane, 8 lanes per column) are conflict-free by the same arithmetic.
constexpr int kTLd = 17;
// Gram partials.  One launch per kind of panel pair (pi <= pj; the Gram matrix is symmetric, only the upper block triangle is
// computed -- 10 of 16 pairs at m = 512): DIAG = the npan diagonal pairs (pi, pi), else the npan (npan - 1) / 2 pairs pi < pj; two
// instantiations, so that each gets its own register allocation.  A launch runs `ngroups` row groups of every one of its pairs
// CONCURRENTLY: workgroup (g, pair) sums the 16-row steps g, g + ngroups, ... of its pair and writes the 128 x 128 partial to
// part[g][slot * 16384 + j * 128 + i] (i = row of G within panel pi, j = column within panel pj; slot = pi for a diagonal pair,
// npan + position among the pairs pi < pj otherwise).  The pairs of one row group read the same rows of the matrix (every panel
// is an operand of npan - 1 off-diagonal pairs), so they are placed on the SAME XCD, next to each other in dispatch order
// (linear workgroup id L -> XCD L % 8): the panels then come from HBM once and from that XCD's L2 for the other pairs.  With
// the pair as the slow grid dimension (one pair after the other) the m = 512 Gram read 61 GB for a 20 GB matrix.
// SHIFT (weighted launches only): the operand is not M but R with rows R_i = M_i + sgn_i M_{i+1} (sgn_i = +-1; M_n = 0), formed in registers
// on the way to LDS -- one more scalar load per column and step (the row below a lane's pair; same cache line but for one lane in eight).
// This is the Gram matrix behind the reduced operator of a tridiagonal Hessian (lfpsqp_projcg_tridiag: U'A U = R'|off| R + U' diag(c) U).
template <bool DIAG, bool WEIGHTED, bool SHIFT = false, int NX = 0>
__global__ __launch_bounds__(kThreads, 2) void gram_kernel(const double* __restrict__ M, int64_t ld, int64_t n, int ncols, int npan,
                                                         int ngroups, const double* __restrict__ w2, double* __restrict__ part,
                                                         int64_t part_ld, const double* __restrict__ ex0, const double* __restrict__ ex1,
                                                         int64_t xoff, const double* __restrict__ sgn) {
    static_assert(!SHIFT || WEIGHTED, "the shifted operand exists for weighted launches only");
    // NX: how many of the extra right-hand columns (ex0, then ex1) this launch carries -- a compile-time fact: as run-time null tests of two
    // pointers the columns' multiply-adds were if-converted into 32 FMAs and ~37 compares / selects per two steps of EVERY launch, more vector
    // instructions than the rest of the loop has, between the matrix-core instructions of a wave (FINDINGS.md 12.8)
    static_assert(NX >= 0 && NX <= 2 && (DIAG || NX == 0), "extra columns ride with the diagonal launches");
    // EXTRA RIGHT-HAND COLUMNS (DIAG launches only; ex0 / ex1, each may be null): besides the Gram block the workgroup sums
    //     X_k[col] = sum_rows (sqrt(w) .* M)[row, col] * ex_k[row]            (the operand as it is staged: weights applied)
    // for the 128 columns of its panel -- with the values a lane holds in registers on their way to LDS (8 multiply-adds per step and lane on
    // the vector pipe, which is idle here; no LDS traffic).  part[g][xoff + k * npan * 128 + pi * 128 + col].  This is how the outer iteration's
    // Jct'd (src/optimize.jl:306: the projection of the step needs it right after the factorisation) and the rank-one term of a view's Gram
    // matrix ride along with the pass that reads the matrix anyway, instead of a GEMV-T pass each.
    // WEIGHTED: w2 holds the SQUARE ROOTS of the weights (gram_impl prepares them) and both operands are scaled: the product stays symmetric, so a
    // diagonal block still needs ONE staged operand (with the weights on one side only it needed two: twice the LDS writes and the LDS
    // footprint -- 4.04 against 3.22 ms for the unweighted kernel at n = 1e7, m = 128, profiles/r04g_streamed_gradients_1e7_128.txt)
    constexpr bool needB = !DIAG;                       // diagonal panel: B is A itself
    // two LDS buffers per operand: step s+1 is written while step s is multiplied -- ONE barrier per step, and no phase in which
    // the matrix cores wait for the staging
    __shared__ double As[2][kPanel][kTLd];
    __shared__ double Bs[needB ? 2 : 1][needB ? kPanel : 1][kTLd];
    // (row group g, pair pr) of this workgroup: ngroups is a multiple of 8; XCD x holds the row groups g = x (mod 8)
    const int np = DIAG ? npan : npan * (npan - 1) / 2;
    const int xslot = (int)blockIdx.x >> 3;
    const int pr = xslot % np, g = (xslot / np) * 8 + ((int)blockIdx.x & 7);
    int pi = pr, pj = pr;
    if (!DIAG) {
        pi = 0;
        while (pj >= npan - 1 - pi) { pj -= npan - 1 - pi; ++pi; }
        pj += pi + 1;
    }
    const int pidx = DIAG ? pr : npan + pr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nsteps = (n + kKStep - 1) / kKStep;
    constexpr int DEPTH = 1;
    // staging role: rows kh, kh + 1 of the step in the four columns c, c + 32, c + 64, c + 96 of the panel -- one load instruction of
    // a wave covers whole 128-byte lines (8 lanes x 16 B per column, 8 columns)
    const int kh = (tid & 7) * 2, c = tid >> 3;
    // software pipeline: the global loads of step s+DEPTH are issued before the MFMAs of step s.  Row weights are applied when a
    // buffer is staged, so that no arithmetic waits on the loads in flight.
    double2 va[DEPTH][4], vb[needB ? DEPTH : 1][4], vw[WEIGHTED ? DEPTH : 1];
    double za[SHIFT ? DEPTH : 1][SHIFT ? 4 : 1], zb[(SHIFT && needB) ? DEPTH : 1][(SHIFT && needB) ? 4 : 1];     // SHIFT: row kh + 2 of the columns
    double2 vs[SHIFT ? DEPTH : 1];
    double2 ve[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};      // the extra columns' entries of the rows kh, kh + 1 of the step in flight
    double xa[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};         // ... and this lane's running sums for its four panel columns
    const double* pa = M + ((int64_t)pi * kPanel + c) * ld + kh;
    const double* pb = M + ((int64_t)pj * kPanel + c) * ld + kh;
    const int na = ncols - pi * kPanel - c, nb = ncols - pj * kPanel - c;      // column c + 32 q of the panel exists iff 32 q < na / nb
    const int64_t cs = 32 * ld;
    // FULL: both panels have all their 128 columns (the usual case: then the loads are unconditional -- the per-column test costs an
    // exec-mask branch and a zero-fill per load in the loop)
    const bool full = ncols - pi * kPanel >= kPanel && ncols - pj * kPanel >= kPanel;
    auto load_step = [&](auto Fc, int buf, int64_t step) {
        constexpr bool FULL = decltype(Fc)::value;
        const int64_t r = step * kKStep;
        if constexpr (WEIGHTED) {       // first: the staging multiplies need it before anything else, and loads return in order
            vw[buf] = ld2(w2 + r + kh);   // (rows >= n: zeros -- SqrtWeightF stores them; no compare / select between the MFMAs)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (FULL) {
                va[buf][q] = ld2(pa + r + q * cs);                       // rows >= n are zero padding
                if constexpr (needB) vb[buf][q] = ld2(pb + r + q * cs);
            } else {
                va[buf][q] = make_double2(0.0, 0.0);
                if (32 * q < na) va[buf][q] = ld2(pa + r + q * cs);
                if constexpr (needB) {
                    vb[buf][q] = make_double2(0.0, 0.0);
                    if (32 * q < nb) vb[buf][q] = ld2(pb + r + q * cs);
                }
            }
        }
        if constexpr (DIAG) {           // (behind the matrix loads: nothing waits for these before the staging of this step.  Entries of rows >= n
            // must be FINITE -- they meet the matrix's zero rows: a caller's vector has zero padding from its allocation, and a column
            // that lives in a scratch slot of the context is written WITH zeros in its pad rows (SignScaleF), whatever an earlier, larger call left
            // there.  Masking here instead -- two compares and selects per column and step between the MFMAs -- cost 0.45 ms of a 3.3 ms pass)
            if constexpr (NX > 0) ve[0] = ld2(ex0 + r + kh);
            if constexpr (NX > 1) ve[1] = ld2(ex1 + r + kh);
        }
        if constexpr (SHIFT) {
            vs[buf] = ld2(sgn + r + kh);
            const bool below = r + kh + 2 < n;       // (the row below the last one is zero by definition -- and may lie outside the allocation)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                za[buf][q] = 0.0;
                if (below && (FULL || 32 * q < na)) za[buf][q] = pa[r + q * cs + 2];
                if constexpr (needB) {
                    zb[buf][q] = 0.0;
                    if (below && (FULL || 32 * q < nb)) zb[buf][q] = pb[r + q * cs + 2];
                }
            }
        }
    };
    auto write_lds = [&](int p, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double2 a = va[buf][q];
            if constexpr (SHIFT) a = make_double2(fma(vs[buf].x, a.y, a.x), fma(vs[buf].y, za[buf][q], a.y));
            if constexpr (WEIGHTED) { a.x *= vw[buf].x; a.y *= vw[buf].y; }
            As[p][c + 32 * q][kh] = a.x;
            As[p][c + 32 * q][kh + 1] = a.y;
            if constexpr (DIAG) {
                if constexpr (NX > 0) xa[0][q] = fma(a.y, ve[0].y, fma(a.x, ve[0].x, xa[0][q]));
                if constexpr (NX > 1) xa[1][q] = fma(a.y, ve[1].y, fma(a.x, ve[1].x, xa[1][q]));
            }
            if constexpr (needB) {
                double2 b = vb[buf][q];
                if constexpr (SHIFT) b = make_double2(fma(vs[buf].x, b.y, b.x), fma(vs[buf].y, zb[buf][q], b.y));
                if constexpr (WEIGHTED) { b.x *= vw[buf].x; b.y *= vw[buf].y; }
                Bs[p][c + 32 * q][kh] = b.x;
                Bs[p][c + 32 * q][kh + 1] = b.y;
            }
        }
    };
    // Row group g takes the steps g, g + ngroups, ... -- except for the SHIFTED operand, whose last row of a step needs the first row of the NEXT
    // step: there a group takes a contiguous chunk of steps, so that this row is in the line the same workgroup loads next anyway (dealt round
    // robin the neighbour step belongs to another workgroup, usually on another XCD: the PMC pass showed 20.6 GB fetched for a 10.3 GB matrix)
    const int64_t chunk = (nsteps + ngroups - 1) / ngroups;
    const int64_t G = SHIFT ? 1 : ngroups;
    const int64_t send = SHIFT ? ((int64_t)(g + 1) * chunk < nsteps ? (int64_t)(g + 1) * chunk : nsteps) : nsteps;
    int64_t step = SHIFT ? (int64_t)g * chunk : g;
    double* out = part + (int64_t)g * part_ld + (int64_t)pidx * (kPanel * kPanel);
    // The MFMA operands of one k-group (4 of the 16 rows of a step): lane (kq = lane / 16, cc = lane % 16) holds row 4 kg + kq of
    // column cc of each 16-column tile it needs.  DIAG (symmetric block, upper tile triangle only -- the host mirrors it, gram_impl):
    // wave w owns tile rows w and 7 - w, (8 - w) + (w + 1) = 9 tiles for every wave, 9/16 of the MFMA work of the square;
    // accumulator k belongs to tile (w, w + k) for k < 8 - w, else to tile (7 - w, k - 1).  Otherwise: tile rows 2w, 2w + 1, all 8 tile
    // columns, accumulator it * 8 + jt.
    // The whole pipeline is instantiated once per wave index for DIAG (WS = 0 .. 3; a switch on the wave picks its copy), so that the
    // tile assignment of a wave is a compile-time fact: with a run-time index every MFMA of the triangle needed two v_cndmask to
    // select its A operand, and those VALU instructions between the MFMAs cost 12 % of the kernel.
    auto pipeline = [&](auto Wc, auto Fc) {
        constexpr int WS = decltype(Wc)::value;
        const int wv = WS >= 0 ? WS : wave;
        constexpr int NB = DIAG ? 9 : 8, NACC = DIAG ? 9 : 16;
        struct Ops { double a0, a1, b[NB]; };
        const int cc = lane & 15, kq = lane >> 4;
        auto read_ops = [&](Ops& o, int p, int kg) {
            const double (*A)[kTLd] = As[p];
            const double (*B)[kTLd] = needB ? Bs[needB ? p : 0] : As[p];
            const int kr = 4 * kg + kq;
            o.a0 = A[(DIAG ? wv : 2 * wv) * 16 + cc][kr];
            o.a1 = A[(DIAG ? 7 - wv : 2 * wv + 1) * 16 + cc][kr];
#pragma unroll
            for (int k = 0; k < NB; ++k) o.b[k] = B[(DIAG ? (k < 8 - wv ? wv + k : k - 1) : k) * 16 + cc][kr];
        };
        f64x4 acc[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = f64x4{0.0, 0.0, 0.0, 0.0};
        auto mfma_ops = [&](const Ops& o) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if constexpr (DIAG) {
                    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(k < 8 - wv ? o.a0 : o.a1, o.b[k], acc[k], 0, 0, 0);
                } else {
                    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0, o.b[k], acc[k], 0, 0, 0);
                    acc[8 + k] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1, o.b[k], acc[8 + k], 0, 0, 0);
                }
            }
        };
        // Pipeline.  LDS: two buffers per operand, step s+1 is written while step s is multiplied.  Registers: the global loads of step
        // s+2 fly during step s; the LDS operand reads run one k-group ahead of the MFMAs (two operand sets), across the step boundary
        // too -- which is why the one barrier of a step sits in its middle: by then every wave has written its share of the next buffer
        // and has issued all its reads of this one (which the next step overwrites).
        Ops o0, o1;
        if (step < send) {
            load_step(Fc, 0, step);
            write_lds(0, 0);
        }
        if (step + G < send) load_step(Fc, 0, step + G);
        __syncthreads();
        if (step < send) read_ops(o0, 0, 0);
        while (step < send) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {          // LDS buffer i holds this step
                if (step >= send) break;
                const bool more = step + G < send;
                read_ops(o1, i, 1);
                mfma_ops(o0);
                if (more) write_lds(i ^ 1, 0);
                if (step + 2 * G < send) load_step(Fc, 0, step + 2 * G);
                read_ops(o0, i, 2);
                mfma_ops(o1);
                read_ops(o1, i, 3);                 // the last read of this buffer: issued before the barrier, after which it may be rewritten
                __syncthreads();
                mfma_ops(o0);
                if (more) read_ops(o0, i ^ 1, 0);
                mfma_ops(o1);
                step += G;
            }
        }
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            int it, jt;
            if constexpr (DIAG) {
                const bool first = k < 8 - wv;
                it = first ? wv : 7 - wv;
                jt = first ? wv + k : k - 1;
            } else {
                it = 2 * wv + k / 8;
                jt = k % 8;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(jt * 16 + cc) * kPanel + it * 16 + kq + 4 * r] = acc[k][r];
        }
    };
    auto run = [&](auto Fc) {
        if constexpr (DIAG) {
            switch (wave) {
                case 0: pipeline(std::integral_constant<int, 0>{}, Fc); break;
                case 1: pipeline(std::integral_constant<int, 1>{}, Fc); break;
                case 2: pipeline(std::integral_constant<int, 2>{}, Fc); break;
                default: pipeline(std::integral_constant<int, 3>{}, Fc); break;
            }
        } else {
            pipeline(std::integral_constant<int, -1>{}, Fc);
        }
    };
    if (full) run(std::true_type{});
    else run(std::false_type{});
    if constexpr (DIAG) {               // the extra columns: sum over the 8 lanes (rows kh = 0, 2 .. 14) that share a panel column, fixed order
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= NX) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double sx = xa[k][q];
                sx += __shfl_xor(sx, 1);
                sx += __shfl_xor(sx, 2);
                sx += __shfl_xor(sx, 4);
                if ((tid & 7) == 0) part[(int64_t)g * part_ld + xoff + ((int64_t)k * npan + pi) * kPanel + c + 32 * q] = sx;
            }
        }
    }
}

// Out[row0 + r, c0 + c] = sum_k In[row0 + r, k] * W[k, c0 + c] for any shape: one workgroup per (128-row tile, 128-column panel).
// The contraction is computed transposed (i = output column c, j = matrix row r) so that each accumulator register stores 16
// consecutive matrix rows (128 contiguous bytes per column); wave w owns the column tiles 2w, 2w + 1 and all 8 row tiles.
// Wd is W laid out for staging: row-major [kpad][rpad] (k padded to 16, columns to 128, zero filled) -- the loads of a step are
// unconditional and fully coalesced, and both operands go to LDS with one 16-byte store per lane on consecutive addresses.
// Same pipeline as gram_kernel: two LDS buffers per operand, the global loads of step s+2 in flight during step s, the operand
// reads one k-group ahead of the MFMAs, one barrier per step (in its middle).
// The column panels of one row tile are neighbours in dispatch order on one XCD (linear id L -> XCD L % 8), so the tile of In
// comes from HBM once and from that XCD's L2 for the other panels (panel-major order read In once per panel: 4x at m = 512).
__global__ __launch_bounds__(kThreads, 2) void rmul_kernel(const double* __restrict__ In, int64_t ld_in, int64_t n, int kcols,
                                                            const double* __restrict__ Wd, int rpad, int rcols, double* __restrict__ Out,
                                                            int64_t ld_out, int64_t ntiles, int ncp) {
    __shared__ __attribute__((aligned(16))) double Ws[2][kKStep][kLdsLd];   // Ws[.][k][c] = W[k0+k, c0+c]
    __shared__ __attribute__((aligned(16))) double Bs[2][kKStep][kLdsLd];   // Bs[.][k][r] = In[row0+r, k0+k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cc = lane & 15, kq = lane >> 4;
    const int xslot = (int)blockIdx.x >> 3;
    const int cp = xslot % ncp;
    const int64_t tile = (int64_t)(xslot / ncp) * 8 + ((int)blockIdx.x & 7);
    if (tile >= ntiles) return;
    const int64_t row0 = tile * kPanel;
    const int c0 = cp * kPanel;
    const int nk = (kcols + kKStep - 1) / kKStep;
    // staging roles: W: rows wk + 4 j of the step, columns wc, wc + 1;  In: column bk of the step, rows br + 32 j, br + 32 j + 1
    const int wk = tid >> 6, wc = (tid & 63) * 2;
    const int bk = tid >> 4, br = (tid & 15) * 2;
    double2 vw[4], vb[4];
    auto load_step = [&](int s) {
        const double* pw = Wd + (int64_t)(s * kKStep + wk) * rpad + c0 + wc;
        const int kc = (s * kKStep + bk < kcols) ? (s * kKStep + bk) : (kcols - 1);     // clamped: rows >= kcols of Wd are zero
        const double* pb = In + (int64_t)kc * ld_in + row0 + br;                         // rows >= n are zero padding
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vw[j] = ld2(pw + (int64_t)(4 * j) * rpad);
            vb[j] = ld2(pb + 32 * j);
        }
    };
    auto write_lds = [&](int p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<double2*>(&Ws[p][wk + 4 * j][wc]) = vw[j];
            *reinterpret_cast<double2*>(&Bs[p][bk][br + 32 * j]) = vb[j];
        }
    };
    struct Ops { double a0, a1, b[8]; };
    auto read_ops = [&](Ops& o, int p, int kg) {
        const int kr = 4 * kg + kq;
        o.a0 = Ws[p][kr][(2 * wave) * 16 + cc];
        o.a1 = Ws[p][kr][(2 * wave + 1) * 16 + cc];
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) o.b[jt] = Bs[p][kr][jt * 16 + cc];
    };
    f64x4 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto mfma_ops = [&](const Ops& o) {
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            acc[0][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0, o.b[jt], acc[0][jt], 0, 0, 0);
            acc[1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1, o.b[jt], acc[1][jt], 0, 0, 0);
        }
    };
    Ops o0, o1;
    if (nk > 0) {
        load_step(0);
        write_lds(0);
    }
    if (nk > 1) load_step(1);
    __syncthreads();
    if (nk > 0) read_ops(o0, 0, 0);
    int s = 0;
    while (s < nk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {          // LDS buffer i holds step s
            if (s >= nk) break;
            const bool more = s + 1 < nk;
            read_ops(o1, i, 1);
            mfma_ops(o0);
            if (more) write_lds(i ^ 1);
            if (s + 2 < nk) load_step(s + 2);
            read_ops(o0, i, 2);
            mfma_ops(o1);
            read_ops(o1, i, 3);                 // the last read of this buffer, issued before the barrier (the next step rewrites it)
            __syncthreads();
            mfma_ops(o0);
            if (more) read_ops(o0, i ^ 1, 0);
            mfma_ops(o1);
            ++s;
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = c0 + (2 * wave + it) * 16 + kq + 4 * r;
            if (col >= rcols) continue;
#pragma unroll
            for (int jt = 0; jt < 8; ++jt) {
                const int64_t row = row0 + jt * 16 + cc;
                if (row < n) Out[(int64_t)col * ld_out + row] = acc[it][jt][r];
            }
        }
}

// Out[:, :rcols] = In[:, :kcols] * W for kcols <= 132, rcols <= 144 -- the shape of the tangent setup at m <= 132.  W (zero
// padded) stays in LDS for the workgroup's lifetime; the grid is persistent (one workgroup per CU, 144 KB of LDS),
// each workgroup a contiguous balanced span of 128-row tiles.  A wave owns 32 rows of every tile and all 128 output
// columns (16 accumulators), and streams its rows of In straight from global memory in the MFMA B-operand layout
// (16 consecutive rows x 4 columns per instruction: 128-byte segments), 16 k-groups ahead through a register ring that
// runs on across tile boundaries: no staging of In through LDS and no barrier in the loop.
// NG k-groups of 4 (kcols <= 4*NG), NI output-column tiles of 16 (rcols <= 16*NI), RING | NG: <32, 8, 16> is the 128 x 128
// case, <33, 9, 11> covers m = 129 .. 132 (one slack / ball column more than 128) with 152 KB of LDS.
template <int NG, int NI, int RING, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rmul_resident_kernel(const double* __restrict__ In, int64_t ld_in, int64_t n, int kcols,
                                                                  const double* __restrict__ W, int ldw, int rcols,
                                                                  double* __restrict__ Out, int64_t ld_out, int64_t ntiles) {
    static_assert(NG % RING == 0, "the register ring must divide the k-groups of a tile");
    static_assert(WAVES == 4 || WAVES == 8, "a wave owns 32 or 16 rows of the 128-row tile");
    constexpr int HN = 8 / WAVES;            // 16-row halves per wave
    __shared__ double Ws[4 * NG][kLdsLd];    // Ws[k][c] = W[k, c]
    for (int idx = threadIdx.x; idx < 4 * NG * 16 * NI; idx += 64 * WAVES) {
        const int k = idx % (4 * NG), c = idx / (4 * NG);
        Ws[k][c] = (k < kcols && c < rcols) ? W[(int64_t)c * ldw + k] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    const int64_t q = ntiles / gridDim.x, rem = ntiles % gridDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * q + ((int64_t)blockIdx.x < rem ? (int64_t)blockIdx.x : rem);
    const int64_t t1 = t0 + q + ((int64_t)blockIdx.x < rem ? 1 : 0);
    if (t0 >= t1) return;
    const int kmax = kcols - 1;
    // B operand of k-group g of tile t: In[row = t*128 + wave*16*HN + h*16 + c, k = 4g + kq] (k clamped: W rows >= kcols are zero)
    auto in_ptr = [&](int64_t t, int g) -> const double* {
        const int k = (4 * g + kq < kmax) ? (4 * g + kq) : kmax;
        return In + (int64_t)k * ld_in + t * kPanel + wave * (16 * HN) + c;
    };
    double bv[HN][RING];
#pragma unroll
    for (int g = 0; g < RING; ++g) {
        const double* p = in_ptr(t0, g);
#pragma unroll
        for (int h = 0; h < HN; ++h) bv[h][g] = __builtin_nontemporal_load(p + 16 * h);
    }
    // the A operands (W from LDS) run one k-group ahead of the MFMAs
    double wa[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) wa[it] = Ws[kq][it * 16 + c];
    for (int64_t t = t0; t < t1; ++t) {
        f64x4 acc[NI][HN];
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
            for (int h = 0; h < HN; ++h) acc[it][h] = f64x4{0.0, 0.0, 0.0, 0.0};
        const int64_t tn = (t + 1 < t1) ? (t + 1) : t;        // the ring reads on into the next tile (or re-reads this one at the end)
#pragma unroll 1
        for (int g8 = 0; g8 < NG; g8 += RING) {
            const bool same = (g8 + RING < NG);                      // the refills of this round stay in tile t
            const int64_t tt = same ? t : tn;
            const int gbase = same ? (g8 + RING) : 0;
#pragma unroll
            for (int slot = 0; slot < RING; ++slot) {
                double v[HN];
#pragma unroll
                for (int h = 0; h < HN; ++h) v[h] = bv[h][slot];
                const double* p = in_ptr(tt, gbase + slot);          // refill the slot with the group RING ahead
#pragma unroll
                for (int h = 0; h < HN; ++h) bv[h][slot] = __builtin_nontemporal_load(p + 16 * h);
                const int gn = (g8 + slot + 1 < NG) ? (g8 + slot + 1) : 0;      // the next k-group (of the next tile after the last)
                double wn[NI];
#pragma unroll
                for (int it = 0; it < NI; ++it) wn[it] = Ws[4 * gn + kq][it * 16 + c];
#pragma unroll
                for (int it = 0; it < NI; ++it)
#pragma unroll
                    for (int h = 0; h < HN; ++h) acc[it][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[it], v[h], acc[it][h], 0, 0, 0);
#pragma unroll
                for (int it = 0; it < NI; ++it) wa[it] = wn[it];
            }
        }
        const int64_t row = t * kPanel + wave * (16 * HN) + c;
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = it * 16 + kq + 4 * r;
                if (col >= rcols) continue;
                double* o = Out + (int64_t)col * ld_out + row;
#pragma unroll
                for (int h = 0; h < HN; ++h) {
                    if (row + 16 * h < n) o[16 * h] = acc[it][h][r];
                }
            }
    }
}

__global__ __launch_bounds__(kThreads) void zero_cols_kernel(double* M, int64_t ld, int64_t n, int c0) {
    double* col = M + (int64_t)(c0 + blockIdx.y) * ld;
    for (int64_t r = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2; r < n; r += (int64_t)gridDim.x * kThreads * 2) {
        if (r + 1 < n) st2(col + r, make_double2(0.0, 0.0));
        else col[r] = 0.0;
    }
}

// GEMV-T producer for a border column of the Gram matrix: v = w2 .* M[:, col]
struct ColTimesW {
    const double* col;
    const double* w2;   // may be null
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        double2 a = ld2(col + r);
        if (w2) {
            const double2 w = ld2(w2 + r);
            a.x *= w.x;
            a.y *= w.y;
        }
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};

// weights of the Gram matrix of a row-scaled view diag(rs) M: out = rs^2 .* w2 (w2 may be null)
struct ViewWeightF {
    const double *rs, *w2;
    double* out;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 s = ld2(rs + i);
        double2 w = make_double2(s.x * s.x, s.y * s.y);
        if (w2) {
            const double2 u = ld2(w2 + i);
            w.x *= u.x;
            w.y *= u.y;
        }
        if (v1) st2(out + i, w);
        else if (v0) out[i] = w.x;
    }
};

// rank-one term of a view's Gram matrix: the producer rs .* w2 .* u of z = A'(D W2 u), and u' W2 u
struct ViewRank1V {
    const double *rs, *w2, *u;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        double2 a = ld2(u + r);
        if (rs) { const double2 s = ld2(rs + r); a.x *= s.x; a.y *= s.y; }
        if (w2) { const double2 w = ld2(w2 + r); a.x *= w.x; a.y *= w.y; }
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};
// the Gram kernel's row factors: out = sqrt(w2)  (weights are squares in every use: Dy^2, phi'^2, the rows of D0^-1; a negative one gives NaN and
// the factorisation reports a non-finite Gram matrix)
struct SqrtWeightF {
    const double* w2;
    double* out;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 w = ld2(w2 + i);
        // (rows >= n of the tile: zeros, see SignScaleF -- the Gram kernel reads the staged weights without a mask)
        st2(out + i, make_double2(v0 ? sqrt(w.x) : 0.0, v1 ? sqrt(w.y) : 0.0));
    }
};
struct ViewRank1DotF {
    const double *w2, *u;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(u + i);
        const double2 w = w2 ? ld2(w2 + i) : make_double2(1.0, 1.0);
        double s = 0.0;
        if (v0) s = fma(a.x * w.x, a.x, s);
        if (v1) s = fma(a.y * w.y, a.y, s);
        red[0] += s;
    }
};

// Extra right-hand columns of a Gram pass (at most two): X[:, k] = M' (sqrt(w2) .* e_k), ncols_all x nx, column-major, replicated (all-reduced)
// -- summed by the Gram kernel from the operand values it stages anyway (gram_kernel: "EXTRA RIGHT-HAND COLUMNS").  With weights the
// column enters in the kernel's scaled space on purpose: the uses are M'(sx .* dx + sy .* dy) of the bound-stacked projection (sqrt(w2) = |Dy|,
// e = |Dy| dx - Dx sgn(Dy) dy: no division by a weight that may be zero) and the unweighted M'd (w2 == nullptr).
struct GramRhs {
    int nx = 0;
    const double* e[2] = {nullptr, nullptr};     // device n-vectors
    std::vector<double>* X = nullptr;
};
// out = sgn(rs) .* (sw ? sqrt(sw) : 1) .* v   (rs, sw optional): a column moved into the scaled space of the plain matrix behind a view
struct SignScaleF {
    const double *v, *rs, *sw;
    double* out;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        double2 a = ld2(v + i);
        if (rs) {
            const double2 s = ld2(rs + i);
            a.x = s.x > 0.0 ? a.x : (s.x < 0.0 ? -a.x : 0.0);
            a.y = s.y > 0.0 ? a.y : (s.y < 0.0 ? -a.y : 0.0);
        }
        if (sw) {
            const double2 w = ld2(sw + i);
            a.x *= sqrt(w.x);
            a.y *= sqrt(w.y);
        }
        // rows >= n of the tile get ZEROS (out is a scratch slot of round_up(n + 1, 2048) doubles: the vector kernel's 512-row tiles end inside it):
        // the Gram kernel reads the column in 16-row steps without a mask, and a stale NaN there would meet the matrix's zero rows as 0 * NaN
        st2(out + i, make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0));
    }
};
// u' (sqrt(w2) .* e) for up to two columns e (the rank-one term of a view against the extra columns)
struct ViewRhsDotF {
    const double *u, *w2, *e0, *e1;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        double2 a = ld2(u + i);
        if (w2) { const double2 w = ld2(w2 + i); a.x *= sqrt(w.x); a.y *= sqrt(w.y); }
        const double2 x0 = ld2(e0 + i), x1 = e1 ? ld2(e1 + i) : make_double2(0.0, 0.0);
        double s0 = 0.0, s1 = 0.0;
        if (v0) { s0 = a.x * x0.x; s1 = a.x * x1.x; }
        if (v1) { s0 = fma(a.y, x0.y, s0); s1 = fma(a.y, x1.y, s1); }
        red[0] += s0;
        red[1] += s1;
    }
};
// GEMV-T producer sqrt(w2) .* e (w2 optional): the extra columns against the border columns of the matrix
struct SqrtWTimesV {
    const double *e, *w2;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        double2 a = ld2(e + r);
        if (w2) { const double2 w = ld2(w2 + r); a.x *= sqrt(w.x); a.y *= sqrt(w.y); }
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};

static int gram_impl(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols_all, const double* w2, std::vector<double>& G, const GramRhs* rhs = nullptr,
                     const double* shift_sgn = nullptr) {
    // shift_sgn != NULL: the Gram matrix of R, R_i = M_i + shift_sgn_i M_{i+1} (gram_kernel SHIFT) -- plain matrix, weights given, no extra columns
    if (shift_sgn && (M->view || !w2 || rhs)) return set_err(ctx, LFPSQP_ERR_ARG, "shifted Gram matrix: plain matrix with weights, no right-hand columns");
    G.assign((size_t)ncols_all * ncols_all, 0.0);
    const int nxu = rhs ? rhs->nx : 0;
    if (rhs && rhs->X) rhs->X->assign((size_t)ncols_all * nxu, 0.0);
    if (ncols_all == 0) return 0;
    if (M->view) {
        // V = D A + u w' (D = diag(rs)):  V' W2 V = A' (D W2 D) A + z w' + w z' + (u' W2 u) w w',  z = A' (D W2 u)
        // -- the weighted kernel over the plain storage; z rides along as an extra right-hand column of that pass (in the kernel's scaled space,
        // weights rs^2 w2: sgn(rs) sqrt(w2) u), as do the caller's own columns (sgn(rs) e_k, plus w (u' sqrt(w2) e_k) for the rank-one term); one
        // dot-product kernel, assembled on the host
        const lfpsqp_mat plain = M->plain();
        const size_t npad = (size_t)round_up(M->n + 1, kPadRows);
        LF_TRY(ensure_nvec(ctx, 4 * npad));       // [combined weights | their square roots (the plain call below) | two columns in scaled space]
        const double* wts = w2;
        if (M->rs) {
            LF_TRY((run_vec<ViewWeightF, 0, NoPost>(ctx, M->n, ViewWeightF{M->rs, w2, ctx->d_nvec}, 0u, nullptr, NoPost())));
            wts = ctx->d_nvec;
        }
        GramRhs inner;
        std::vector<double> Xin;
        inner.X = &Xin;
        for (int k = 0; k < nxu; ++k) {
            if (!M->rs) { inner.e[inner.nx++] = rhs->e[k]; continue; }
            double* dst = ctx->d_nvec + (2 + inner.nx) * npad;
            LF_TRY((run_vec<SignScaleF, 0, NoPost>(ctx, M->n, SignScaleF{rhs->e[k], M->rs, nullptr, dst}, 0u, nullptr, NoPost())));
            inner.e[inner.nx++] = dst;
        }
        const bool z_rides = M->ru && inner.nx < 2;
        if (z_rides) {
            if (!M->rs && !w2) inner.e[inner.nx++] = M->ru;
            else {
                double* dst = ctx->d_nvec + (2 + inner.nx) * npad;
                LF_TRY((run_vec<SignScaleF, 0, NoPost>(ctx, M->n, SignScaleF{M->ru, M->rs, w2, dst}, 0u, nullptr, NoPost())));
                inner.e[inner.nx++] = dst;
            }
        }
        LF_TRY(gram_impl(ctx, &plain, ncols_all, wts, G, inner.nx > 0 ? &inner : nullptr));
        const int mm = ncols_all;
        if (rhs && rhs->X)
            for (int k = 0; k < nxu; ++k)
                for (int i = 0; i < mm; ++i) (*rhs->X)[(size_t)k * mm + i] = Xin[(size_t)k * mm + i];
        if (!M->ru) return 0;
        LF_TRY(ensure_mvec(ctx, (size_t)2 * mm + 16));
        double* dz = ctx->d_m;                                   // [z (mm) ; u' W2 u ; w (mm) ; u' sqrt(W2) e_k (2)] -> host
        if (!z_rides) LF_TRY(run_gemv_t(ctx, &plain, mm, M->n, ViewRank1V{M->rs, w2, M->ru}, dz));
        LF_TRY((run_vec<ViewRank1DotF, 1, NoPost>(ctx, M->n, ViewRank1DotF{w2, M->ru}, 0u, dz + mm, NoPost())));
        LF_HIP(ctx, hipMemcpyAsync(dz + mm + 1, M->rw, sizeof(double) * mm, hipMemcpyDeviceToDevice, ctx->stream));
        if (nxu > 0)
            LF_TRY((run_vec<ViewRhsDotF, 2, NoPost>(ctx, M->n, ViewRhsDotF{M->ru, w2, rhs->e[0], nxu > 1 ? rhs->e[1] : nullptr}, 0u, dz + 2 * mm + 1, NoPost())));
        LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, dz, sizeof(double) * (2 * mm + 3), hipMemcpyDeviceToHost, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double* z = z_rides ? &Xin[(size_t)(inner.nx - 1) * mm] : ctx->h_m;
        const double uu = ctx->h_m[mm];
        const double* w = ctx->h_m + mm + 1;
        for (int jj = 0; jj < mm; ++jj)
            for (int ii = 0; ii < mm; ++ii) G[(size_t)jj * mm + ii] += z[ii] * w[jj] + w[ii] * z[jj] + uu * w[ii] * w[jj];
        if (rhs && rhs->X)
            for (int k = 0; k < nxu; ++k)
                for (int i = 0; i < mm; ++i) (*rhs->X)[(size_t)k * mm + i] += w[i] * ctx->h_m[2 * mm + 1 + k];
        return 0;
    }
    // A few columns beyond a multiple of the 128-column panel (m + 1 constraints with a slack/ball column, say) would cost
    // a whole extra panel row and column of MFMA tiles; they are cheaper as GEMV-T passes: G[:, j] = M' (w2 .* M[:, j]).
    const int rem = ncols_all % kPanel;
    const int border = (!shift_sgn && ncols_all > kPanel && rem > 0 && rem <= 4) ? rem : 0;        // (the GEMV-T passes of the border know M only)
    const int ncols = ncols_all - border;
    const int npan = (ncols + kPanel - 1) / kPanel;
    const int npair = npan * (npan + 1) / 2;
    const int64_t pp = (int64_t)npair * kPanel * kPanel;
    // border columns ride with the pass as extra right-hand columns while slots are free (two in all): G[0:ncols, j] = (sqrt(w) M)'(sqrt(w) M_j);
    // their rows against each other come from a GEMV-T over the border columns alone (below)
    const int nride = std::min(border, 2 - nxu);
    const int nslots = nxu + nride;
    const int xcols = nslots * npan * kPanel;            // the extra columns' sums sit behind the pair slots of a partial row
    const int64_t pld = pp + xcols;
    const int64_t nsteps = (M->n + kKStep - 1) / kKStep;
#ifndef LFPSQP_GRAM_WGS
#define LFPSQP_GRAM_WGS 2
#endif
    // row groups per launch: as many as fill the device with all pairs of the launch resident at once (two workgroups per CU), a
    // multiple of 8 (one XCD each, see gram_kernel)
    const int64_t slots = LFPSQP_GRAM_WGS * (int64_t)(ctx->num_cu > 0 ? ctx->num_cu : 128);
    auto groups_for = [&](int np) -> int {
        int64_t gcount = slots / np / 8 * 8;
        const int64_t need = (nsteps + 7) / 8 * 8;
        if (gcount > need) gcount = need;
        return (int)(gcount < 8 ? 8 : gcount);
    };
    const int noff = npair - npan;
    const int gd = groups_for(npan), go = noff > 0 ? groups_for(noff) : 0;
    LF_TRY(ensure_part(ctx, (size_t)std::max(gd, go) * pld));
    LF_TRY(ensure_small(ctx, (size_t)pld));
    const int64_t tile2 = (int64_t)kPanel * kPanel;
    const double* exs[2] = {nxu > 0 ? rhs->e[0] : nullptr, nxu > 1 ? rhs->e[1] : nullptr};
    if (nride > 0) {
        const size_t npad = (size_t)round_up(M->n + 1, kPadRows);
        if (w2) {                                        // the column in the kernel's scaled space: sqrt(w2) .* M[:, j]  (scratch slots 2, 3 of d_nvec)
            const bool own = ctx->d_nvec && w2 == ctx->d_nvec;
            LF_TRY(ensure_nvec(ctx, 4 * npad));
            if (own) w2 = ctx->d_nvec;
        }
        for (int b = 0; b < nride; ++b) {
            const double* colp = M->p + (int64_t)(ncols + b) * M->ld;
            if (!w2) { exs[nxu + b] = colp; continue; }
            double* dst = ctx->d_nvec + (2 + nxu + b) * npad;
            if (dst == exs[0] || dst == exs[1]) dst = ctx->d_nvec + (2 + ((nxu + b + 1) & 1)) * npad;      // (a view's own column may already sit in a slot)
            LF_TRY((run_vec<SignScaleF, 0, NoPost>(ctx, M->n, SignScaleF{colp, nullptr, w2, dst}, 0u, nullptr, NoPost())));
            exs[nxu + b] = dst;
        }
    }
    const double* ex0 = exs[0];
    const double* ex1 = exs[1];
    if (w2) {
        // the kernel scales BOTH operands by sqrt(w2): staged in the second half of the n-vector scratch (the first may hold the weights themselves)
        const size_t npad = (size_t)round_up(M->n + 1, kPadRows);
        const bool own = ctx->d_nvec && w2 == ctx->d_nvec;              // (a view's combined weights live in the first half already)
        LF_TRY(ensure_nvec(ctx, 2 * npad));
        if (own) w2 = ctx->d_nvec;
        double* sw = ctx->d_nvec + npad;
        LF_TRY((run_vec<SqrtWeightF, 0, NoPost>(ctx, M->n, SqrtWeightF{w2, sw}, 0u, nullptr, NoPost())));
        LF_TRY(ensure_part(ctx, (size_t)std::max(gd, go) * pld));         // (run_vec may not shrink it, but keep the reservation next to its use)
        if (shift_sgn) {
            hipLaunchKernelGGL((gram_kernel<true, true, true>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, sw, ctx->part, pld, nullptr, nullptr, pp, shift_sgn);
            if (noff > 0)
                hipLaunchKernelGGL((gram_kernel<false, true, true>), dim3(go * noff), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, go, sw, ctx->part, pld, nullptr, nullptr, pp, shift_sgn);
        } else {
            if (nslots == 2) hipLaunchKernelGGL((gram_kernel<true, true, false, 2>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, sw, ctx->part, pld, ex0, ex1, pp, nullptr);
            else if (nslots == 1) hipLaunchKernelGGL((gram_kernel<true, true, false, 1>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, sw, ctx->part, pld, ex0, ex1, pp, nullptr);
            else hipLaunchKernelGGL((gram_kernel<true, true, false, 0>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, sw, ctx->part, pld, ex0, ex1, pp, nullptr);
            if (noff > 0)
                hipLaunchKernelGGL((gram_kernel<false, true>), dim3(go * noff), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, go, sw, ctx->part, pld, nullptr, nullptr, pp, nullptr);
        }
    } else {
        if (nslots == 2) hipLaunchKernelGGL((gram_kernel<true, false, false, 2>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, w2, ctx->part, pld, ex0, ex1, pp, nullptr);
        else if (nslots == 1) hipLaunchKernelGGL((gram_kernel<true, false, false, 1>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, w2, ctx->part, pld, ex0, ex1, pp, nullptr);
        else hipLaunchKernelGGL((gram_kernel<true, false, false, 0>), dim3(gd * npan), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, gd, w2, ctx->part, pld, ex0, ex1, pp, nullptr);
        if (noff > 0)
            hipLaunchKernelGGL((gram_kernel<false, false>), dim3(go * noff), dim3(kThreads), 0, ctx->stream, M->p, M->ld, M->n, ncols, npan, go, w2, ctx->part, pld, nullptr, nullptr, pp, nullptr);
    }
    LF_LAUNCH_CHECK(ctx);
    // reduce the partials of each launch over its row groups (32 columns per workgroup)
    hipLaunchKernelGGL((reduce_rows_kernel<NoPost>), dim3((unsigned)((npan * tile2 + 31) / 32), 1), dim3(1024), 0, ctx->stream, ctx->part,
                       (int64_t)gd, (int)(npan * tile2), (int)pld, 0u, ctx->small, 0, (int64_t)gd, 5, NoPost());
    LF_LAUNCH_CHECK(ctx);
    if (xcols > 0) {                          // (the diagonal launch's row groups carry the extra columns)
        hipLaunchKernelGGL((reduce_rows_kernel<NoPost>), dim3((unsigned)((xcols + 31) / 32), 1), dim3(1024), 0, ctx->stream, ctx->part + pp,
                           (int64_t)gd, xcols, (int)pld, 0u, ctx->small + pp, 0, (int64_t)gd, 5, NoPost());
        LF_LAUNCH_CHECK(ctx);
    }
    if (noff > 0) {
        hipLaunchKernelGGL((reduce_rows_kernel<NoPost>), dim3((unsigned)((noff * tile2 + 31) / 32), 1), dim3(1024), 0, ctx->stream,
                           ctx->part + npan * tile2, (int64_t)go, (int)(noff * tile2), (int)pld, 0u, ctx->small + npan * tile2, 0, (int64_t)go, 5,
                           NoPost());
        LF_LAUNCH_CHECK(ctx);
    }
    LF_TRY(allreduce_dev(ctx, ctx->small, pld, 0));
    // (into the context's PINNED staging block: a copy to pageable memory goes through the runtime's own staging and costs 0.1 ms more)
    LF_TRY(ensure_mvec(ctx, (size_t)pld + 8));
    LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->small, sizeof(double) * pld, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double* const hpin = ctx->h_m;
    struct { const double* p; const double* data() const { return p; } } h{hpin};
    {
        size_t poff = npan;                     // slots of the pairs pi < pj follow the npan diagonal ones
        for (int pi = 0; pi < npan; ++pi)
            for (int pj = pi; pj < npan; ++pj) {
                const size_t pidx = (pi == pj) ? (size_t)pi : poff++;
                const double* blk = h.data() + pidx * kPanel * kPanel;
                for (int j = 0; j < kPanel; ++j) {
                    const int gj = pj * kPanel + j;
                    if (gj >= ncols) break;
                    for (int i = 0; i < kPanel; ++i) {
                        const int gi = pi * kPanel + i;
                        if (gi >= ncols) break;
                        // a diagonal block holds its tiles with (i / 16) <= (j / 16) only: the others are their mirror images
                        const double v = (pi == pj && i / 16 > j / 16) ? blk[i * kPanel + j] : blk[j * kPanel + i];
                        G[(size_t)gj * ncols_all + gi] = v;
                        if (pi != pj) G[(size_t)gi * ncols_all + gj] = v;                        // the mirrored block
                    }
                }
            }
    }
    if (rhs && rhs->X)
        for (int k = 0; k < nxu; ++k)
            for (int i = 0; i < ncols; ++i) (*rhs->X)[(size_t)k * ncols_all + i] = hpin[pp + (size_t)k * npan * kPanel + i];
    // (hpin is reused by the copies below: the riding border columns' sums are taken out first)
    std::vector<double> ride((size_t)nride * ncols);
    for (int b = 0; b < nride; ++b)
        for (int i = 0; i < ncols; ++i) ride[(size_t)b * ncols + i] = hpin[pp + (size_t)(nxu + b) * npan * kPanel + i];
    if (border > 0) {
        LF_TRY(ensure_mvec(ctx, (size_t)ncols_all + 8));
        lfpsqp_mat bcols = *M;                           // the border columns as a matrix of their own
        bcols.p = M->p + (int64_t)ncols * M->ld;
        bcols.m = border;
        for (int j = ncols; j < ncols_all; ++j) {
            const bool rides = j - ncols < nride;
            if (rides) {                                 // the panel part came with the Gram pass; the border block from a pass over the border columns alone
                LF_TRY(run_gemv_t(ctx, &bcols, border, M->n, ColTimesW{M->p + (int64_t)j * M->ld, w2}, ctx->d_m + ncols));
                LF_HIP(ctx, hipMemcpyAsync(ctx->h_m + ncols, ctx->d_m + ncols, sizeof(double) * border, hipMemcpyDeviceToHost, ctx->stream));
                LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
                for (int i = 0; i < ncols; ++i) ctx->h_m[i] = ride[(size_t)(j - ncols) * ncols + i];
            } else {
                LF_TRY(run_gemv_t(ctx, M, ncols_all, M->n, ColTimesW{M->p + (int64_t)j * M->ld, w2}, ctx->d_m));
                LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->d_m, sizeof(double) * ncols_all, hipMemcpyDeviceToHost, ctx->stream));
                LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            for (int i = 0; i < ncols_all; ++i) G[(size_t)j * ncols_all + i] = G[(size_t)i * ncols_all + j] = ctx->h_m[i];
        }
        // the extra columns against the border columns: a GEMV-T over those few columns alone
        lfpsqp_mat bm = *M;
        bm.p = M->p + (int64_t)ncols * M->ld;
        bm.m = border;
        for (int k = 0; k < nxu; ++k) {
            LF_TRY(run_gemv_t(ctx, &bm, border, M->n, SqrtWTimesV{rhs->e[k], w2}, ctx->d_m));
            LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->d_m, sizeof(double) * border, hipMemcpyDeviceToHost, ctx->stream));
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (rhs->X)
                for (int i = 0; i < border; ++i) (*rhs->X)[(size_t)k * ncols_all + ncols + i] = ctx->h_m[i];
        }
    }
    // G is symmetric up to rounding of the two summation orders; symmetrise
    for (int j = 0; j < ncols_all; ++j)
        for (int i = 0; i < j; ++i) {
            const double v = 0.5 * (G[(size_t)j * ncols_all + i] + G[(size_t)i * ncols_all + j]);
            G[(size_t)j * ncols_all + i] = G[(size_t)i * ncols_all + j] = v;
        }
    return 0;
}

// Out[:, j] = rs .* Out[:, j] + u c_j in place, j < m (the product of a view: the MFMA kernels ran over the plain storage; c = W'w)
__global__ __launch_bounds__(kThreads) void view_rows_kernel(double* D, int64_t ld, int64_t n, int m, const double* __restrict__ rs, const double* __restrict__ u,
                                                             const double* __restrict__ c) {
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2;
    if (i >= n) return;
    const double2 s = rs ? ld2(rs + i) : make_double2(1.0, 1.0);
    const double2 uu = u ? ld2(u + i) : make_double2(0.0, 0.0);
    const bool v1 = i + 1 < n;
    const int j0 = blockIdx.y * 16, j1 = (j0 + 16 < m) ? j0 + 16 : m;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const double2 a = ld2(D + (int64_t)j * ld + i);
        const double cj = u ? c[j] : 0.0;
        const double2 o = make_double2(fma(uu.x, cj, s.x * a.x), fma(uu.y, cj, s.y * a.y));
        if (v1) st2(D + (int64_t)j * ld + i, o);
        else D[(int64_t)j * ld + i] = o.x;
    }
}

static int rmul_impl(lfpsqp_ctx* ctx, const lfpsqp_mat* In, int kcols, const double* W_host, int rcols, lfpsqp_mat* Out) {
    if (rcols == 0 || In->n == 0) return 0;
    if (In->view) {    // (D In + u w') W = D (In W) + u (W'w)': a second pass over the output -- this product is off the fast path (the basis of a
                       // view stays in factored form; only the refinement rounds of an ill-conditioned block and callers that insist on Z come here)
        const lfpsqp_mat plain = In->plain();
        LF_TRY(rmul_impl(ctx, &plain, kcols, W_host, rcols, Out));
        double* dc = nullptr;
        if (In->ru) {
            LF_TRY(ensure_mvec(ctx, (size_t)kcols + rcols + 16));
            LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, In->rw, sizeof(double) * kcols, hipMemcpyDeviceToHost, ctx->stream));
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            double* hc = ctx->h_m + kcols + (kcols & 1);
            for (int j = 0; j < rcols; ++j) {
                double acc = 0.0;
                for (int k = 0; k < kcols; ++k) acc += W_host[(size_t)j * kcols + k] * ctx->h_m[k];
                hc[j] = acc;
            }
            dc = ctx->d_m + kcols + (kcols & 1);
            LF_HIP(ctx, hipMemcpyAsync(dc, hc, sizeof(double) * rcols, hipMemcpyHostToDevice, ctx->stream));
        }
        hipLaunchKernelGGL(view_rows_kernel, dim3((unsigned)((In->n + 2 * kThreads - 1) / (2 * kThreads)), (unsigned)((rcols + 15) / 16)), dim3(kThreads), 0,
                           ctx->stream, Out->p, Out->ld, In->n, rcols, In->rs, In->ru, dc);
        LF_LAUNCH_CHECK(ctx);
        if (dc) LF_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (h_m is the context's shared pinned staging block)
        return 0;
    }
    LF_TRY(ensure_small(ctx, (size_t)kcols * rcols + 32 + ((size_t)kcols + kKStep) * ((size_t)rcols + kPanel)));
    LF_HIP(ctx, hipMemcpyAsync(ctx->small, W_host, sizeof(double) * (size_t)kcols * rcols, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));   // W_host is caller-owned pageable memory
    const int64_t ntiles = (In->n + kPanel - 1) / kPanel;
    const int64_t cus = ctx->num_cu > 0 ? ctx->num_cu : 1;
    const dim3 pgrid((unsigned)(ntiles < cus ? ntiles : cus));
#ifndef LFPSQP_RMUL_WAVES
#define LFPSQP_RMUL_WAVES 8
#endif
    constexpr int kRW = LFPSQP_RMUL_WAVES;    // waves of the W-resident kernel: 8 = two per SIMD, each covers the other's tile epilogue
    if (kcols <= kPanel && rcols <= kPanel && ctx->tune_onepass >= 0) {      // W resident in LDS, persistent grid
        hipLaunchKernelGGL((rmul_resident_kernel<32, 8, 16, kRW>), pgrid, dim3(64 * kRW), 0, ctx->stream, In->p, In->ld, In->n, kcols,
                           ctx->small, kcols, rcols, Out->p, Out->ld, ntiles);
    } else if (kcols <= 132 && rcols <= 144 && ctx->tune_onepass >= 0) {     // ... a few columns more (m = 128 + slack / ball)
        hipLaunchKernelGGL((rmul_resident_kernel<33, 9, 11, kRW>), pgrid, dim3(64 * kRW), 0, ctx->stream, In->p, In->ld, In->n, kcols,
                           ctx->small, kcols, rcols, Out->p, Out->ld, ntiles);
    } else {
        // W laid out for staging (rmul_kernel): row-major, k padded to the 16-deep step, columns to the 128-wide panel
        const int ncp = (rcols + kPanel - 1) / kPanel, rpad = ncp * kPanel, kpad = (kcols + kKStep - 1) / kKStep * kKStep;
        std::vector<double> Wd((size_t)kpad * rpad, 0.0);
        for (int c = 0; c < rcols; ++c)
            for (int k = 0; k < kcols; ++k) Wd[(size_t)k * rpad + c] = W_host[(size_t)c * kcols + k];
        double* wdev = ctx->small + (((size_t)kcols * rcols + 15) & ~(size_t)15);
        LF_HIP(ctx, hipMemcpyAsync(wdev, Wd.data(), sizeof(double) * Wd.size(), hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t tgroups = (ntiles + 7) / 8;
        hipLaunchKernelGGL(rmul_kernel, dim3((unsigned)(tgroups * 8 * ncp)), dim3(kThreads), 0, ctx->stream, In->p, In->ld, In->n, kcols,
                           wdev, rpad, rcols, Out->p, Out->ld, ntiles, ncp);
    }
    LF_LAUNCH_CHECK(ctx);
    return 0;
}

// ---------------------------------------------------------------------------
// Host side of the factorisation: the replicated m x m problems (smallla.h: one-sided Jacobi, Cholesky).
// ---------------------------------------------------------------------------
// Eigen-decomposition of a Gram matrix G = V diag(sig^2) V' (sig descending).  G positive definite (the usual case):
// G = L L', and one-sided Jacobi on the columns of L turns them into sig_j v_j -- the singular values themselves (not their
// squares) and the eigenvectors, with no rotation accumulator.  A failed Cholesky (numerically singular G) takes Jacobi on G.
// A (rows x cols) = U diag(S) V' by one-sided Jacobi: on the device (jacobi.hip; with want_v the rotations are accumulated
// by carrying an identity below A) from kDevJacobiMinCols columns on, else -- and for shapes the kernels do not cover -- by
// the host routine of smallla.h.  Same contract as jacobi_svd: S descending, U = normalised columns, V orthogonal.
constexpr int kDevJacobiMinCols = 64;
static void small_svd(lfpsqp_ctx* ctx, int rows, int cols, const std::vector<double>& A, std::vector<double>& U, std::vector<double>& S,
                      std::vector<double>& V, bool want_v = true) {
    const int rows_all = want_v ? rows + cols : rows;
    std::vector<double> X;
    bool dev = ctx && cols >= kDevJacobiMinCols && rows_all <= 1024 && ctx->tune_onepass >= 0;
#ifdef LFPSQP_HIP_EMULATED
    // (CPU emulator of the tests only: the block rounds of the wide shapes run ~100 x slower there than the host routine -- 12 s per factorisation
    // at m = 300 -- so they are emulated on request: tests/test_capi_parity.py::test_small_svd_one_sided_jacobi sets the variable)
    if (dev && rows_all > 256 && !getenv("LFPSQP_EMU_DEVICE_JACOBI")) dev = false;
#endif
    if (dev) {
        X.assign((size_t)rows_all * cols, 0.0);
        for (int j = 0; j < cols; ++j) {
            for (int i = 0; i < rows; ++i) X[(size_t)j * rows_all + i] = A[(size_t)j * rows + i];
            if (want_v) X[(size_t)j * rows_all + rows + j] = 1.0;
        }
        dev = device_jacobi(ctx, rows, rows_all, cols, X);
    }
    if (!dev) {
        jacobi_svd(rows, cols, A, U, S, V, want_v);
        return;
    }
    std::vector<double> nrm(cols);
    for (int j = 0; j < cols; ++j) {
        double sq = 0.0;
        for (int i = 0; i < rows; ++i) sq += X[(size_t)j * rows_all + i] * X[(size_t)j * rows_all + i];
        nrm[j] = sqrt(sq);
    }
    std::vector<int> idx(cols);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return nrm[a] > nrm[b]; });
    U.assign((size_t)rows * cols, 0.0);
    S.assign(cols, 0.0);
    V.assign(want_v ? (size_t)cols * cols : 0, 0.0);
    for (int jj = 0; jj < cols; ++jj) {
        const int j = idx[jj];
        S[jj] = nrm[j];
        for (int i = 0; i < rows; ++i) U[(size_t)jj * rows + i] = nrm[j] > 0 ? X[(size_t)j * rows_all + i] / nrm[j] : 0.0;
        for (int i = 0; want_v && i < cols; ++i) V[(size_t)jj * cols + i] = X[(size_t)j * rows_all + rows + i];
    }
}

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static const bool kTraceFactorize = getenv("LFPSQP_TRACE_FACTORIZE") != nullptr;     // development: phase times on stderr

// Warm start of the eigenproblem of G = L L' (lfpsqp_factorize_hint): V0 = the eigenvectors of a NEARBY Gram matrix -- the previous outer
// iteration's (the constraint gradients move little between iterations; with linear constraints and no bounds, not at all).  The one-sided
// Jacobi then runs on X = L' V0, whose columns are already nearly orthogonal (X'X = V0' G V0): one or two sweeps instead of eight on the
// clustered spectra of the BASELINE configs.  X R = Q S  =>  G = (V0 R) S^2 (V0 R)', and V = V0 R = L^-T (Q S) by back substitution -- R is
// never accumulated.  Used only when V0 is orthogonal to 1e-8 (checked: a rank-deficient or foreign hint is ignored) and m <= 256.
static bool gram_eig_warm(lfpsqp_ctx* ctx, int m, const std::vector<double>& Lc, const std::vector<double>& V0, std::vector<double>& sig, std::vector<double>& V) {
    if (m < 2 || m > 256 || (int)V0.size() != m * m) return false;
    // orthogonality of the hint, probed with two fixed vectors: |V0'(V0 z) - z| <= 1e-8 |z| (4 m^2 flops; if V0'V0 = I + E, this is |E z|)
    {
        std::vector<double> z(m), t(m), b(m);
        for (int probe = 0; probe < 2; ++probe) {
            uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(probe + 1);
            double zz = 0.0;
            for (int i = 0; i < m; ++i) {
                h ^= h >> 12; h ^= h << 25; h ^= h >> 27;
                z[i] = (double)((h * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
                zz += z[i] * z[i];
            }
            std::fill(t.begin(), t.end(), 0.0);
            for (int j = 0; j < m; ++j) {               // t = V0 z
                const double* c = &V0[(size_t)j * m];
                const double zj = z[j];
#pragma omp simd
                for (int i = 0; i < m; ++i) t[i] += c[i] * zj;
            }
            double err = 0.0;
            for (int j = 0; j < m; ++j) {               // b = V0' t
                const double* c = &V0[(size_t)j * m];
                double sdot = 0.0;
#pragma omp simd reduction(+ : sdot)
                for (int i = 0; i < m; ++i) sdot += c[i] * t[i];
                err += (sdot - z[j]) * (sdot - z[j]);
            }
            if (!(err <= 1e-16 * zz)) return false;
        }
    }
    std::vector<double> X((size_t)m * m), Q, none;
    const double t_x = now_ms();
    // (the columns are independent: shared over the host threads -- at m = 128 the two triangular loops of this function are 2 Mflop of
    // latency-bound scalar work between the Gram kernel and the next pass over the matrix)
    [[maybe_unused]] const int nth = m >= 64 ? small_threads() : 1;
#pragma omp parallel for if (nth > 1) num_threads(nth) schedule(static)
    for (int j = 0; j < m; ++j) {                     // X[:, j] = L' V0[:, j]: (L'v)_i = sum_{k >= i} L[k, i] v_k  (Lc column-major, lower: contiguous in k)
        const double* v = &V0[(size_t)j * m];
        for (int i = 0; i < m; ++i) {
            const double* l = &Lc[(size_t)i * m];
            double sdot = 0.0;
#pragma omp simd reduction(+ : sdot)
            for (int k = i; k < m; ++k) sdot += l[k] * v[k];
            X[(size_t)j * m + i] = sdot;
        }
    }
    const double t_j = now_ms();
    small_svd(ctx, m, m, X, Q, sig, none, false);      // Q: unit columns, sorted by norm (descending); sig: the norms
    if (!(sig[m - 1] > 0.0)) return false;
    const double t_b = now_ms();
    V.assign((size_t)m * m, 0.0);
#pragma omp parallel for if (nth > 1) num_threads(nth) schedule(static)
    for (int j = 0; j < m; ++j) {                     // L' y = sig_j Q[:, j]: back substitution on the upper triangular L'
        double* y = &V[(size_t)j * m];
        for (int i = m - 1; i >= 0; --i) {
            const double* l = &Lc[(size_t)i * m];
            double sdot = 0.0;
#pragma omp simd reduction(+ : sdot)
            for (int k = i + 1; k < m; ++k) sdot += l[k] * y[k];
            y[i] = (sig[j] * Q[(size_t)j * m + i] - sdot) / l[i];
        }
        double nn = 0.0;
        for (int i = 0; i < m; ++i) nn += y[i] * y[i];
        nn = 1.0 / sqrt(nn);
        for (int i = 0; i < m; ++i) y[i] *= nn;
    }
    if (kTraceFactorize) fprintf(stderr, "[factorize] warm: probes + X = L'V0 %.3f ms, Jacobi %.3f ms, back substitution %.3f ms\n", t_j - t_x, t_b - t_j, now_ms() - t_b);
    return true;
}

static void gram_eig(lfpsqp_ctx* ctx, int m, const std::vector<double>& G, std::vector<double>& sig, std::vector<double>& V) {
    std::vector<double> Lc, Ug, lam;
    sig.assign(m, 0.0);
    const double t0 = now_ms();
    const bool pd = cholesky_lower(m, G, Lc);
    if (kTraceFactorize) fprintf(stderr, "[factorize] cholesky %.3f ms (pd=%d)\n", now_ms() - t0, (int)pd);
    std::vector<double> V0;
    if (ctx && ctx->warm_m == m) V0.swap(ctx->warm_V);       // a hint is consumed by the first factorisation of its size ...
    if (ctx) { ctx->warm_m = 0; ctx->warm_V.clear(); }       // ... or dropped
    if (pd && !V0.empty() && gram_eig_warm(ctx, m, Lc, V0, sig, V)) {
        if (kTraceFactorize) fprintf(stderr, "[factorize] warm start used\n");
        return;
    }
    if (pd) {
        std::vector<double> none;
        small_svd(ctx, m, m, Lc, V, sig, none, false);
    } else {
        small_svd(ctx, m, m, G, Ug, lam, V);
        for (int j = 0; j < m; ++j) sig[j] = sqrt(lam[j] > 0 ? lam[j] : 0.0);
    }
}

// One refinement round of the rank-revealing factorisation (lfpsqp_factorize, step 3).  In: the orthogonal V (m x m), the
// scalings s (m) the trial basis Z = A V diag(1/s) was formed with, and its measured Gram matrix GZ = Z'Z.  The products
// z_i'z_j are accurate RELATIVE to |z_i||z_j| however ill-conditioned A is, so the unit-diagonal matrix Gs = GZ ./ (dz dz'),
// dz = sqrt(diag GZ), is known to absolute accuracy eps -- and the Gram matrix of B = A V, G' = D Gs D with D = diag(s.*dz) (the
// TRUE column norms of B), is known in factored form.  Its eigen-decomposition to high RELATIVE accuracy (Demmel & Veselic):
// Gs = Ls Ls' (well conditioned once the columns of B are roughly orthogonal), X = Ls' D, one-sided Jacobi on the columns of
// X: X V2 = Ux diag(sig_new), so G' = X'X = V2 diag(sig_new^2) V2'.  Then V <- V V2 and sig <- sig_new.
// Columns whose true norm is at the rounding level of the products (<= noise) are numerically zero: they keep their V column,
// get sig = their measured norm and take no part.
// Out: V, sig updated (sig descending); returns through `offmax` the largest |Gs_ij| (i != j) among active columns, through
// `conv` whether every pair is orthogonal to max(tol, the rounding floor of the product A*w for that pair), through `devmax`
// the largest |dz_j - 1| among active columns (how well s matched the true norms) and through `rotated` whether V changed.
static void refine_round(lfpsqp_ctx* ctx, int m, std::vector<double>& V, std::vector<double>& sig, const std::vector<double>& s, const std::vector<double>& GZ,
                         double tol, double* offmax, bool* conv, double* devmax, bool* rotated) {
    const double eps = 2.220446049250313e-16;
    std::vector<double> dz(m), bt(m);
    double btmax = 0.0;
    for (int j = 0; j < m; ++j) {
        const double g = GZ[(size_t)j * m + j];
        dz[j] = (g > 0.0 && isfinite(g)) ? sqrt(g) : 0.0;
        bt[j] = s[j] * dz[j];
        btmax = std::max(btmax, bt[j]);
    }
    const double noise = 64.0 * eps * sqrt((double)m) * btmax;
    std::vector<int> act;
    for (int j = 0; j < m; ++j)
        if (bt[j] > noise) act.push_back(j);
    const int ma = (int)act.size();
    *offmax = 0.0; *devmax = 0.0; *conv = true; *rotated = false;
    std::vector<double> Gs((size_t)ma * ma);
    for (int b = 0; b < ma; ++b) {
        *devmax = std::max(*devmax, fabs(dz[act[b]] - 1.0));
        for (int a = 0; a < ma; ++a) {
            const double v = GZ[(size_t)act[b] * m + act[a]] / (dz[act[a]] * dz[act[b]]);
            Gs[(size_t)b * ma + a] = (a == b) ? 1.0 : v;
            if (a != b) {
                *offmax = std::max(*offmax, fabs(v));
                const double floor_ab = 8.0 * sqrt((double)m) * eps * btmax / std::min(bt[act[a]], bt[act[b]]);
                if (!(fabs(v) <= std::max(tol, floor_ab))) *conv = false;
            }
        }
    }
    std::vector<double> signew(bt), Vnew(V);
    if (!*conv && ma > 1) {
        std::vector<double> Ls, X((size_t)ma * ma, 0.0), Ux, S2, V2;
        if (cholesky_lower(ma, Gs, Ls, 1e-13)) {
            for (int j = 0; j < ma; ++j)                       // X = Ls' D:  X[i, j] = Ls[j, i] * bt_j
                for (int i = 0; i <= j; ++i) X[(size_t)j * ma + i] = Ls[(size_t)i * ma + j] * bt[act[j]];
        } else {                                               // Gs = Ve diag(le) Ve':  X = diag(sqrt(le)) Ve' D
            std::vector<double> Ue, le, Ve;
            small_svd(ctx, ma, ma, Gs, Ue, le, Ve);
            for (int j = 0; j < ma; ++j)
                for (int i = 0; i < ma; ++i) X[(size_t)j * ma + i] = sqrt(le[i] > 0 ? le[i] : 0.0) * Ve[(size_t)i * ma + j] * bt[act[j]];
        }
        small_svd(ctx, ma, ma, X, Ux, S2, V2);
        for (int b = 0; b < ma; ++b) {                         // V[:, act] <- V[:, act] * V2
            double* dst = &Vnew[(size_t)act[b] * m];
            for (int i = 0; i < m; ++i) dst[i] = 0.0;
            for (int a = 0; a < ma; ++a) {
                const double c = V2[(size_t)b * ma + a];
                const double* src = &V[(size_t)act[a] * m];
                for (int i = 0; i < m; ++i) dst[i] += src[i] * c;
            }
            signew[act[b]] = S2[b];
        }
        *rotated = true;
    }
    // order by singular value, descending (numerically zero columns end up last)
    std::vector<int> idx(m);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return signew[a] > signew[b]; });
    bool moved = false;
    for (int j = 0; j < m; ++j) {
        moved = moved || idx[j] != j;
        sig[j] = signew[idx[j]];
        for (int i = 0; i < m; ++i) V[(size_t)j * m + i] = Vnew[(size_t)idx[j] * m + i];
    }
    if (moved) *rotated = true;
}

// ---------------------------------------------------------------------------
// The factorisation of an n x m matrix A that is only reachable through two operations: gramA(G) = A' diag(w2) A (m x m, host,
// replicated) and rmulA(Wh, r) = "Z[:, :r] = A * Wh" (Wh: m x r, host, column-major).  lfpsqp_factorize supplies the dense MFMA
// kernels, lfpsqp_factorize_sp the products of a sparse A.
using GramFn = std::function<int(std::vector<double>&)>;
using RmulFn = std::function<int(const double*, int)>;
// Z == nullptr: the caller wants the factors only (Sigma, Vt, W with the basis = A W left in factored form, FINDINGS.md 5.3): on the fast path no
// basis-forming product runs at all; the refinement rounds of an ill-conditioned block still need their trial basis -- needZ() provides one.
static int factorize_core(lfpsqp_ctx* ctx, int m, const GramFn& gramA, const RmulFn& rmulA, const double* w2p, lfpsqp_mat*& Z,
                          const std::function<int()>& needZ, double* Sigma, double* Vt, double* W, int64_t* rank_out, double eps_rank) {
    const bool want_basis = Z != nullptr;
    *rank_out = 0;
    if (m == 0) return 0;
    for (size_t i = 0; i < (size_t)m * m; ++i) Vt[i] = 0.0;
    if (W)
        for (size_t i = 0; i < (size_t)m * m; ++i) W[i] = 0.0;
    auto zero_from = [&](int r) -> int {       // columns >= r of Z are zero (the rmul passes overwrite columns < r completely)
        if (Z && r < Z->m) {
            hipLaunchKernelGGL(zero_cols_kernel, dim3(256, (unsigned)(Z->m - r)), dim3(kThreads), 0, ctx->stream, Z->p, Z->ld, Z->n, r);
            LF_LAUNCH_CHECK(ctx);
        }
        return 0;
    };
    auto finish = [&](const std::vector<double>& sig, const std::vector<double>& V, int r, const std::vector<double>* Wr) {
        for (int j = 0; j < m; ++j) Sigma[j] = sig[j];
        for (int k = 0; k < r; ++k)
            for (int j = 0; j < m; ++j) Vt[(size_t)j * m + k] = V[(size_t)k * m + j];     // Vt[k, j] = V[j, k]; rows >= rank stay zero
        if (W && Wr)
            for (size_t i = 0; i < (size_t)m * r; ++i) W[i] = (*Wr)[i];                  // Z[:, :r] = Jct * W[:, :r]; columns >= r stay zero
        *rank_out = r;
    };
    // 1. G = A'A (A = diag(sqrt(w2)) Jct), eigen-decomposition: estimates of the singular values and right singular vectors
    std::vector<double> G, sig, V;
    const double t_start = now_ms();
    LF_TRY(gramA(G));
    for (double g : G)
        if (!isfinite(g)) return set_err(ctx, LFPSQP_ERR_NUMERIC, "factorize: non-finite Gram matrix");
    const double t_eig = now_ms();
    gram_eig(ctx, m, G, sig, V);
    if (kTraceFactorize) fprintf(stderr, "[factorize] gram %.3f ms, eig %.3f ms\n", t_eig - t_start, now_ms() - t_eig);
    if (!(sig[0] > 0.0)) {                     // A == 0
        LF_TRY(zero_from(0));
        finish(sig, V, 0, nullptr);
        return 0;
    }
    // 2. A well-conditioned full-rank factor needs nothing more: the loss of orthogonality of A V S^-1 and the relative error of the
    //    small singular values are eps * cond(A)^2, i.e. rounding level for cond(A)^2 <= 10 (the dense random equality blocks of
    //    the BASELINE configs have cond ~ 1.1).  Then A = (A V S^-1) S V' is already the factorisation.
    if (sig[m - 1] >= eps_rank && sig[0] * sig[0] <= 10.0 * sig[m - 1] * sig[m - 1]) {
        std::vector<double> W1((size_t)m * m);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) W1[(size_t)j * m + i] = V[(size_t)j * m + i] / sig[j];
        if (want_basis) LF_TRY(rmulA(W1.data(), m));
        LF_TRY(zero_from(m));
        finish(sig, V, m, &W1);
        return 0;
    }
    // 3. Otherwise the eigenvalues of G below ~eps * sig_1^2 are noise (a Gram matrix squares the condition number), while the
    //    reference's dgesvd is backward stable and its rank test is ABSOLUTE (sig_j >= eps_rank = 1e-10, src/optimize.jl:297-302).
    //    Refinement rounds resolve what dgesvd resolves: form the trial basis Z = A V diag(1/s) with the current estimates,
    //    measure its Gram matrix on the device, and correct V and the singular values from it on the host (refine_round: a
    //    one-sided Jacobi step on a well-scaled m x m factor -- every round gains ~8 digits of relative range; converged when
    //    the trial basis is orthogonal to the rounding floor of the products).  Cost per round: one rmul + one Gram pass.
    if (!Z) LF_TRY(needZ());
    const double tol = 4e-13;
    constexpr int kMaxRounds = 6;
    std::vector<double> s(m), Wk((size_t)m * m), GZ;
    bool z_is_final = false;
    for (int round = 1; round <= kMaxRounds; ++round) {
        for (int j = 0; j < m; ++j) s[j] = std::max(sig[j], 1e-12 * sig[0]);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) Wk[(size_t)j * m + i] = V[(size_t)j * m + i] / s[j];
        LF_TRY(rmulA(Wk.data(), m));
        LF_TRY(gram_impl(ctx, Z, m, w2p, GZ));
        for (double g : GZ)
            if (!isfinite(g)) return set_err(ctx, LFPSQP_ERR_NUMERIC, "factorize: non-finite Gram matrix of the trial basis");
        double offmax = 0.0, devmax = 0.0;
        bool conv = false, rotated = false;
        refine_round(ctx, m, V, sig, s, GZ, tol, &offmax, &conv, &devmax, &rotated);
        if (conv && !rotated && devmax <= tol) {          // the basis on the device IS A V diag(1/sig) for every kept column
            z_is_final = true;
            break;
        }
        // quadratic convergence: a correction computed from an almost orthogonal trial basis (off-diagonals E) leaves E^2
        if (conv || offmax <= 5e-7) break;
    }
    int r = 0;
    while (r < m && sig[r] >= eps_rank && sig[r] > 0) ++r;                               // the reference's rule, src/optimize.jl:297-302
    std::vector<double> Wr((size_t)m * std::max(r, 1));
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < m; ++i) Wr[(size_t)j * m + i] = V[(size_t)j * m + i] / sig[j];
    if (r > 0 && !z_is_final && want_basis) LF_TRY(rmulA(Wr.data(), r));
    LF_TRY(zero_from(r));
    finish(sig, V, r, &Wr);
    return 0;
}

}  // namespace lfpsqp

using namespace lfpsqp;

// The weighted Gram kernel scales BOTH operands by sqrt(w2) (one staged operand for a diagonal block): weights must be >= 0 -- they are squares
// in every use of the hot path (Dy^2, phi'^2).  A negative weight turns its row into NaN: reported as an argument error, not returned as data.
static int check_weights(lfpsqp_ctx* ctx, bool weighted, const std::vector<double>& G) {
    if (!weighted) return 0;
    for (double g : G)
        if (g != g) return lfpsqp::set_err(ctx, LFPSQP_ERR_ARG, "lfpsqp_gram: non-finite weighted Gram matrix -- the weights w2 must be >= 0 (the kernel applies sqrt(w2) to both operands)");
    return 0;
}

// G (ncols x ncols, column-major) = R' diag(w) R for R_i = M_i + sgn_i M_{i+1} (w >= 0 and sgn = +-1: device n-vectors; projcg.hip)
int lfpsqp::gram_shifted(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, const double* w, const double* sgn, std::vector<double>& G) {
    LF_TRY(gram_impl(ctx, M, ncols, w, G, nullptr, sgn));
    return check_weights(ctx, true, G);
}

extern "C" {

int lfpsqp_gram(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* w2, double* G_host) {
    LF_RANGE("lfpsqp_gram");
    LF_ARG(ctx, ctx && M && G_host && ncols >= 0 && ncols <= M->m && (!w2 || w2->n == M->n));
    std::vector<double> G;
    LF_TRY(gram_impl(ctx, M, (int)ncols, w2 ? w2->p : nullptr, G));
    LF_TRY(check_weights(ctx, w2 != nullptr, G));
    for (size_t i = 0; i < G.size(); ++i) G_host[i] = G[i];
    return 0;
}

int lfpsqp_rmul(lfpsqp_ctx* ctx, const lfpsqp_mat* In, int64_t kcols, const double* W_host, int64_t rcols, lfpsqp_mat* Out) {
    LF_RANGE("lfpsqp_rmul");
    LF_ARG(ctx, ctx && In && plain_mat(Out) && W_host && In->p != Out->p && kcols >= 0 && kcols <= In->m && rcols >= 0 && rcols <= Out->m &&
                    In->n == Out->n);
    return rmul_impl(ctx, In, (int)kcols, W_host, (int)rcols, Out);
}

int lfpsqp_small_svd(lfpsqp_ctx* ctx, int64_t rows, int64_t cols, const double* A, double* U, double* S, double* V) {
    LF_ARG(ctx, ctx && A && U && S && rows >= 1 && cols >= 0 && rows <= (1 << 20) && cols <= (1 << 14));
    if (cols == 0) return 0;
    std::vector<double> a(A, A + (size_t)rows * cols), u, sv, v;
    small_svd(ctx, (int)rows, (int)cols, a, u, sv, v, V != nullptr);
    for (size_t i = 0; i < u.size(); ++i) U[i] = u[i];
    for (size_t i = 0; i < sv.size(); ++i) S[i] = sv[i];
    for (size_t i = 0; V && i < v.size(); ++i) V[i] = v[i];
    return 0;
}

int lfpsqp_factorize_hint(lfpsqp_ctx* ctx, const double* Vt_prev, int64_t m) {
    LF_ARG(ctx, ctx && m >= 0 && (m == 0 || Vt_prev));
    ctx->warm_m = 0;
    ctx->warm_V.clear();
    if (m < 2 || m > 256) return 0;                   // (no warm start outside the single-launch regime of the device Jacobi)
    ctx->warm_V.resize((size_t)m * m);
    for (int64_t j = 0; j < m; ++j)                   // Vt[k, i] = V[i, k]: row k of Vt is eigenvector k
        for (int64_t i = 0; i < m; ++i) ctx->warm_V[(size_t)j * m + i] = Vt_prev[(size_t)i * m + j];
    ctx->warm_m = (int)m;
    return 0;
}

int lfpsqp_gram_rhs(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int64_t ncols, const lfpsqp_vec* w2, int64_t nx, const lfpsqp_vec* const* e, double* G_host,
                    double* X_host) {
    LF_RANGE("lfpsqp_gram_rhs");
    LF_ARG(ctx, ctx && M && G_host && ncols >= 0 && ncols <= M->m && (!w2 || w2->n == M->n) && nx >= 0 && nx <= 2 && (nx == 0 || (e && X_host)));
    GramRhs rhs;
    std::vector<double> G, X;
    rhs.X = &X;
    for (int k = 0; k < (int)nx; ++k) {
        LF_ARG(ctx, e[k] && e[k]->n >= M->n);
        rhs.e[rhs.nx++] = e[k]->p;
    }
    LF_TRY(gram_impl(ctx, M, (int)ncols, w2 ? w2->p : nullptr, G, nx > 0 ? &rhs : nullptr));
    LF_TRY(check_weights(ctx, w2 != nullptr, G));
    for (size_t i = 0; i < G.size(); ++i) G_host[i] = G[i];
    for (size_t i = 0; i < X.size(); ++i) X_host[i] = X[i];
    return 0;
}

int lfpsqp_factorize_rhs(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma, double* Vt,
                         double* W, int64_t* rank_out, double eps_rank, const lfpsqp_vec* e, double* Jte_host, double* G_host) {
    LF_RANGE("lfpsqp_factorize");
    LF_ARG(ctx, ctx && Jct && Sigma && Vt && rank_out && (Z ? (plain_mat(Z) && Jct->p != Z->p && Jct->n == Z->n && Z->m >= Jct->m) : W != nullptr) &&
                    (!w2 || w2->n == Jct->n) && (!e || (Jte_host && e->n >= Jct->n)));
    const int m = (int)Jct->m;
    const double* w2p = w2 ? w2->p : nullptr;
    lfpsqp_mat* Zu = Z;
    lfpsqp_mat* Ztmp = nullptr;
    GramRhs rhs;
    std::vector<double> X;
    if (e) { rhs.nx = 1; rhs.e[0] = e->p; rhs.X = &X; }
    const int rc = factorize_core(
        ctx, m,
        [&](std::vector<double>& G) {
            LF_TRY(gram_impl(ctx, Jct, m, w2p, G, e ? &rhs : nullptr));
            if (G_host)
                for (size_t i = 0; i < G.size(); ++i) G_host[i] = G[i];          // the Gram matrix the factors come from (U'U = W'GW for the caller)
            return 0;
        },
        [&](const double* Wh, int r) { return rmul_impl(ctx, Jct, m, Wh, r, Zu); }, w2p, Zu,
        [&]() -> int { LF_TRY(lfpsqp_mat_alloc(ctx, Jct->n, m, &Ztmp)); Zu = Ztmp; return 0; }, Sigma, Vt, W, rank_out, eps_rank);
    if (Ztmp) lfpsqp_mat_free(ctx, Ztmp);
    if (rc == 0 && e)
        for (int i = 0; i < m; ++i) Jte_host[i] = i < (int)X.size() ? X[i] : 0.0;
    return rc;
}

int lfpsqp_factorize(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma, double* Vt,
                     double* W, int64_t* rank_out, double eps_rank) {
    return lfpsqp_factorize_rhs(ctx, Jct, w2, Z, Sigma, Vt, W, rank_out, eps_rank, nullptr, nullptr, nullptr);
}

int lfpsqp_factorize_sp(lfpsqp_ctx* ctx, const lfpsqp_spmat* S, const lfpsqp_mat* Jct, const lfpsqp_vec* w2, lfpsqp_mat* Z, double* Sigma,
                        double* Vt, double* W, int64_t* rank_out, double eps_rank) {
    LF_RANGE("lfpsqp_factorize_sp");
    LF_ARG(ctx, (!Jct || plain_mat(Jct)) && (!Z || plain_mat(Z)));      // (a sparse class scales its nonzeros: lfpsqp_spmat_rowscale)
    LF_ARG(ctx, ctx && S && Sigma && Vt && rank_out && (!w2 || w2->n == S->n) && (Z ? S->n == Z->n : W != nullptr) &&
                    (!Jct || ((!Z || Jct->p != Z->p) && Jct->n == S->n && Jct->m >= S->m && Jct->m - S->m <= 4)) &&
                    (!Z || Z->m >= (Jct ? Jct->m : S->m)));
    const int ms = (int)S->m, m = Jct ? (int)Jct->m : ms;
    const double* w2p = w2 ? w2->p : nullptr;
    // Z == NULL: factors only, the basis stays in factored form U = [S | X] W (as lfpsqp_factorize): no basis-forming product on the fast
    // path; a temporary trial basis for the refinement rounds (or for a dense Gram matrix of S alone) is allocated on demand and freed.
    lfpsqp_mat* Zu = Z;
    lfpsqp_mat* Ztmp = nullptr;
    lfpsqp_vec* stmp = nullptr;
    auto needZ = [&]() -> int {
        if (Zu) return 0;
        LF_TRY(lfpsqp_mat_alloc(ctx, S->n, m, &Ztmp));
        Zu = Ztmp;
        return 0;
    };
    // Gram matrix from the nonzeros (sp_gram: exact fixed-point accumulation, so reproducible; the extra dense columns through SpMV-T and
    // dot products; Z's first column is its scratch vector -- the basis-forming product overwrites Z afterwards).  Where that is refused
    // (rows wider than 16 nonzeros -- 8 when the column sets are scattered --, extreme values): on the dense twin, or on S expanded into Z.
    auto gramA = [&](std::vector<double>& G) -> int {
        G.assign((size_t)m * m, 0.0);
        double* scratch = Zu ? Zu->p : nullptr;
        if (!scratch && m > ms && w2) {                         // (the weighted extra columns need an n-vector of scratch)
            LF_TRY(lfpsqp_vec_alloc(ctx, S->n, &stmp));
            scratch = stmp->p;
        }
        // rows of 9 .. 16 nonzeros: from the nonzeros when consecutive rows mostly share their column sets (banded / block-structured systems:
        // 1.1 / 2.4 ms at K = 12 / 16 against 2.8 ms for the dense Gram at n = 1e7, m = 128, and 1.6 / 4.0 against 25 at m = 512); with scattered
        // column sets the lane groups flush their sums at every row (4.2 / 9.1 ms) and the dense twin is the faster one
        const int kmax = S->run_frac >= 0.5 ? 16 : 8;
        const int rc = ctx->tune_spgram >= 0 ? sp_gram(ctx, S, Jct, ms, m - ms, w2, scratch, G.data(), kmax) : LFPSQP_ERR_UNSUPPORTED;
        if (rc != LFPSQP_ERR_UNSUPPORTED) return rc;
        if (Jct) return gram_impl(ctx, Jct, m, w2p, G);
        LF_TRY(needZ());
        LF_TRY(lfpsqp_spmat_to_dense(ctx, S, Zu));
        return gram_impl(ctx, Zu, m, w2p, G);
    };
    auto rmulA = [&](const double* Wh, int r) -> int {
        if (r == 0) return 0;
        LF_TRY(ensure_small(ctx, (size_t)m * r + 16));
        LF_HIP(ctx, hipMemcpyAsync(ctx->small, Wh, sizeof(double) * (size_t)m * r, hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));                      // Wh is pageable host memory
        return spmm(ctx, S, Jct, ms, m - ms, ctx->small, m, r, Zu);
    };
    lfpsqp_mat* Zcore = Z;           // what factorize_core sees: the caller's basis, or none (it asks needZ for a trial basis when it must)
    const int rc = factorize_core(ctx, m, gramA, rmulA, w2p, Zcore, [&]() -> int { LF_TRY(needZ()); Zcore = Zu; return 0; }, Sigma, Vt, W, rank_out,
                                  eps_rank);
    if (Ztmp) lfpsqp_mat_free(ctx, Ztmp);
    if (stmp) lfpsqp_vec_free(ctx, stmp);
    return rc;
}

}  // extern "C"
