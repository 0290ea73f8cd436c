// One-sided (Hestenes) Jacobi on the device for the replicated m x m problems of the tangent setup (factorize.hip): the
// eigen-decomposition of a Gram matrix through its Cholesky factor and the SVD of the small factor of a refinement round.
// The host version (smallla.h) costs 3.5 ms at m = 128 and 105 ms at m = 512 per problem -- a fifth to a half of
// lfpsqp_factorize; the vendor's solver would mean loading a 0.9 GB library.
//
// Two-level block Jacobi.  The columns are cut into blocks of B; a workgroup takes a PAIR of blocks (2B columns, all rows)
// into LDS, runs one complete cyclic sweep over those 2B columns there (2B-1 local rounds of B disjoint column pairs, a
// group of 256/B lanes per pair: dot products over the lanes' rows, one rotation, a barrier) and writes the columns
// back.  One launch = one round of the circle-method tournament over the blocks (nb/2 disjoint block pairs = nb/2
// workgroups, no communication between them).  A sweep = an opening launch in which neighbouring blocks run a full cyclic sweep
// over their union (all pairs inside a block meet there) + the nb-1 tournament rounds restricted to the CROSS pairs of a block
// pair (B local rounds instead of 2B-1): every pair of columns meets at least once per sweep.  Launch boundaries are the only grid-wide synchronisation, so the same code runs on the CPU emulator.
//   Rotations act on rows [0, rows_all), the dot products that define them use rows [0, rows_dot): with X stacked on an
// identity, [X; I], the lower half accumulates the right singular vectors for free.
//   Convergence: every rotated pair reports |p.q| / (|p||q|); the largest value of a sweep lands in off[sweep] (an
// atomic max on the bit pattern of a non-negative double: order-independent, so bit-reproducible and identical on every
// rank).  A launch whose predecessors include a converged sweep is a no-op; the host looks at off[] one sweep behind, and stops
// two sweeps after the level enters the quadratic regime (kJacQuadTol) rather than after a sweep that measures rounding level.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "internal.h"

namespace lfpsqp {

constexpr unsigned long long kJacUnset = ~0ull;      // off[] slot not written yet

// circle method: who sits at position `pos` (0 .. np-1, np even) in round `round`
__device__ __forceinline__ int circle_player(int pos, int round, int np) {
    if (pos == 0) return 0;
    int v = (pos - 1 - round) % (np - 1);
    if (v < 0) v += np - 1;
    return 1 + v;
}

// Sum over the LPP (4, 8, 16 or 32) consecutive lanes of a column pair; every lane gets the total.  A butterfly over the
// lane-index masks 1, 2, 7, 15 (which generate all of 0..15): quad permutations and the half-row / row mirrors of the DPP unit
// (two v_mov_b32 dpp per step -- a shuffle through the LDS crossbar costs several times that).  The emulator's xor shuffles
// pair the same operands, so the bits agree.
template <int MASK>
__device__ __forceinline__ double lane_xor(double v) {
#ifdef LFPSQP_HIP_EMULATED
    return __shfl_xor(v, MASK);
#else
    constexpr int ctrl = MASK == 1 ? 0xB1 : (MASK == 2 ? 0x4E : (MASK == 7 ? 0x141 : 0x140));   // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
    static_assert(MASK == 1 || MASK == 2 || MASK == 7 || MASK == 15, "DPP pattern");
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, ctrl, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), ctrl, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
#endif
}
template <int LPP>
__device__ __forceinline__ double pair_sum(double v) {
    if (LPP >= 32) v += __shfl_xor(v, 16);
    if (LPP >= 16) v += lane_xor<15>(v);
    if (LPP >= 8) v += lane_xor<7>(v);
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    return v;
}
// 1/sqrt(x) and 1/x for finite positive x / non-zero x: the hardware estimate and two Newton steps (the compiler's IEEE
// sequences for double sqrt and division are ~30 instructions each, and every lane of a pair runs them redundantly)
__device__ __forceinline__ double fast_rsqrt(double x) {
#ifdef LFPSQP_HIP_EMULATED
    return 1.0 / sqrt(x);
#else
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
#endif
}
__device__ __forceinline__ double fast_rcp(double x) {
#ifdef LFPSQP_HIP_EMULATED
    return 1.0 / x;
#else
    double y = __builtin_amdgcn_rcp(x);
    y = y * fma(-x, y, 2.0);
    y = y * fma(-x, y, 2.0);
    return y;
#endif
}

constexpr double kJacQuadTol = 1e-7;    // relative off-diagonal level from which two more sweeps reach rounding (quadratic convergence)
constexpr double kJacQuadTol1 = 1e-8;   // ... and from which ONE does: the sweep that measured it leaves (1e-8)^2 = 1e-16 behind.  (At m = 128 the
                                        // clustered Gram spectra of the configs enter at 2e-9: the ninth sweep only re-measured rounding, 0.28 ms.)

// B columns per block, 2B per workgroup; LPP lanes per column pair (B * LPP threads), each holding MAXROWS / LPP rows of both
// columns in registers for the dot products and the rotation of a local round (one batch of LDS reads, one of writes per round).
template <int B, int MAXROWS, int LPP>
__global__ __launch_bounds__(B* LPP) void jacobi_round_kernel(double* __restrict__ X, int ld, int rows_dot, int rows_all, int nb, int round,
                                                               int sweep, unsigned long long* off, double tol) {
    // done already?  A sweep that found nothing left to rotate (off <= tol) ends the iteration; so does, one sweep later, a sweep
    // that STARTED from off <= kJacQuadTol: its own rotations square that (cyclic Jacobi converges quadratically, also for
    // clustered values), and the sweep after it -- which the host has queued by the time it sees the number -- squares it again.
    for (int k = 0; k < sweep; ++k) {
        const unsigned long long u = (unsigned long long)ld_stat(reinterpret_cast<const int64_t*>(off) + k);
        if (u == kJacUnset) continue;
        const double o = __builtin_bit_cast(double, u);
        if (o <= tol || (k + 1 <= sweep && o <= kJacQuadTol1) || (k + 2 <= sweep && o <= kJacQuadTol)) return;
    }
    constexpr int THREADS = B * LPP;
    constexpr int RPL = MAXROWS / LPP;                 // rows per lane
    constexpr int LD = MAXROWS + 1;                    // odd column stride in LDS
    static_assert(THREADS % 64 == 0 && MAXROWS % LPP == 0, "lane groups");
    __shared__ double cols[2 * B][LD];
    __shared__ double offw[THREADS / 64];
    const int tid = threadIdx.x;
    // round < 0: the sweep's opening launch -- neighbouring blocks (2k, 2k+1), a FULL cyclic sweep over their 2B columns (this is where
    // the pairs inside a block meet); round >= 0: round of the block tournament, CROSS pairs only (column i of one block with every
    // column of the other: B local rounds instead of 2B-1)
    const bool full = round < 0;
    const int bp = full ? 2 * (int)blockIdx.x : circle_player(blockIdx.x, round, nb);
    const int bq = full ? 2 * (int)blockIdx.x + 1 : circle_player(nb - 1 - blockIdx.x, round, nb);
    // load the two blocks: a wave takes whole columns, 16 bytes per lane, all loads of a column in flight together
    for (int c = tid >> 6; c < 2 * B; c += THREADS / 64) {
        const double* src = X + (size_t)((c < B ? bp * B + c : bq * B + (c - B))) * ld;
        constexpr int NCH = (MAXROWS + 127) / 128;
        double2 v[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int r = 2 * (tid & 63) + 128 * k;
            v[k] = (r + 1 < rows_all) ? ld2(src + r) : make_double2(r < rows_all ? src[r] : 0.0, 0.0);     // (ld is even: columns are 16-byte aligned)
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int r = 2 * (tid & 63) + 128 * k;
            if (r < MAXROWS) cols[c][r] = v[k].x;
            if (r + 1 < MAXROWS) cols[c][r + 1] = v[k].y;
        }
    }
    __syncthreads();
    const int g = tid / LPP, l = tid % LPP;
    double offmax = 0.0;
    const int nlr = full ? 2 * B - 1 : B;
#pragma unroll 1
    for (int lr = 0; lr < nlr; ++lr) {
        int i = full ? circle_player(g, lr, 2 * B) : g;
        int j = full ? circle_player(2 * B - 1 - g, lr, 2 * B) : B + ((g + lr) % B);
        if (i > j) { const int t = i; i = j; j = t; }
        double* p = cols[i] + l;
        double* q = cols[j] + l;
        double x[RPL], y[RPL];
#pragma unroll
        for (int k = 0; k < RPL; ++k) { x[k] = p[k * LPP]; y[k] = q[k * LPP]; }
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
        for (int k = 0; k < RPL; ++k)
            if (l + k * LPP < rows_dot) {
                alpha = fma(x[k], x[k], alpha);
                beta = fma(y[k], y[k], beta);
                gamma = fma(x[k], y[k], gamma);
            }
        alpha = pair_sum<LPP>(alpha);
        beta = pair_sum<LPP>(beta);
        gamma = pair_sum<LPP>(gamma);
        const double ab = alpha * beta;
        const double rlim = (ab > 0.0 && ab < 1e300) ? fast_rsqrt(ab) : 0.0;     // 1 / (|p||q|); a zero column rotates with nothing
        const double rel = fabs(gamma) * rlim;                                   // |cos| of the angle between the columns
        if (rel > 1e-16) {
            offmax = fmax(offmax, rel);
            // the smaller-angle rotation that makes the pair orthogonal: tan(2 theta) = 2 gamma / (beta - alpha).  With d = beta - alpha and
            // r = sqrt(d^2 + 4 gamma^2): cos^2 = (1 + |d| / r) / 2 and sin = sign(d) gamma / (r cos) -- TWO reciprocal square roots instead of
            // the classical zeta / t sequence's two reciprocals and two reciprocal square roots (every lane of a pair runs them redundantly:
            // a tenth of the round's instructions).
            const double d = beta - alpha;
            const double dn = d * rlim, gn = gamma * rlim;     // scaled by 1 / (|p||q|), which the convergence measure needs anyway: |gn| <= 1
            double c, s;
            if (fabs(dn) < 1e100) {
                const double rinv = fast_rsqrt(fma(dn, dn, 4.0 * gn * gn));
                const double h = fma(0.5 * fabs(dn), rinv, 0.5);   // cos^2 in [1/2, 1]
                const double cinv = fast_rsqrt(h);
                c = h * cinv;
                s = (d >= 0 ? gn : -gn) * rinv * cinv;
            } else {                                           // columns of wildly different size: the rotation is the identity to rounding
                c = 1.0;
                s = gn * fast_rcp(dn);
            }
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                p[k * LPP] = c * x[k] - s * y[k];
                q[k * LPP] = s * x[k] + c * y[k];
            }
        }
        __syncthreads();
    }
    for (int c = tid >> 6; c < 2 * B; c += THREADS / 64) {
        double* dst = X + (size_t)((c < B ? bp * B + c : bq * B + (c - B))) * ld;
        for (int r = tid & 63; r < rows_all; r += 64) dst[r] = cols[c][r];
    }
    offmax = wave_max(offmax);
    if ((tid & 63) == 0) offw[tid >> 6] = offmax;
    __syncthreads();
    if (tid == 0) {
        double o = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) o = fmax(o, offw[w]);
        // off[sweep] = max(off[sweep], o): non-negative doubles order like their bit patterns; an unset slot becomes o
        unsigned long long* slot = off + sweep;
        unsigned long long cur = atomicCAS(slot, kJacUnset, __builtin_bit_cast(unsigned long long, o));
        if (cur != kJacUnset) atomicMax(slot, __builtin_bit_cast(unsigned long long, o));
    }
}

// Columns of the host matrix X (rows_all x cols, column-major, tight) are rotated until mutually orthogonal over the
// leading rows_dot rows.  Returns false (X untouched) when the shape is outside what the kernels cover: the caller then
// uses the host routine.  `sweeps_out` (optional) receives the number of sweeps run.
bool device_jacobi(lfpsqp_ctx* ctx, int rows_dot, int rows_all, int cols, std::vector<double>& X, int* sweeps_out) {
    if (cols < 2 || rows_all > 1024 || rows_dot > rows_all) return false;
    // block size by the rows a workgroup must hold: 2B columns x rows_all in <= 132 KB of LDS
    const int B = rows_all <= 128 ? 64 : (rows_all <= 256 ? 32 : (rows_all <= 512 ? 16 : 8));
    int nb = (cols + B - 1) / B;
    if (nb < 2) nb = 2;
    nb += nb & 1;
    const int colsp = nb * B;
    constexpr int kMaxSweeps = 40;
    const int ldx = rows_all + (rows_all & 1);          // even leading dimension: 16-byte aligned columns on the device
    const size_t need = (size_t)colsp * ldx + kMaxSweeps + 8;
    if (ensure_small(ctx, need) != 0) return false;
    double* dX = ctx->small;
    unsigned long long* doff = reinterpret_cast<unsigned long long*>(ctx->small + (size_t)colsp * ldx);
    bool ok = hipMemsetAsync(dX, 0, sizeof(double) * (size_t)colsp * ldx, ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(doff, 0xff, sizeof(unsigned long long) * kMaxSweeps, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpy2DAsync(dX, sizeof(double) * ldx, X.data(), sizeof(double) * rows_all, sizeof(double) * rows_all, (size_t)cols,
                                hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;          // X is pageable caller memory
    if (!ok) return false;
    const double tol = 1e-15;
    volatile unsigned long long* hoff = reinterpret_cast<volatile unsigned long long*>(ctx->h_scal + 64);   // pinned: slots 64..191 of h_scal
    for (int k = 0; k < kMaxSweeps; ++k) hoff[k] = kJacUnset;
    int sweep = 0;
    bool done = false;
    for (; sweep < kMaxSweeps && !done; ++sweep) {
        for (int round = -1; round < (nb > 2 ? nb - 1 : 0); ++round) {
#define LF_JR(BB, MR, LP) hipLaunchKernelGGL((jacobi_round_kernel<BB, MR, LP>), dim3(nb / 2), dim3(BB * LP), 0, ctx->stream, dX, ldx, rows_dot, rows_all, nb, round, sweep, doff, tol)
            // lanes per pair as measured on MI355X (ms per problem at m = 128 / 256 / 512): 4 lanes 5.3 / - / -, 8 lanes 2.9 / 12.9 / -,
            // 16 lanes 2.6 / 9.0 / 32.2, 32 lanes - / - / 26.5
            if (B == 64) LF_JR(64, 128, 16);
            else if (B == 32) LF_JR(32, 256, 16);
            else if (B == 16) LF_JR(16, 512, 32);
            else LF_JR(8, 1024, 32);
#undef LF_JR
        }
        if (hipGetLastError() != hipSuccess) return false;
        ok = hipMemcpyAsync(const_cast<unsigned long long*>(hoff) + sweep, doff + sweep, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
        ok = ok && hipEventRecord(ctx->ev_slot[sweep & 3], ctx->stream) == hipSuccess;
        if (!ok) return false;
        if (sweep >= 1) {                                // one sweep behind: the next one is queued while this result travels
            if (hipEventSynchronize(ctx->ev_slot[(sweep - 1) & 3]) != hipSuccess) return false;
            const unsigned long long u = hoff[sweep - 1];
            // (unset: that sweep's launches were no-ops already; <= kJacQuadTol: sweep - 1 and the queued sweep finish the job)
            if (u == kJacUnset || __builtin_bit_cast(double, u) <= kJacQuadTol) done = true;
        }
    }
    if (hipMemcpy2DAsync(X.data(), sizeof(double) * rows_all, dX, sizeof(double) * ldx, sizeof(double) * rows_all, (size_t)cols, hipMemcpyDeviceToHost,
                         ctx->stream) != hipSuccess)
        return false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    if (sweeps_out) *sweeps_out = sweep;
    if (getenv("LFPSQP_TRACE_FACTORIZE")) {
        fprintf(stderr, "[jacobi] rows %d/%d cols %d B %d: %d sweeps queued, off per sweep:", rows_dot, rows_all, cols, B, sweep);
        for (int k = 0; k < sweep; ++k) fprintf(stderr, " %.1e", hoff[k] == kJacUnset ? -1.0 : __builtin_bit_cast(double, (unsigned long long)hoff[k]));
        fprintf(stderr, "\n");
    }
    return true;
}

}  // namespace lfpsqp
