// The tangent step of an outer iteration in ONE pass over the constraint gradients.
//
// Between the factorisation and the first projected-CG iteration the reference's outer loop (src/optimize.jl:305-343, 366-381) and the
// head of projcg! (src/projcg.jl:55-62) make, for a basis U = Jct W kept in factored form, FIVE passes over the matrix:
//     tmp_m = U'd                     GEMV-T   :306          (1)
//     d    -= U tmp_m                 GEMV-N   :307          (2)
//     hess_lag_vec!'s constraint term GEMV-N   (autodiff_generators.jl:72-107; the nonlinear class: phi''(x) .* (A lam))   (3)
//     Utr = U'r0,  r0 = A x0 - b = -d GEMV-T   projcg.jl:56-59   (4)
//     g = r0 - U Utr ...              the INIT form of the fused iteration (projcg.hip)                                     (5)
// (1) rides with the Gram pass (lfpsqp_factorize_rhs: d is known before jac! runs).  (2), (3), (4) are ONE pass here: both first products
// (coefficients W tmp_m and lam) meet the row in registers, d is projected, the Hessian diagonal is completed, and r0 = -d goes straight back
// into the tile for the second product.  lambda_kkt = V S^-1 tmp_m (:331-343) is replicated m x m work on the host in between.
#include <math.h>

#include "internal.h"
#include "sparse.h"

#ifndef LFPSQP_TANGENT_STAGE_STACKED
#define LFPSQP_TANGENT_STAGE_STACKED 0
#endif
#ifndef LFPSQP_TANGENT_STAGE
#define LFPSQP_TANGENT_STAGE 1      // (0: the tangent step stores row by row -- A/B builds, tools/build_variant.py)
#endif

namespace lfpsqp {

// MODE 0: projection only; 1: + the constant part of the class's Hessian term on the rows < n_x (ball / common quadratic term);
// 2: + phi''(x_i) (A lam)_i from the second first product (NA = 2).
// The functor takes a VIEW itself (kRowScaled: run_onepass launches it over the plain storage as it is): the projection's product is the
// view's, diag(rs) A u1 + u (w'u1), the Hessian term's is the plain A's -- one matrix stream serves both.
// INITF: the pass is ALSO projcg!'s initial projection (src/projcg.jl:58-62): a further first product brings (U Utr0)_i, Utr0 = U'r0 =
// -(I - U'U) U'd from the host (U'U = W'GW with the Gram matrix of the factorisation: no pass of its own), and the row goes on
//     g0 = r0 - (U Utr0)_i;  d_cg = -g0;  second products of g0 and A g0;  sums r0'g0, g0'g0, g0'A g0
// -- what lfpsqp_projcg's initial one-pass launch would do with a residual vector in between.  gout / dcg: projcg's work vectors g and d.
template <int MODE, bool INITF = false>
struct TangentStepE {
    static constexpr bool kRowScaled = true;
    static constexpr bool kSplitRed = false;
    static constexpr int kNV = INITF ? 2 : 1;
    static constexpr int kNRED = INITF ? 6 : 2;         // |d|^2, u'r0   /   |d|^2, r0'g0, g0'g0, g0'A g0, u'g0, u'(A g0)
    double* d;          // in: the step; out: its projection
    double* rp;         // out: r0 = -d (projcg!'s stored residual, src/projcg.jl:56-57 with x0 = 0); INITF: g0 goes to `gout`, -g0 to `dcg`, rp is not written
    double* gout = nullptr;
    double* dcg = nullptr;
    double a0 = 0.0;    // INITF with MODE == 0: the operator diagonal is a0 + hx (read only)
    double* hx;         // MODE > 0: the Hessian diagonal, completed in place
    const double* x;
    const double* kind;
    double cq;
    int64_t n_x;
    ViewD vw;
    struct Uni { double tau, taub; };
    struct Row { double d, s, u, hx, x, kk; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ Uni uniform() const {
        return Uni{vw.u ? uniform_f64(ld_scal(vw.tau)) : 0.0, (INITF && vw.u) ? uniform_f64(ld_scal(vw.tau + 1)) : 0.0};
    }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.d = at(d, o);
        w.s = vw.rs ? at(vw.rs, o) : 1.0;
        w.u = vw.u ? at(vw.u, o) : 0.0;
        w.hx = (MODE > 0 || INITF) ? (hx ? at(hx, o) : 0.0) : 0.0;
        w.x = MODE > 1 ? at(x, o) : 0.0;
        w.kk = (MODE > 1 && kind) ? at(kind, o) : 0.0;
        return w;
    }
    // staged form (onepass_kernel STG; INITF only): the projected step, g0, -g0 and (MODE > 0) the completed Hessian diagonal wait in LDS and
    // leave in bursts -- four thin store streams inside the matrix read stream are what kept this pass at 0.6 of the HBM peak
    static constexpr bool kStageAnyNA = true;
    static constexpr int kStageStreams = (INITF && LFPSQP_TANGENT_STAGE) ? (MODE > 0 ? 4 : 3) : 0;
    __device__ __forceinline__ double* stage_out(int sv) const { return sv == 0 ? d : (sv == 1 ? gout : (sv == 2 ? dcg : hx)); }
    template <int NA>
    __device__ __forceinline__ void apply_staged(int64_t row, uint32_t o, const double (&acc)[NA], bool valid, bool owner, bool lead, const Uni& u, const Row& w,
                                                 double (&v)[kNV], double (&red)[kNRED], double* slot, int sstride) const {
        apply(row, o, acc, valid, owner, lead, u, w, v, red, slot, sstride);
    }
    template <int NA>
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&acc)[NA], bool valid, bool owner, bool, const Uni& u, const Row& w,
                                          double (&v)[kNV], double (&red)[kNRED], double* slot = nullptr, int sstride = 0) const {
        const double s = valid ? w.s : 0.0;
        const double proj = valid ? fma(w.u, u.tau, acc[0] * s) : 0.0;      // (U tmp_m)_i
        const double dp = w.d - proj;                                       // :307
        const double r0 = valid ? -dp : 0.0;
        double h = w.hx;
        if (MODE > 1) h += ew_phi2(w.kk, w.x) * acc[NA - 1];
        if (MODE > 0) h += (row < n_x) ? cq : 0.0;
        if (valid && owner) {
            if (slot) {
                slot[0] = dp;
                if (MODE > 0) slot[3 * sstride] = h;
            } else {
                put(d, o, dp);
                if (!INITF) put(rp, o, r0);
                if (MODE > 0) put(hx, o, h);
            }
            red[0] = fma(dp, dp, red[0]);                                   // |d|^2 (the truncated-Newton tolerance, :373-375)
        }
        if constexpr (!INITF) {
            if (valid && owner) red[1] = fma(w.u, r0, red[1]);              // u'r0: the view's rank-one term of the second product
            v[0] = r0 * s;
        } else {
            const double yb = valid ? fma(w.u, u.taub, acc[1] * s) : 0.0;   // (U Utr0)_i
            const double g0 = r0 - yb;                                      // src/projcg.jl:60
            const double ag = (a0 + h) * g0;
            if (valid && owner) {
                if (slot) { slot[sstride] = g0; slot[2 * sstride] = -g0; }
                else {
                    put(gout, o, g0);
                    put(dcg, o, -g0);                                       // :62
                }
                red[1] = fma(r0, g0, red[1]);
                red[2] = fma(g0, g0, red[2]);
                red[3] = fma(g0, ag, red[3]);
                red[4] = fma(w.u, g0, red[4]);                              // the view's rank-one terms of the two second products
                red[5] = fma(w.u, ag, red[5]);
            }
            v[0] = valid ? g0 * s : 0.0;
            v[1] = valid ? ag * s : 0.0;
        }
    }
};

// The same with BOUNDS (src/optimize.jl:312-318, src/inequality_helper.jl:286-308 and :144-158): Q = [[diag Dx; diag Dy], [sx; sy] .* Z], stacked
// vectors [x-half | y-half].  One pass replaces mul!(tmp, Q', d) / mul!(d, Q, tmp, -1, 1), the GEMV-N of calculate_lambda_kkt!, and projcg!'s
// first Q'r:    w = Dx dx + Dy dy;  d -= [Dx w + sx (Z t); Dy w + sy (Z t)];  lamy = (w - Dx (Jct lam)) / S;
//               a = [hx (+ cq on the rows < n_x) + 2 lamy q ; 2 lamy s];  r0 = -d;  second product of sx r0x + sy r0y.
template <bool INITF = false>
struct TangentStepSE {
    static constexpr bool kRowScaled = true;            // (taken as is by run_onepass; views are refused by the caller)
    static constexpr bool kSplitRed = false;
    static constexpr int kNA = INITF ? 3 : 2;           // first products: Z t, (Z Utr0,) Jct lam
    static constexpr int kNV = INITF ? 2 : 1;
    static constexpr int kNRED = INITF ? 6 : 2;
    double* d;          // stacked, in / out
    double* rp;         // stacked, out (not written with INITF)
    double* gout;       // INITF: projcg's g (stacked)
    double* dcg;        // INITF: projcg's d (stacked)
    double* a;          // stacked, out: the diagonal of augmented_hess_lag_vec!
    const double* hx;   // n: the objective's part of the Hessian diagonal on the x-half
    double* lamy;       // n, optional out
    int64_t hs;
    const double *Dx, *Dy, *sx, *sy, *S, *q, *s;
    double cq;
    int64_t n_x;
    struct Uni {};
    struct Row { double dx, dy, Dx, Dy, sx, sy, S, q, s, hx; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ Uni uniform() const { return Uni{}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        return Row{at(d, o), at(d + hs, o), at(Dx, o), at(Dy, o), at(sx, o), at(sy, o), at(S, o), at(q, o), at(s, o), at(hx, o)};
    }
    // staged form (INITF): the nine output streams -- both halves of d, a, g0 and -g0, and lamy -- wait in LDS and leave in bursts
    static constexpr bool kStageAnyNA = true;
    // (measured at (1e7, 129): 15.6 ms per outer iteration staged against 15.2 row by row -- nine streams leave bursts of 15 rounds, too short to
    // pay: the staged form of the STACKED step stays off; LFPSQP_TANGENT_STAGE_STACKED=1 builds it, profiles/r06d_*)
    static constexpr int kStageStreams = (INITF && LFPSQP_TANGENT_STAGE_STACKED) ? 9 : 0;
    __device__ __forceinline__ double* stage_out(int sv) const {
        switch (sv) {
            case 0: return d;
            case 1: return d + hs;
            case 2: return a;
            case 3: return a + hs;
            case 4: return gout;
            case 5: return gout + hs;
            case 6: return dcg;
            case 7: return dcg + hs;
            default: return lamy;       // (nullptr: not wanted)
        }
    }
    __device__ __forceinline__ void apply_staged(int64_t row, uint32_t o, const double (&acc)[kNA], bool valid, bool owner, bool lead, const Uni& u, const Row& w,
                                                 double (&v)[kNV], double (&red)[kNRED], double* slot, int sstride) const {
        apply(row, o, acc, valid, owner, lead, u, w, v, red, slot, sstride);
    }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&acc)[kNA], bool valid, bool owner, bool, const Uni&, const Row& w,
                                          double (&v)[kNV], double (&red)[kNRED], double* slot = nullptr, int sstride = 0) const {
        const double ww = w.Dx * w.dx + w.Dy * w.dy;                        // the diagonal block of Q'd
        const double dxn = w.dx - fma(w.sx, acc[0], w.Dx * ww);             // d - Q [w; t]
        const double dyn = w.dy - fma(w.sy, acc[0], w.Dy * ww);
        const double ly = acc[kNA - 1] * (-1.0 * w.Dx / w.S) + ww / w.S;    // (the reference's order: lamy = Jct lam; lamy *= -Dx / S; lamy += w / S)
        const double ax = (w.hx + ((row < n_x) ? cq : 0.0)) + 2.0 * ly * w.q;
        const double ay = 2.0 * ly * w.s;
        if (valid && owner) {
            if (slot) {
                slot[0] = dxn; slot[sstride] = dyn; slot[2 * sstride] = ax; slot[3 * sstride] = ay; slot[8 * sstride] = ly;
            } else {
                put(d, o, dxn); put(d + hs, o, dyn);
                if (!INITF) { put(rp, o, -dxn); put(rp + hs, o, -dyn); }
                put(a, o, ax);
                put(a + hs, o, ay);
                if (lamy) put(lamy, o, ly);
            }
            red[0] = fma(dxn, dxn, fma(dyn, dyn, red[0]));
        }
        if constexpr (!INITF) {
            v[0] = valid ? -(w.sx * dxn + w.sy * dyn) : 0.0;                // the Z-block of Q'r0
        } else {                                                            // projcg!'s initial projection with Q (src/projcg.jl:58-62, PcgFuseE<true, true>)
            const double rx = -dxn, ry = -dyn;
            const double w0 = w.Dx * rx + w.Dy * ry;
            const double gx = rx - fma(w.sx, acc[1], w.Dx * w0);
            const double gy = ry - fma(w.sy, acc[1], w.Dy * w0);
            const double agx = ax * gx, agy = ay * gy;
            if (valid && owner) {
                if (slot) {
                    slot[4 * sstride] = gx; slot[5 * sstride] = gy; slot[6 * sstride] = -gx; slot[7 * sstride] = -gy;
                } else {
                    put(gout, o, gx); put(gout + hs, o, gy);
                    put(dcg, o, -gx); put(dcg + hs, o, -gy);
                }
                red[1] = fma(rx, gx, fma(ry, gy, red[1]));
                red[2] = fma(gx, gx, fma(gy, gy, red[2]));
                red[3] = fma(gx, agx, fma(gy, agy, red[3]));
            }
            v[0] = valid ? (w.sx * gx + w.sy * gy) : 0.0;
            v[1] = valid ? (w.sx * agx + w.sy * agy) : 0.0;
        }
    }
};

}  // namespace lfpsqp

using namespace lfpsqp;

extern "C" int lfpsqp_tangent_step(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m64, const double* Jtd,
                                   const double* G, lfpsqp_vec* d, const lfpsqp_constraints* cons, const lfpsqp_vec* x, lfpsqp_vec* hdiag,
                                   const lfpsqp_ineq_data* idata, const lfpsqp_vec* hx, const lfpsqp_vec* S, lfpsqp_vec* lamy,
                                   const lfpsqp_projcg_work* work, int flags, double* Utd, double* lam, double* d_sumsq) {
    LF_RANGE("lfpsqp_tangent_step");
    LF_ARG(ctx, ctx && U && Sigma && Vt && Jtd && d && work && work->rp && work->Utr && Utd && lam && d_sumsq && m64 >= 1);
    const int m = (int)m64, rank = (int)U->ncols;
    const bool stacked = U->Dx != nullptr;
    const bool initf = (flags & LFPSQP_TANGENT_INIT_PROJCG) != 0;
    if (U->Z || !U->A || !U->W || U->SA || rank < 1 || rank > m || U->A->m != m || onepass_cw(ctx, m, U->A->ld, U->A->n) == 0 || (stacked && U->A->view))
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "lfpsqp_tangent_step: needs a basis in factored form over a dense generator of 4 .. 1024 columns "
                                                    "(lfpsqp_basis.Z == NULL, A and W given, no sparse twin; with bounds: no matrix view)");
    const lfpsqp_mat* A = U->A;
    const int64_t N = A->n;
    const int64_t hs = stacked ? lfpsqp_half_stride(N) : 0, nv = stacked ? hs + N : N;
    LF_ARG(ctx, d->n >= nv && work->rp->n >= nv && work->Utr->n >= rank && d->p != work->rp->p);
    LF_ARG(ctx, !initf || (G && hdiag && work->g && work->d && work->g->n >= nv && work->d->n >= nv && work->Utr->n >= 3 * (int64_t)m + 5 &&
                           work->g->p != d->p && work->d->p != d->p));
    // U'U = W'GW is only as good as the Gram matrix resolves the small singular values: its rounding, eps * sigma_1^2, enters divided by sigma_j^2.
    // The fold is therefore offered exactly where lfpsqp_factorize itself trusts G (its fast path, cond^2 <= 10: BASELINE's random blocks have
    // cond 1.1); an ill-conditioned block keeps projcg!'s own measurement of U'r0 (LFPSQP_PROJCG_START_GIVEN, one pass more).
    if (initf && !(rank == m && Sigma[0] * Sigma[0] <= 10.0 * Sigma[m - 1] * Sigma[m - 1]))
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "LFPSQP_TANGENT_INIT_PROJCG: needs a full-rank block with cond^2 <= 10 (U'U from the Gram matrix); "
                                                    "call without the flag and start projcg with LFPSQP_PROJCG_START_GIVEN");
    // the class's Hessian term: which part of it this pass can carry
    int mode = 0;
    double cq = 0.0;
    const double* kind = nullptr;
    // 1. the replicated part: tmp_m = U'd = W'(A'd), lambda = V S^-1 tmp_m (:331-343), u1 = W tmp_m
    std::vector<double> u1((size_t)m, 0.0), uB((size_t)m, 0.0);
    for (int j = 0; j < rank; ++j) {
        const double* wj = U->W + (size_t)j * m;
        double sdot = 0.0;
        for (int k = 0; k < m; ++k) sdot = fma(wj[k], Jtd[k], sdot);
        Utd[j] = sdot;
    }
    for (int j = rank; j < m; ++j) Utd[j] = 0.0;
    for (int i = 0; i < m; ++i) lam[i] = 0.0;
    for (int j = 0; j < rank; ++j) {
        const double tj = Utd[j] / Sigma[j];
        for (int i = 0; i < m; ++i) lam[i] = fma(Vt[(size_t)i * m + j], tj, lam[i]);       // lam = Vt' (tmp ./ Sigma): Vt[j, i] at [j + i m]
        const double* wj = U->W + (size_t)j * m;
        for (int k = 0; k < m; ++k) u1[k] = fma(wj[k], Utd[j], u1[k]);
    }
    if (initf) {
        // Utr0 = U'r0 = -(U'd - (U'U) U'd) with U'U = W'GW (src/projcg.jl:59 measures it with a pass; here it follows from the Gram matrix the
        // factors came from -- exact arithmetic gives 0, what is left is the departure of the basis from orthonormality), and uB = W Utr0
        std::vector<double> gz((size_t)m, 0.0);
        for (int k = 0; k < m; ++k) {
            const double* gk = G + (size_t)k * m;
            const double c = u1[k];
            for (int i = 0; i < m; ++i) gz[i] = fma(gk[i], c, gz[i]);                       // G (W tmp_m)
        }
        for (int j = 0; j < rank; ++j) {
            const double* wj = U->W + (size_t)j * m;
            double sdot = 0.0;
            for (int k = 0; k < m; ++k) sdot = fma(wj[k], gz[k], sdot);                      // ((U'U) tmp_m)_j
            const double utr0 = -(Utd[j] - sdot);
            for (int k = 0; k < m; ++k) uB[k] = fma(wj[k], utr0, uB[k]);
        }
    }
    if (cons) {
        LF_ARG(ctx, x && hdiag && x->n >= N && hdiag->n >= nv && hdiag->p != x->p && hdiag->p != d->p && cons->Jct && cons->m_lin >= 0 &&
                        cons->m_lin + (cons->has_ball ? 1 : 0) <= m);
        const lfpsqp_elementwise* ew = cons->ew;
        const int ml = (int)cons->m_lin;
        cq = cons->has_ball ? 2.0 * lam[ml] : 0.0;
        if (ew && ew->qw)
            for (int j = 0; j < ml; ++j) cq += 2.0 * ew->qw[j] * lam[j];
        kind = (ew && ew->kind) ? ew->kind->p : nullptr;
        if (stacked && kind)
            return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "lfpsqp_tangent_step: bounds with a nonlinear constraint class (phi'' term): statement-by-statement sequence");
        if (kind && !(ew->A && !ew->Asp && ew->A->p == A->p && ew->A->ld == A->ld && ml <= m)) {
            // the class does not stream its gradients from the matrix this pass reads (a materialised Jct, or sparse A): its own pass
            LF_TRY(lfpsqp_constraints_hess_diag(ctx, cons, x, lam, hdiag));
        } else if (kind) mode = 2;
        else if (cq != 0.0) mode = 1;
    }
    // 2. coefficients of the first products onto the device: [u1 | (uB) | lam], each ms long; then the raw sums and the folded ones
    const int ms = (int)round_up(m, 2);
    LF_TRY(ensure_mvec(ctx, (size_t)7 * ms + 64));
    const int nlam = stacked ? m : ((mode == 2) ? (int)cons->m_lin : 0);     // last coefficient vector: lam (bounds: Jct lam; nonlinear class: A lam)
    const int slotB = initf ? 1 : -1, slotL = initf ? 2 : 1;
    for (int k = 0; k < m; ++k) {
        ctx->h_m[k] = u1[k];
        if (slotB >= 0) ctx->h_m[slotB * ms + k] = uB[k];
        ctx->h_m[slotL * ms + k] = k < nlam ? lam[k] : 0.0;
    }
    // (only the slots written above travel: 2 or 3 coefficient vectors; the layout of d_m -- t | raw | folded inside 7 ms + 64 doubles -- is checked below)
    LF_HIP(ctx, hipMemcpyAsync(ctx->d_m, ctx->h_m, sizeof(double) * (size_t)(slotL + 1) * ms, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));       // h_m is the context's shared pinned staging block (the next call may rewrite it)
    double* t = ctx->d_m;
    double* raw = ctx->d_m + 3 * ms;                        // [second products (1 or 2 x m) ; the functor's sums ; u'v terms]
    double* folded = raw + 2 * ms + 8;                      // the same with a view's rank-one term folded into the second products
    const int nvp = initf ? 2 : 1, nredf = initf ? 4 : 1;   // second-product vectors; functor sums ahead of the u'v terms
    LF_ARG(ctx, nvp * m + (initf ? 6 : 2) <= 2 * ms + 8 && nvp * m + nredf <= 2 * ms + 56);        // raw and folded hold what the pass and the fold write
    double *dW, *tA, *uA;
    LF_TRY(factored_setup(ctx, A, U->W, rank, &dW, &tA, &uA));
    double* stash = work->Utr->p + m;                       // INIT_PROJCG: [t1 (m); t2 (m); r0'g0; g0'g0; g0'A g0; 0; 0] for lfpsqp_projcg's START_PROJECTED
    auto finish = [&](const double* wfold) -> int {
        hipLaunchKernelGGL((view_fold_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, raw, folded, wfold, nvp, m, nredf);
        LF_LAUNCH_CHECK(ctx);
        if (!initf) {
            LF_TRY(sp_basis_small(ctx, dW, m, rank, folded, work->Utr->p, nullptr));
        } else {
            LF_TRY(sp_basis_small(ctx, dW, m, rank, folded, stash, nullptr));
            LF_TRY(sp_basis_small(ctx, dW, m, rank, folded + m, stash + rank, nullptr));
            LF_HIP(ctx, hipMemsetAsync(stash + 2 * rank, 0, sizeof(double) * 5, ctx->stream));
            LF_HIP(ctx, hipMemcpyAsync(stash + 2 * rank, folded + 2 * m + 1, sizeof(double) * 3, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return read_back(ctx, folded + nvp * m, d_sumsq, 1);
    };
    if (stacked) {
        LF_ARG(ctx, idata && hx && S && hdiag && U->Dy && U->sx && U->sy && idata->q && idata->s && U->Dx->n == N && S->n == N && hx->n >= N &&
                        hdiag->n >= nv && (!lamy || lamy->n >= N) && hdiag->p != d->p);
        const int64_t n_x = cons ? cons->n_x : 0;
        if (initf) {
            const TangentStepSE<true> se{d->p, work->rp->p, work->g->p, work->d->p, hdiag->p, hx->p, lamy ? lamy->p : nullptr, hs, U->Dx->p, U->Dy->p,
                                         U->sx->p, U->sy->p, S->p, idata->q->p, idata->s->p, cq, n_x};
            LF_TRY((run_onepass<TangentStepSE<true>, 2, 6, 3>(ctx, A, m, m, N, t, se, raw, 6, ms)));
        } else {
            const TangentStepSE<false> se{d->p, work->rp->p, nullptr, nullptr, hdiag->p, hx->p, lamy ? lamy->p : nullptr, hs, U->Dx->p, U->Dy->p,
                                          U->sx->p, U->sy->p, S->p, idata->q->p, idata->s->p, cq, n_x};
            LF_TRY((run_onepass<TangentStepSE<false>, 1, 2, 2>(ctx, A, m, m, N, t, se, raw, 6, ms)));
        }
        return finish(nullptr);
    }
    ViewD vw{nullptr, nullptr, nullptr};
    const lfpsqp_mat plain = A->plain();
    if (A->view) {
        vw = ViewD{A->rs, A->ru, nullptr};
        if (A->ru) {                                        // tau = w'u1 (and w'uB) ahead of the launch
            LF_TRY(ensure_view(ctx));
            hipLaunchKernelGGL((view_tau_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, A->rw, t, m, ctx->d_view);
            if (initf) hipLaunchKernelGGL((view_tau_kernel<0>), dim3(1), dim3(256), 0, ctx->stream, A->rw, t + ms, m, ctx->d_view + 1);
            LF_LAUNCH_CHECK(ctx);
            vw.tau = ctx->d_view;
        }
    }
    // 3. the pass
    double* hxp = hdiag ? hdiag->p : nullptr;
    const double* xp = x ? x->p : nullptr;
    const int64_t n_x = cons ? cons->n_x : 0;
    if (initf) {
        double *gp = work->g->p, *dc = work->d->p;
        if (mode == 2) LF_TRY((run_onepass<TangentStepE<2, true>, 2, 6, 3>(ctx, &plain, m, m, N, t, TangentStepE<2, true>{d->p, work->rp->p, gp, dc, 0.0, hxp, xp, kind, cq, n_x, vw}, raw, 6, ms)));
        else if (mode == 1) LF_TRY((run_onepass<TangentStepE<1, true>, 2, 6, 2>(ctx, &plain, m, m, N, t, TangentStepE<1, true>{d->p, work->rp->p, gp, dc, 0.0, hxp, xp, kind, cq, n_x, vw}, raw, 6, ms)));
        else LF_TRY((run_onepass<TangentStepE<0, true>, 2, 6, 2>(ctx, &plain, m, m, N, t, TangentStepE<0, true>{d->p, work->rp->p, gp, dc, 0.0, hxp, xp, kind, cq, n_x, vw}, raw, 6, ms)));
    } else {
        if (mode == 2) LF_TRY((run_onepass<TangentStepE<2>, 1, 2, 2>(ctx, &plain, m, m, N, t, TangentStepE<2>{d->p, work->rp->p, nullptr, nullptr, 0.0, hxp, xp, kind, cq, n_x, vw}, raw, 6, ms)));
        else if (mode == 1) LF_TRY((run_onepass<TangentStepE<1>, 1, 2, 1>(ctx, &plain, m, m, N, t, TangentStepE<1>{d->p, work->rp->p, nullptr, nullptr, 0.0, hxp, xp, kind, cq, n_x, vw}, raw)));
        else LF_TRY((run_onepass<TangentStepE<0>, 1, 2, 1>(ctx, &plain, m, m, N, t, TangentStepE<0>{d->p, work->rp->p, nullptr, nullptr, 0.0, hxp, xp, kind, cq, n_x, vw}, raw)));
    }
    // 4. the second products leave the view (+ w (u'v)) and the generator (W')
    return finish((A->view && A->ru) ? A->rw : nullptr);
}
