// Retractions onto the constraint manifold (reference src/retractions.jl).
//   * device-resident constraint classes (linear equalities + ball with slack) so that c! and
//     jac! need no n-sized PCIe traffic;
//   * the Newton retraction retract!(..., ::NR) (:75-177): per iteration ONE pass over U
//     (x += U*delta, stacked when bounds exist), the fused y_retract!, and ONE pass over Jct for c!;
//     the m x m inverse-Jacobian / Broyden algebra (:126-130, :140, :156-160) is replicated
//     host math on m-vectors.
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "internal.h"
#include "sparse.h"
#include "smallla.h"

namespace lfpsqp {

enum { I_NR_STATUS = 8, I_NR_ITER = 9, I_NR_FLAG = 10 };   // istat slots of the Newton retraction
constexpr int kNRRing = 8, kNRRingOff = 24;                 // per-iteration host status ring (see projcg.hip HostMirror)
constexpr int kNRMaxM = 1024;                               // m x m Broyden state handled by one workgroup

// rows of D0^-1 for the stacked operator of ProjPenalty's inner solve: D0 = mu I + [a; b][a b] per variable (a = Dx.*S, b = Dy.*S), so
// det = mu (a^2 + b^2 + mu) and D0^-1 = [b^2 + mu, -ab; -ab, a^2 + mu] / det
struct D0InvF {
    const double *a, *b;
    double *i11, *i12, *i22;
    double mu;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 aa = ld2(a + i), bb = ld2(b + i);
        const double dx = mu * (aa.x * aa.x + bb.x * bb.x + mu), dy = mu * (aa.y * aa.y + bb.y * bb.y + mu);
        const double2 o11 = make_double2((bb.x * bb.x + mu) / dx, (bb.y * bb.y + mu) / dy);
        const double2 o12 = make_double2(-(aa.x * bb.x) / dx, -(aa.y * bb.y) / dy);
        const double2 o22 = make_double2((aa.x * aa.x + mu) / dx, (aa.y * aa.y + mu) / dy);
        if (v1) { st2(i11 + i, o11); st2(i12 + i, o12); st2(i22 + i, o22); }
        else if (v0) { i11[i] = o11.x; i12[i] = o12.x; i22[i] = o22.x; }
    }
};

struct BallF {  // partial of sum_{i<n_x} x_i^2 - x[slack_row]
    const double* x;
    int64_t n_x, slack_row;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(x + i);
        double s = 0.0;
        if (v0) s += (i < n_x) ? a.x * a.x : ((i == slack_row) ? -a.x : 0.0);
        if (v1) s += (i + 1 < n_x) ? a.y * a.y : ((i + 1 == slack_row) ? -a.y : 0.0);
        red[0] += s;
    }
};

// Newton step when the constraint gradients have a sparse twin and the basis its generator (Z = Jct * W): U*ddelta = Jct*(W*ddelta), so the
// step needs no pass over a dense matrix at all -- a row's entry of Jct*(W ddelta) from its ELL entries (+ the dense extra columns, e.g.
// the ball column), the same row update as the dense kernels (NRStepE::apply), then c! = S'xnew by the sparse product.
struct NRSparseStepF;   // (defined after NRStepE)

struct BallColF {  // Jct[:, m_lin] = [2x (i < n_x); -1 at slack_row; 0 elsewhere]
    const double* x;
    double* col;
    int64_t n_x, slack_row;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double one(int64_t i, double xi) const { return i < n_x ? 2.0 * xi : (i == slack_row ? -1.0 : 0.0); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 a = ld2(x + i);
        if (v1) st2(col + i, make_double2(one(i, a.x), one(i + 1, a.y)));
        else if (v0) col[i] = one(i, a.x);
    }
};

struct PlainVec {  // GEMV-T producer reading the first N entries of a (possibly stacked) vector
    const double* v;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 a = ld2(v + r);
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};

// y_retract! for one variable -- defined in ineq.hip's translation unit as a device inline; repeated here
// (same statements, src/retractions.jl:459-497) so the fused Newton step can apply it in registers.
// In two parts: what depends on the ROW alone (the base point and the bound data: a square root and two divisions of the parabola, a square
// root of the circle) and what depends on the trial point -- the batched step (nrbatch.h) does the first part once per row for all of a lane's
// trial points; the same statements in the same order either way, so the single and the batched step round alike.
struct YRowPre { double gx, gy, rho; };
__device__ __forceinline__ YRowPre y_retract_pre(double yo, double qq, double rr, double ss, double tt) {
    YRowPre p{0.0, 0.0, 0.0};
    if (ss == 0.0) {
    } else if (qq == 0.0) {
        const double g1 = -ss, g2 = -2.0 * (yo - rr);
        const double ng = sqrt(g1 * g1 + g2 * g2);
        p.gx = g1 / ng;
        p.gy = g2 / ng;
    } else {
        p.rho = sqrt(tt);
    }
    return p;
}
__device__ __forceinline__ void y_retract_apply(double& xn, double& yn, double xo, double yo, double qq, double rr, double ss, const YRowPre& p) {
    if (ss == 0.0) {
        xn = yn;
    } else if (qq == 0.0) {
        const double ux = xo - xn + p.gx;
        const double uy = yo - yn + p.gy;
        const double a = ss * (uy * uy);
        const double b = ux + 2.0 * ss * (yn - rr) * uy;
        const double c = xn + ss * ((yn - rr) * (yn - rr)) - rr;
        const double a1 = -b / (2.0 * a);
        const double a2 = sqrt(b * b - 4.0 * a * c) / (2.0 * a);
        const double gam = fmin(a1 + a2, a1 - a2);
        xn += gam * ux;
        yn += gam * uy;
    } else {
        const double c = rr, rho = p.rho;
        const double dist = sqrt((xn - c) * (xn - c) + (yn - c) * (yn - c));
        const double y2 = c + rho * (yn - c) / dist;
        const double x2 = c + rho * (xn - c) / dist;
        yn = y2;
        xn = x2;
    }
}
__device__ __forceinline__ void y_retract_one_nr(double& xn, double& yn, double xo, double yo, double qq, double rr, double ss, double tt) {
    const YRowPre p = y_retract_pre(yo, qq, rr, ss, tt);
    y_retract_apply(xn, yn, xo, yo, qq, rr, ss, p);
}

// (the elementwise transforms ew_phi / ew_phi1 / ew_phi2 of the nonlinear constraint class lfpsqp_elementwise: kernels.h)

// Fused Newton-retraction step (src/retractions.jl:141-149): xnew += U*delta (stacked when bounds exist),
// y_retract!, and the value handed to the c! product; one reduction term = the ball partial.
struct NRStepE {
    double* xnew;
    const double* xold;
    int64_t hs;                      // 0 => no bounds (plain basis)
    const double *sx, *sy;           // row scalings of the stacked basis
    const double *q, *r, *s, *t;     // InequalityData
    int64_t n_x, slack_row;          // ball term: sum_{i<n_x} x_i^2 - x[slack_row]  (slack_row < 0: none here)
    int has_ball;                    // (also set for the common quadratic term of lfpsqp_elementwise: the same partial sum)
    const int64_t* istat;            // NR status word: a finished retraction turns further launches into no-ops
    const double* kind = nullptr;    // lfpsqp_elementwise: the c! product takes phi(xnew) instead of xnew  (two-stream / sparse step only)
    double* phi_out = nullptr;       // ... stored here when the product is a separate (sparse) launch
    // One-stream step of the nonlinear class with a dense A (onepass_kernel over A itself): the step product over Jct(x) = diag(phi'(x)) A +
    // 2 x qw' is rebuilt from the kernel's first product over A -- (Jct u)_i = phi'(xold_i) (A u)_i + 2 xold_i (qw . u) [i < n_x] -- and the
    // second product takes phi(xnew).  sq = qw . u, a device scalar written by the small kernel (nullptr: no quadratic term).
    int ew_one = 0;
    const double* sq = nullptr;
    int eval_only = 0;               // the same launch as c!: no update, no store, products of phi(x) (so cval is bit for bit c!(xnew))
    __device__ __forceinline__ bool skip() const { return !eval_only && ld_stat(istat + I_NR_STATUS) != 0; }
    __device__ __forceinline__ double ball(int64_t i, double xi) const {
        return !has_ball ? 0.0 : (i < n_x ? xi * xi : (i == slack_row ? -xi : 0.0));
    }
    __device__ __forceinline__ double2 apply(int64_t i, double2 acc, bool v0, bool v1, double* red) const {
        double2 xn = ld2(xnew + i);
        if (hs == 0) {
            xn.x += acc.x; xn.y += acc.y;
            if (v1) st2(xnew + i, xn);
            else if (v0) xnew[i] = xn.x;
        } else {
            const double2 ax = ld2(sx + i), ay = ld2(sy + i);
            double2 yn = ld2(xnew + hs + i);
            xn.x += ax.x * acc.x; xn.y += ax.y * acc.y;
            yn.x += ay.x * acc.x; yn.y += ay.y * acc.y;
            const double2 xo = ld2(xold + i), yo = ld2(xold + hs + i);
            const double2 qq = ld2(q + i), rr = ld2(r + i), ss = ld2(s + i), tt = ld2(t + i);
            if (v0) y_retract_one_nr(xn.x, yn.x, xo.x, yo.x, qq.x, rr.x, ss.x, tt.x);
            if (v1) y_retract_one_nr(xn.y, yn.y, xo.y, yo.y, qq.y, rr.y, ss.y, tt.y);
            if (v1) { st2(xnew + i, xn); st2(xnew + hs + i, yn); }
            else if (v0) { xnew[i] = xn.x; xnew[hs + i] = yn.x; }
        }
        double b = 0.0;
        if (v0) b += ball(i, xn.x);
        if (v1) b += ball(i + 1, xn.y);
        red[0] += b;
        if (kind) {
            const double2 kk = ld2(kind + i);
            xn.x = ew_phi(kk.x, xn.x);
            xn.y = ew_phi(kk.y, xn.y);
            if (phi_out) {
                if (v1) st2(phi_out + i, xn);
                else if (v0) phi_out[i] = xn.x;
            }
        }
        return make_double2(v0 ? xn.x : 0.0, v1 ? xn.y : 0.0);
    }
    // one-row form for the one-stream step kernel: the inputs of a row are fetched a tile ahead of their use.
    // `o` is the row's BYTE offset (uniform base + 32-bit lane offset: the scalar-base addressing mode).
    struct Row { double xn, yn, xo, yo, ax, ay, q, r, s, t, kk; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    template <bool ST>
    __device__ __forceinline__ Row fetch1(uint32_t o) const {
        Row w;
        w.xn = at(xnew, o);
        if (ST) {
            w.yn = at(xnew + hs, o); w.xo = at(xold, o); w.yo = at(xold + hs, o); w.ax = at(sx, o); w.ay = at(sy, o);
            w.q = at(q, o); w.r = at(r, o); w.s = at(s, o); w.t = at(t, o);
        } else {
            w.yn = w.xo = w.yo = w.ax = w.ay = w.q = w.r = w.s = w.t = 0.0;
        }
        w.kk = 0.0;
        if (ew_one) {
            if (!ST && !eval_only) w.xo = at(xold, o);
            if (kind) w.kk = at(kind, o);
        }
        return w;
    }
    template <bool ST>
    __device__ __forceinline__ double apply1(int64_t i, uint32_t o, double acc, bool valid, bool owner, const Row& w, double& red,
                                             double* slot = nullptr, int sstride = 0, double sqv = 0.0, const YRowPre* pre = nullptr) const {
        double xn = w.xn;
        if (ew_one && !eval_only) acc = fma(ew_phi1(w.kk, w.xo), acc, (i < n_x) ? 2.0 * w.xo * sqv : 0.0);      // (Jct(xold) u)_i from (A u)_i
        if (eval_only) {
            // c! alone: xn is the point itself
        } else if (!ST) {
            xn += acc;
            if (valid && owner) {
                if (slot) *slot = xn;      // staged stores (onepass_kernel STG)
                else put(xnew, o, xn);
            }
        } else {
            double yn = w.yn;
            xn += w.ax * acc;
            yn += w.ay * acc;
            if (valid) {
                if (pre) y_retract_apply(xn, yn, w.xo, w.yo, w.q, w.r, w.s, *pre);       // (the row's part done once by the caller, nrbatch.h)
                else y_retract_one_nr(xn, yn, w.xo, w.yo, w.q, w.r, w.s, w.t);
            }
            if (valid && owner) {
                if (slot) { slot[0] = xn; slot[sstride] = yn; }
                else { put(xnew, o, xn); put(xnew + hs, o, yn); }
            }
        }
        if (valid && owner) red += ball(i, xn);
        if (ew_one) xn = ew_phi(w.kk, xn);
        return valid ? xn : 0.0;
    }
};

// One-stream Newton step (onepass_kernel, kernels.h).  When the basis was formed as Z = Jct * W (lfpsqp_factorize), the
// step product is U*delta = [sx; sy] .* (Jct * (W*delta)), so BOTH products of a Newton step (src/retractions.jl:141 and
// the c! of :146/148) run over the same rows of the same matrix: Jct is streamed once per step instead of Z plus Jct.
//   t = W*delta; out[0:m_lin) = Jct[:, :m_lin]' xnew; out[m_lin] = ball partial.   ST = stacked vectors (bounds).
template <bool ST>
struct NRStepRow {
    NRStepE e;
    using Row = NRStepE::Row;
    struct Uni { double sq; };
    static constexpr bool kSplitRed = false;
    __device__ __forceinline__ bool skip() const { return e.skip(); }
    __device__ __forceinline__ Uni uniform() const { return Uni{(e.ew_one && e.sq && !e.eval_only) ? uniform_f64(ld_scal(e.sq)) : 0.0}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const { return e.fetch1<ST>(o); }
    __device__ __forceinline__ void apply(int64_t i, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool, const Uni& u,
                                          const Row& w, double (&v)[1], double (&red)[1]) const {
        v[0] = e.apply1<ST>(i, o, acc[0], valid, owner, w, red[0], nullptr, 0, u.sq);
    }
    // staged form: the new point (stacked: its x and y halves) waits in LDS and is stored in bursts
    static constexpr int kStageStreams = ST ? 2 : 1;
    __device__ __forceinline__ double* stage_out(int sv) const { return e.eval_only ? nullptr : (sv == 0 ? e.xnew : e.xnew + e.hs); }
    __device__ __forceinline__ void apply_staged(int64_t i, uint32_t o, const double (&acc)[1], bool valid, bool owner, bool, const Uni& u,
                                                 const Row& w, double (&v)[1], double (&red)[1], double* slot, int sstride) const {
        v[0] = e.apply1<ST>(i, o, acc[0], valid, owner, w, red[0], slot, sstride, u.sq);
    }
};

// The replicated m x m part of the Newton retraction on the device (src/retractions.jl:126-160): turns the raw
// constraint products of the fused step into c(xnew), applies the good-Broyden update of the inverse Jacobian
// D, tests ||c||_inf < tol / the iteration limit, and prepares delta = -D c (and W*delta for the one-stream
// step) for the next step.  One workgroup of 1024 threads; D is column-major, products over its rows are
// coalesced over k with the column range split over thread groups, products over its columns use one wave
// per column; all sums in a fixed order.
struct NRSmall {
    double* D;            // m x m
    const double* Vt;     // m x m
    const double* Sigma;  // m
    const double* b;      // m_lin
    double* cval;         // m (current constraint values)
    double* delta;        // m (step coefficients, read by the fused step kernel)
    const double* raw;    // m_lin: J'x
    const double* raw_ball;   // 1: ball partial
    const double* W;      // wm x m generator factor (Z = A*W) or nullptr
    double* wdelta;       // wm: W * delta
    int64_t* ist;         // this retraction's [status, iterations, flag] words (device)
    int64_t* hstat;       // pinned host block (nullptr: the caller publishes)
    int m, m_lin, has_ball, wm;
    double R2, tol;
    int64_t maxiter;
    const double* qw = nullptr;   // m_lin weights of the common quadratic term (lfpsqp_elementwise) or nullptr
};
constexpr int kNRThreads = 1024;

// y[k] = sum_j A[k + j*lda] * x[j], k < rows, j < cols (x in LDS); result left in y (LDS), valid after the trailing barrier
__device__ __forceinline__ void small_matvec(const double* A, int lda, int rows, int cols, const double* x, double* y, double* scratch) {
    int kw = 32;
    while (kw < rows) kw <<= 1;                 // rows <= kNRMaxM = 1024
    const int groups = kNRThreads / kw;
    const int k = threadIdx.x & (kw - 1), g = threadIdx.x / kw;
    double acc = 0.0;
    if (k < rows)
        for (int j = g; j < cols; j += groups) acc += A[(size_t)j * lda + k] * x[j];
    scratch[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && k < rows) {
        double sum = 0.0;
        for (int q = 0; q < groups; ++q) sum += scratch[q * kw + k];
        y[k] = sum;
    }
    __syncthreads();
}

__device__ void nr_small_trial(const NRSmall& s, int init) {
    if (!init && ld_stat(s.ist) != 0) return;
    __shared__ double cnew[kNRMaxM], del[kNRMaxM], t2[kNRMaxM], dcs[kNRMaxM], tv[kNRMaxM];
    __shared__ double scratch[kNRThreads];
    __shared__ int oldnan;                              // the PREVIOUS constraint values held a NaN (see the retirement rule below)
    const int m = s.m, tid = threadIdx.x;
    if (tid == 0) oldnan = 0;
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();                                    // (a previous trial of a batch may still be reading the arrays)
    for (int k = tid; k < m; k += kNRThreads) {
        if (k < s.m_lin) cnew[k] = s.qw ? fma(s.qw[k], ld_scal(s.raw_ball), s.raw[k]) - s.b[k] : (s.raw[k] - s.b[k]);   // (explicit fma: the host's cons_eval rounds alike)
        else cnew[k] = ld_scal(s.raw_ball) - s.R2;
    }
    __syncthreads();
    int64_t iter = init ? 0 : ld_stat(s.ist + 1);
    if (init) {
        for (int k = tid; k < m; k += kNRThreads) s.cval[k] = cnew[k];
        for (size_t e = tid; e < (size_t)m * m; e += kNRThreads) s.D[e] = s.Vt[e] / s.Sigma[e % m];      // :126-130
    } else {
        for (int k = tid; k < m; k += kNRThreads) {
            const double cold = s.cval[k];
            if (cold != cold) oldnan = 1;                                                       // (every writer writes 1)
            dcs[k] = cnew[k] - cold;                                                            // :152
            s.cval[k] = cnew[k];                                                                // :153
            del[k] = s.delta[k];
        }
        __syncthreads();
        for (int j = wave; j < m; j += kNRThreads / 64) {                                       // :156  t2 = D' delta
            double a = 0.0;
            for (int k = lane; k < m; k += 64) a += s.D[(size_t)j * m + k] * del[k];
            a = wave_sum(a);
            if (lane == 0) t2[j] = a;
        }
        small_matvec(s.D, m, m, m, dcs, tv, scratch);                                           // :157  tv = delta - D dc
        if (tid < m) tv[tid] = del[tid] - tv[tid];
        double part = 0.0;
        for (int k = tid; k < m; k += kNRThreads) part += t2[k] * dcs[k];
        scratch[tid] = part;
        __syncthreads();
        for (int o = kNRThreads / 2; o > 0; o >>= 1) {
            if (tid < o) scratch[tid] += scratch[tid + o];
            __syncthreads();
        }
        const double alpha = 1.0 / scratch[0];                                                  // :159
        __syncthreads();
        {                                                                                       // :160  D += alpha tv t2'
            int kw = 32;
            while (kw < m) kw <<= 1;
            const int groups = kNRThreads / kw, k = tid & (kw - 1), g = tid / kw;
            if (k < m) {
                const double a = alpha * tv[k];
                for (int j = g; j < m; j += groups) s.D[(size_t)j * m + k] += a * t2[j];
            }
        }
        iter += 1;                                                                              // :168
    }
    __syncthreads();
    double mx = 0.0;
    for (int k = tid; k < m; k += kNRThreads) mx = nanmax(mx, fabs(cnew[k]));
    scratch[tid] = mx;
    __syncthreads();
    for (int o = kNRThreads / 2; o > 0; o >>= 1) {
        if (tid < o) scratch[tid] = nanmax(scratch[tid], scratch[tid + o]);
        __syncthreads();
    }
    if (tid == 0) {
        const double c = scratch[0];
        int64_t st = 0, fl = 0;
        int64_t iter_out = iter;
        if (iter >= s.maxiter) { st = 1; fl = 1; }                       // :133 loop bound first: flag = (i == maxiter), :171-174
        else if (c < s.tol) { st = 1; fl = 0; }                          // :135
        else if (!init && oldnan && c != c) {
            // RETIRED: the constraint values were NaN before this step and are NaN after it.  NaN is absorbing here -- delta = -D c is NaN in
            // every component, so the step just taken made every entry of the iterate NaN, and every later c!, Broyden update and step stays
            // NaN -- so the reference's loop (src/retractions.jl:133-168: `norm(cval, Inf) < tol` is false for NaN) runs on to maxiter and returns
            // flag 1 with maxiter iterations, an all-NaN iterate and NaN constraint values: exactly the state published here, maxiter - iter
            // passes over the matrix earlier.  (The diverged trial steps of a failing line search end this way: config 4's first searches
            // retire 9 of 16 trials.)
            st = 1; fl = 1; iter_out = s.maxiter;
        }
        if (st) s.ist[2] = fl;
        s.ist[1] = iter_out;
        s.ist[0] = st;
        if (s.hstat) {
            __hip_atomic_store(s.hstat + I_NR_ITER, iter_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(s.hstat + I_NR_FLAG, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(s.hstat + kNRRingOff + ((iter + 1) % kNRRing), st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(s.hstat + I_NR_STATUS, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        scratch[0] = (double)st;
    }
    __syncthreads();
    const bool finished = scratch[0] != 0.0;
    __syncthreads();
    if (finished) return;
    small_matvec(s.D, m, m, m, cnew, del, scratch);                                             // :140  delta = -D cval
    if (tid < m) { del[tid] = -del[tid]; s.delta[tid] = del[tid]; }
    __syncthreads();
    if (s.W) {                                                                                  // t = W delta (one-stream step)
        small_matvec(s.W, s.wm, s.wm, m, del, t2, scratch);
        if (tid < s.wm) s.wdelta[tid] = t2[tid];
        if (s.qw && tid == 0) {                                                                 // qw . t behind it (one-stream step of the nonlinear class)
            double sq = 0.0;
            for (int j = 0; j < s.m_lin && j < s.wm; ++j) sq = fma(s.qw[j], t2[j], sq);
            s.wdelta[s.wm] = sq;
        }
    }
}
__global__ __launch_bounds__(kNRThreads) void nr_small_kernel(NRSmall s, int init) { nr_small_trial(s, init); }

// ---- batched Newton retractions: NB independent trial points share every pass over Jct ---------------------------------
// (the trial steps alpha, alpha*s, alpha*s^2, ... of an Armijo search whose retractions fail, src/linesearch.jl:57-60)
struct NRSparseStepF {
    NRStepE e;
    EllRows E;                       // E.t = W * ddelta (first S.m entries)
    const double* xcol;              // dense extra columns of Jct (column-major, leading dimension ldx), nx <= 4 of them
    int64_t ldx;
    const double* wx;                // their coefficients: (W * ddelta)[S.m ..]
    int nx;
    __device__ __forceinline__ bool skip() const { return e.skip(); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        double2 acc = E.acc(i);
        for (int j = 0; j < nx; ++j) {
            const double w = ld_scal(wx + j);
            const double2 c = ld2(xcol + (int64_t)j * ldx + i);
            acc.x = fma(c.x, w, acc.x);
            acc.y = fma(c.y, w, acc.y);
        }
        e.apply(i, acc, v0, v1, red);
    }
};

constexpr int kNRBatchMax = 16;                           // (2 trials: the VALU form of the one-pass kernel, which also serves shapes the matrix-core form does not cover, up to 4; 3..16: the matrix cores, nrbatch.h)
enum { I_NRB = 32, I_NRB_ALL = I_NRB + 4 * kNRBatchMax };   // istat: [status, iterations, flag, -] per trial, then the all-done word
struct NRSmallB {
    NRSmall t[kNRBatchMax];
    int nb;
    int64_t* all;         // device: 1 when every trial has finished
    int64_t* hstat;
};
// one workgroup per trial, then a one-thread kernel that forms and publishes the all-done word
__global__ __launch_bounds__(kNRThreads) void nr_small_batch_kernel(NRSmallB s, int init) {
    if (!init && ld_stat(s.all) != 0) return;
    const NRSmall& t = s.t[blockIdx.x];
    if (ld_stat(t.ist) == 0) nr_small_trial(t, init);       // (a padding trial is born finished)
}
__global__ void nr_batch_all_kernel(NRSmallB s, int init, int64_t step) {
    if (!init && ld_stat(s.all) != 0) return;
    int64_t all = 1;
    for (int b = 0; b < s.nb; ++b) all &= (ld_stat(s.t[b].ist) != 0);
    *s.all = all;
    __hip_atomic_store(s.hstat + kNRRingOff + ((step + 1) % kNRRing), all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(s.hstat + I_NR_STATUS, all, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// row functor of the batched one-stream step.  The four lane groups H of a row all see the same first-product results, so
// they SPLIT the trials: group H does the row update (y_retract!: square roots and divisions, the expensive part) of trial
// H mod NB only, stores it, and the groups then exchange their results for the second product.
template <bool ST, int NB>
struct NRStepBatchRow {
    NRStepE e;                    // shared fields; e.xnew is trial 0's iterate
    double* xnew[NB];
    const int64_t* ist[NB];       // per-trial status words
    const int64_t* all;
    struct Row { NRStepE::Row sh; unsigned active; };       // sh.xn / sh.yn: this lane group's trial
    using Uni = NoUni;
    static constexpr bool kSplitRed = false;
    static constexpr bool kLaccAnyNA = true;      // (running sums of the second product in LDS: 2 waves per SIMD instead of 1 at 17..33 column groups)
    __device__ __forceinline__ Uni uniform() const { return Uni{}; }
    __device__ __forceinline__ bool skip() const { return ld_stat(all) != 0; }
    static __device__ __forceinline__ int my_trial() { return (int)((threadIdx.x >> 2) & 3u) % NB; }   // lane bits 3..2 = H
    __device__ __forceinline__ double* my_xnew(int tr) const {
        double* p = xnew[0];
#pragma unroll
        for (int b = 1; b < NB; ++b) p = (tr == b) ? xnew[b] : p;
        return p;
    }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        NRStepE eb = e;
        eb.xnew = my_xnew(my_trial());
        w.sh = eb.fetch1<ST>(o);
        w.active = 0u;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (ld_stat(ist[b]) == 0) w.active |= 1u << b;
        return w;
    }
    __device__ __forceinline__ void apply(int64_t i, uint32_t o, const double (&acc)[NB], bool valid, bool, bool lead, const Uni&,
                                          const Row& w, double (&v)[NB], double (&red)[NB]) const {
        const int h = (int)((threadIdx.x >> 2) & 3u), tr = h % NB;
        // this lane group's first product, by a tree of selects over OPAQUE copies: written as `acc[tr]` -- or as any chain or tree of selects
        // on the array's elements, which the compiler folds back into one indexed load -- the array goes to scratch memory, and the scratch
        // load, issued behind the next tile's row-input loads, waits for all of them (loads return in order): an exposed memory round trip
        // per tile step (FINDINGS.md 12.7)
        static_assert(NB == 2 || NB == 4, "instantiated batch widths");
        double acc_mine;
        if constexpr (NB == 2) {
            const double a0 = opaque_f64(acc[0]), a1 = opaque_f64(acc[1]);
            acc_mine = (tr & 1) ? a1 : a0;
        } else {
            const double a0 = opaque_f64(acc[0]), a1 = opaque_f64(acc[1]), a2 = opaque_f64(acc[2]), a3 = opaque_f64(acc[3]);
            const double lo = (tr & 1) ? a1 : a0, hi = (tr & 1) ? a3 : a2;
            acc_mine = (tr & 2) ? hi : lo;
        }
        double mine = 0.0, ball = 0.0;
        if ((w.active >> tr) & 1u) {                   // a finished trial keeps its iterate and contributes nothing
            NRStepE eb = e;
            eb.xnew = my_xnew(tr);
            // (h < NB: one storing lane per row and trial; in the wide form all four waves hold the row and compute its update -- wave 0, `lead`,
            // stores and counts)
            mine = eb.apply1<ST>(i, o, acc_mine, valid, h < NB && lead, w.sh, ball);
        }
        const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            v[b] = __shfl(mine, (lane & ~12) | (b << 2));                        // trial b's value from lane group H = b
            red[b] += (tr == b) ? ball : 0.0;
        }
    }
};

// ---- lfpsqp_elementwise: c(x) = A' phi(x) + qw * sum_{i<n_x} x_i^2 - b --------------------------------------------------------------
// c! as the SAME launch shape as the two-stream Newton step (gemv_nt_kernel with an empty first product): the raw products and the
// quadratic partial are then summed in the same order in both, so the cval a retraction returns is bit for bit c!(xnew)
// (test/test_retractions.jl:97 asserts exactly that of the reference).
struct EwEvalE {
    const double* x;
    const double* kind;              // nullptr: phi = identity
    double* phi_out;                 // optional: phi(x) stored (the sparse product is a separate launch)
    int64_t n_x, slack_row;
    int has_ball;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double ball(int64_t i, double xi) const {
        return !has_ball ? 0.0 : (i < n_x ? xi * xi : (i == slack_row ? -xi : 0.0));
    }
    __device__ __forceinline__ double2 apply(int64_t i, double2, bool v0, bool v1, double* red) const {
        double2 a = ld2(x + i);
        double b = 0.0;
        if (v0) b += ball(i, a.x);
        if (v1) b += ball(i + 1, a.y);
        red[0] += b;
        if (kind) {
            const double2 kk = ld2(kind + i);
            a.x = ew_phi(kk.x, a.x);
            a.y = ew_phi(kk.y, a.y);
        }
        if (phi_out) {
            if (v1) st2(phi_out + i, a);
            else if (v0) phi_out[i] = a.x;
        }
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};
struct EwEvalVecF {   // the same as a plain row pass (sparse A: phi(x) -> work, quadratic partial)
    EwEvalE e;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const { (void)e.apply(i, make_double2(0.0, 0.0), v0, v1, red); }
};
struct EwDerivF {     // out = phi'(x)  (row scales of the sparse / streamed constraint gradients), uout = 2 x [i < n_x] (their rank-one term); either may be null
    const double *x, *kind;
    double *out, *uout;
    int64_t n_x;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 a = ld2(x + i);
        if (out) {
            const double2 kk = kind ? ld2(kind + i) : make_double2(0.0, 0.0);
            const double2 d = make_double2(ew_phi1(kk.x, a.x), ew_phi1(kk.y, a.y));
            if (v1) st2(out + i, d);
            else if (v0) out[i] = d.x;
        }
        if (uout) {
            const double2 d = make_double2(i < n_x ? 2.0 * a.x : 0.0, i + 1 < n_x ? 2.0 * a.y : 0.0);
            if (v1) st2(uout + i, d);
            else if (v0) uout[i] = d.x;
        }
    }
};
// Jct[:, j] = phi'(x) .* A[:, j] + 2 qw_j x [i < n_x]   (jac! of the dense class: one read of A, one write of Jct)
constexpr int kEwJacCols = 16;
__global__ __launch_bounds__(kThreads) void ew_jac_kernel(const double* __restrict__ A, int64_t lda, double* __restrict__ J, int64_t ldj, int64_t n, int m,
                                                          const double* __restrict__ x, const double* __restrict__ kind, const double* __restrict__ qw,
                                                          int64_t n_x) {
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2;
    if (i >= n) return;
    const double2 a = ld2(x + i);
    const double2 kk = kind ? ld2(kind + i) : make_double2(0.0, 0.0);
    const double2 d = make_double2(ew_phi1(kk.x, a.x), ew_phi1(kk.y, a.y));
    const double2 xq = make_double2(i < n_x ? 2.0 * a.x : 0.0, i + 1 < n_x ? 2.0 * a.y : 0.0);
    const bool v1 = i + 1 < n;
    const int j0 = blockIdx.y * kEwJacCols, j1 = (j0 + kEwJacCols < m) ? j0 + kEwJacCols : m;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const double2 c = ld2(A + (int64_t)j * lda + i);
        const double w = qw ? qw[j] : 0.0;
        const double2 o = make_double2(fma(d.x, c.x, w * xq.x), fma(d.y, c.y, w * xq.y));
        if (v1) st2(J + (int64_t)j * ldj + i, o);
        else J[(int64_t)j * ldj + i] = o.x;
    }
}
struct EwHessE {      // hx[i] += phi''(x_i) * (A lam)_i + cq [i < n_x]       (GEMV-N consumer)
    const double *x, *kind;
    double* hx;
    double cq;
    int64_t n_x;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, double2 acc, bool v0, bool v1, double*) const {
        const double2 a = ld2(x + i);
        const double2 kk = kind ? ld2(kind + i) : make_double2(0.0, 0.0);
        double2 h = ld2(hx + i);
        h.x += ew_phi2(kk.x, a.x) * acc.x + (i < n_x ? cq : 0.0);
        h.y += ew_phi2(kk.y, a.y) * acc.y + (i + 1 < n_x ? cq : 0.0);
        if (v1) st2(hx + i, h);
        else if (v0) hx[i] = h.x;
    }
};
struct EwHessVecF {   // the same with (A lam) from the ELL entries, or without a product at all (R.t == nullptr: phi'' = 0 everywhere)
    EwHessE e;
    EllRows R;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        e.apply(i, R.t ? R.acc(i) : make_double2(0.0, 0.0), v0, v1, red);
    }
};

// the m_lin weights of the quadratic term on the device (a buffer of its own: d_m / small are carved up by the callers)
static int stage_qw(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const double** out) {
    *out = nullptr;
    if (!cons->ew || !cons->ew->qw || cons->m_lin == 0) return 0;
    if ((size_t)cons->m_lin > ctx->qw_cap) {
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_qw) LF_HIP(ctx, hipFree(ctx->d_qw));
        ctx->d_qw = nullptr; ctx->qw_cap = 0;
        LF_HIP(ctx, hipMalloc((void**)&ctx->d_qw, sizeof(double) * (size_t)round_up(cons->m_lin, 64)));
        ctx->qw_cap = (size_t)round_up(cons->m_lin, 64);
    }
    LF_HIP(ctx, hipMemcpyAsync(ctx->d_qw, cons->ew->qw, sizeof(double) * (size_t)cons->m_lin, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));            // caller-owned pageable source
    *out = ctx->d_qw;
    return 0;
}

// The stacked step, the plain step and the c! evaluation are different instantiations of the one-pass kernel; with the same number of
// workgroups (two per CU: what each of them keeps resident at least) they cut the rows alike and sum their partials in the same order,
// so the cval a retraction returns is c!(xnew) bit for bit with bounds as well.
constexpr int kEwWgPerCu = 2;
// nonlinear class with a dense A of a shape the one-pass kernel covers: c! and the Newton step stream A through onepass_kernel
static bool ew_onepass_ok(const lfpsqp_ctx* ctx, const lfpsqp_constraints* cons) {
    const lfpsqp_elementwise* ew = cons->ew;
    return ew && !ew->Asp && ew->A && !cons->has_ball && cons->m_lin >= 4 && onepass_cw(ctx, (int)cons->m_lin, ew->A->ld, cons->Jct->n) != 0;
}

// raw[0:m_lin) = the constraint products (J x, or A' phi(x)), raw[m_lin] = sum_{i<n_x} x_i^2 - x[slack_row] (when the class has a ball or a
// common quadratic term) -- device buffer, all-reduced, stream-ordered.  `x` has >= rows(Jct) entries.
static int cons_raw(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const double* x, double* raw) {
    const lfpsqp_mat* J = cons->Jct;
    const int ml = (int)cons->m_lin;
    const int64_t N = J->n;
    const lfpsqp_elementwise* ew = cons->ew;
    const bool quad = cons->has_ball || (ew && ew->qw);
    if (ew) {
        const EwEvalE ee{x, ew->kind ? ew->kind->p : nullptr, ew->Asp ? ew->work->p : nullptr, cons->n_x, cons->slack_row, quad ? 1 : 0};
        if (ew->Asp) {
            LF_TRY((run_vec<EwEvalVecF, 1, NoPost>(ctx, N, EwEvalVecF{ee}, 0u, raw + ml, NoPost())));
            return spmv_t(ctx, ew->Asp, ew->work->p, raw);
        }
        if (ew_onepass_ok(ctx, cons)) {
            // c! in the launch shape of the one-stream Newton step (first-product coefficients zero, nothing updated or stored): the sums of a
            // step's second product and of this evaluation are then taken in the same order -- cval of a retraction == c!(xnew), bit for bit
            if (!ctx->d_zeros) {
                LF_HIP(ctx, hipMalloc((void**)&ctx->d_zeros, sizeof(double) * (kOnepassMaxCols + 8)));
                LF_HIP(ctx, hipMemsetAsync(ctx->d_zeros, 0, sizeof(double) * (kOnepassMaxCols + 8), ctx->stream));
            }
            NRStepE e0{const_cast<double*>(x), nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cons->n_x, cons->slack_row, quad ? 1 : 0,
                       ctx->istat};
            e0.kind = ew->kind ? ew->kind->p : nullptr;
            e0.ew_one = 1;
            e0.eval_only = 1;
            return run_onepass<NRStepRow<false>, 1, 1>(ctx, ew->A, ml, ml, N, ctx->d_zeros, NRStepRow<false>{e0}, raw, -1, 0, kEwWgPerCu);
        }
        return run_gemv_nt<EwEvalE, 1>(ctx, nullptr, 0, nullptr, ew->A, ml, N, ee, raw);
    }
    if (ml > 0 && cons->Jsp) LF_TRY(spmv_t(ctx, cons->Jsp, x, raw));                       // c! streams the nonzeros
    else if (ml > 0) LF_TRY(run_gemv_t(ctx, J, ml, N, PlainVec{x}, raw));
    if (cons->has_ball) LF_TRY((run_vec<BallF, 1, NoPost>(ctx, N, BallF{x, cons->n_x, cons->slack_row}, 0u, raw + ml, NoPost())));
    return 0;
}

int cons_eval(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, double* cval) {
    const int ml = (int)cons->m_lin;
    const int mt = ml + (cons->has_ball ? 1 : 0);
    const double* qw = cons->ew ? cons->ew->qw : nullptr;
    LF_TRY(ensure_mvec(ctx, (size_t)ml + 9));
    LF_TRY(cons_raw(ctx, cons, x->p, ctx->d_m));
    if (ml + 1 > 0) {
        LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->d_m, sizeof(double) * (ml + 1), hipMemcpyDeviceToHost, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    (void)mt;
    for (int j = 0; j < ml; ++j) cval[j] = qw ? fma(qw[j], ctx->h_m[ml], ctx->h_m[j]) - cons->b[j] : (ctx->h_m[j] - cons->b[j]);
    if (cons->has_ball) cval[ml] = ctx->h_m[ml] - cons->R2;
    return 0;
}

template <bool ST, int NB>
int nr_batch_step(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, int wm, int ml, int64_t N, const double* dwdelta, int wstride,
                         const NRStepE& ep0, lfpsqp_vec* const* xnew, double* draw) {
    NRStepBatchRow<ST, NB> ep;
    ep.e = ep0;
    for (int b = 0; b < NB; ++b) { ep.xnew[b] = xnew[b]->p; ep.ist[b] = ctx->istat + I_NRB + 4 * b; }
    ep.all = ctx->istat + I_NRB_ALL;
    // EXACT batch: the rows are cut into the spans of the SINGLE-trial step's launch (lfpsqp_retract_nr: NRStepRow<ST> over the same matrix and
    // shape) and the second stage is shaped as for its ml + 1 columns, so every sum of a trial -- first product, row update, second product,
    // partial rows, second stage -- is formed from the same operands in the same order as when the trial is retracted alone: batching is
    // invisible in the bits (tests: test_exact_batch_is_bit_identical).
    int g1 = 0;
    LF_TRY((run_onepass<NRStepRow<ST>, 1, 1>(ctx, cons->Jct, wm, ml, N, dwdelta, NRStepRow<ST>{ep0}, draw, -1, 0, 0, false, 0, &g1)));
    return run_onepass<NRStepBatchRow<ST, NB>, NB, NB, NB>(ctx, cons->Jct, wm, ml, N, dwdelta, ep, draw, 7, wstride, ctx->batch_wg_cap, false, g1, nullptr,
                                                           ml + 1);     // (profiling slot 7)
}

}  // namespace lfpsqp

#include "nrbatch.h"

namespace lfpsqp {

// shape covered by the matrix-core form of the batched step?  (first product over wm <= 132 columns, second over ml <= 128)
static bool nrb_mfma_shape(int wm, int ml) { return wm >= 4 && wm <= 132 && ml >= 1 && ml <= 128 && ml <= wm; }

template <bool ST, int CPL, int NBLK>
static int nrb_mfma_launch(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int wm, int ml, int64_t N, const double* dwdelta, int wstride, const NRBatchArgs& ep,
                           double* out) {
    ++ctx->launch_epoch;
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nrb_mfma_kernel<ST, CPL, NBLK>, kThreads, 0) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
    }
    const int64_t rounds = (N + kOnepassRound - 1) / kOnepassRound;
    const int nout = kNRBW * ml + kNRBW;
    const int part_ld = (int)round_up(nout, 32);
    if (rounds <= 0) {
        LF_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double) * nout, ctx->stream));
        return allreduce_dev(ctx, out, nout);
    }
    int64_t g = (int64_t)per_cu * (ctx->num_cu > 0 ? ctx->num_cu : 1);
    if (g > rounds) g = rounds;
    const int grid = (int)(g < 1 ? 1 : g);
    LF_TRY(ensure_part(ctx, (size_t)grid * kWaves * part_ld + reduce_scratch(part_ld)));
    prof_begin(ctx, 7);
    hipLaunchKernelGGL((nrb_mfma_kernel<ST, CPL, NBLK>), dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, M->p, M->ld, wm, ml, N, rounds, dwdelta,
                       wstride, ep, ctx->part, part_ld);
    prof_end(ctx, 7);
    LF_LAUNCH_CHECK(ctx);
    LF_TRY(launch_reduce(ctx, (int64_t)grid * kWaves, nout, part_ld, 0u, out, NoPost()));
    return allreduce_dev(ctx, out, nout);
}

// shape covered by the WIDE matrix-core form (nrb_mfma_wide_kernel: the four waves of a workgroup split the columns; up to 8 trials a pass)
static bool nrb_wide_shape(int wm, int ml) { return wm > 132 && wm <= 528 && ml >= 1 && ml <= wm; }

template <bool ST, int CPL>
static int nrb_wide_launch(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int wm, int ml, int64_t N, const double* dwdelta, int wstride, const NRBatchArgs& ep,
                           double* out) {
    ++ctx->launch_epoch;
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nrb_mfma_wide_kernel<ST, CPL>, kThreads, 0) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb;
    }
    const int64_t rounds = (N + 15) / 16;
    const int nout = kNRBWideTrials * ml + kNRBWideTrials;
    const int part_ld = (int)round_up(nout, 32);
    if (rounds <= 0) {
        LF_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double) * nout, ctx->stream));
        return allreduce_dev(ctx, out, nout);
    }
    int64_t g = (int64_t)per_cu * (ctx->num_cu > 0 ? ctx->num_cu : 1);
    if (g > rounds) g = rounds;
    const int grid = (int)(g < 1 ? 1 : g);
    LF_TRY(ensure_part(ctx, (size_t)grid * part_ld + reduce_scratch(part_ld)));
    prof_begin(ctx, 7);
    hipLaunchKernelGGL((nrb_mfma_wide_kernel<ST, CPL>), dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, M->p, M->ld, wm, ml, N, rounds, dwdelta,
                       wstride, ep, ctx->part, part_ld);
    prof_end(ctx, 7);
    LF_LAUNCH_CHECK(ctx);
    LF_TRY(launch_reduce(ctx, (int64_t)grid, nout, part_ld, 0u, out, NoPost()));
    return allreduce_dev(ctx, out, nout);
}

template <bool ST>
static int nrb_wide_step(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, int wm, int ml, int64_t N, const double* dwdelta, int wstride,
                         const NRStepE& ep0, lfpsqp_vec* const* xnew, int nb, double* draw) {
    NRBatchArgs ep;
    ep.e = ep0;
    for (int b = 0; b < kNRBW; ++b) ep.xnew[b] = xnew[b < nb ? b : 0]->p;
    ep.ist = ctx->istat + I_NRB;
    ep.all = ctx->istat + I_NRB_ALL;
    ep.nb = nb;
    const int cplw = ((wm + 3) / 4 + kWaves - 1) / kWaves;          // column groups per wave
    if (cplw <= 16) return nrb_wide_launch<ST, 16>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    if (cplw <= 24) return nrb_wide_launch<ST, 24>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    if (cplw <= 32) return nrb_wide_launch<ST, 32>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    return nrb_wide_launch<ST, 33>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
}

template <bool ST>
static int nrb_mfma_step(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, int wm, int ml, int64_t N, const double* dwdelta, int wstride,
                         const NRStepE& ep0, lfpsqp_vec* const* xnew, int nb, double* draw) {
    NRBatchArgs ep;
    ep.e = ep0;
    for (int b = 0; b < kNRBW; ++b) ep.xnew[b] = xnew[b < nb ? b : 0]->p;
    ep.ist = ctx->istat + I_NRB;
    ep.all = ctx->istat + I_NRB_ALL;
    ep.nb = nb;
    const int cpl = (wm + 3) / 4, nblk = (ml + 15) / 16;
    if (cpl <= 9 && nblk <= 3) return nrb_mfma_launch<ST, 9, 3>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    if (cpl <= 17 && nblk <= 5) return nrb_mfma_launch<ST, 17, 5>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    if (cpl <= 32) return nrb_mfma_launch<ST, 32, 8>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
    return nrb_mfma_launch<ST, 33, 8>(ctx, cons->Jct, wm, ml, N, dwdelta, wstride, ep, draw);
}

}  // namespace lfpsqp

using namespace lfpsqp;

static bool cons_ok(const lfpsqp_constraints* c) {
    if (!(c && c->Jct && c->m_lin >= 0 && c->m_lin + (c->has_ball ? 1 : 0) <= c->Jct->m && (c->m_lin == 0 || c->b) &&
          (!c->has_ball || (c->n_x >= 0 && c->n_x <= c->Jct->n && c->slack_row < c->Jct->n)) &&
          (!c->Jsp || (c->Jsp->n == c->Jct->n && c->Jsp->m == c->m_lin))))
        return false;
    // a row-scaled view as Jct (constant or streamed gradients): no ball column (jac! would write it into the borrowed storage, and it is not
    // scaled with the rows) and no sparse twin
    if (c->Jct->view && (c->has_ball || c->Jsp)) return false;
    const lfpsqp_elementwise* e = c->ew;
    if (!e) return true;
    // STREAMED gradients: Jct is a row-scaled view of A itself (lfpsqp_mat_rowscaled_view) -- Jct(x) = diag(phi'(x)) A never exists in memory, jac!
    // rewrites the view's scale vector.  No rank-one / ball column then (they are not row scalings of A).
    // The view carries what the class has: row scales iff kind, the rank-one term u qw' (u = 2 x on the first n_x rows, refreshed by jac!; the view's w
    // must hold qw) iff qw.
    const bool streamed = e->A && e->A->p == c->Jct->p;
    if (streamed && !(c->Jct->view && plain_mat(e->A) && !e->Asp && !c->has_ball && !c->Jsp && (c->Jct->rs != nullptr) == (e->kind != nullptr) &&
                      (c->Jct->ru != nullptr) == (e->qw != nullptr)))
        return false;
    if (!streamed && !plain_mat(c->Jct)) return false;
    return c->m_lin >= 1 && (e->Asp || e->A) && (!e->A || (e->A->n == c->Jct->n && e->A->m >= c->m_lin)) &&
           (!e->kind || e->kind->n >= c->Jct->n) &&
           (!e->qw || (!c->has_ball && !e->Asp && c->n_x >= 0 && c->n_x <= c->Jct->n)) &&
           (!e->Asp || (c->Jsp && c->Jsp != e->Asp && e->Asp->n == c->Jct->n && e->Asp->m == c->m_lin && c->Jsp->ell_col == e->Asp->ell_col &&
                        c->Jsp->csc_row == e->Asp->csc_row && e->work && e->work->n >= c->Jct->n)) &&
           (e->Asp || !c->Jsp);
}

extern "C" {

int lfpsqp_constraints_eval(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, double* cval) {
    LF_RANGE("lfpsqp_constraints_eval");
    LF_ARG(ctx, ctx && cons_ok(cons) && x && cval && x->n >= cons->Jct->n);
    return cons_eval(ctx, cons, x, cval);
}

int lfpsqp_constraints_jac(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, lfpsqp_mat* Jct, double* cval) {
    LF_RANGE("lfpsqp_constraints_jac");
    LF_ARG(ctx, ctx && cons_ok(cons) && x && Jct && x->n >= cons->Jct->n && Jct->p == cons->Jct->p && Jct->rs == cons->Jct->rs && Jct->ru == cons->Jct->ru);
    if (const lfpsqp_elementwise* ew = cons->ew) {             // Jct[:, :m_lin] = diag(phi'(x)) A + 2 x qw'
        const int64_t N = Jct->n;
        const int ml = (int)cons->m_lin;
        if (Jct->view) {                                          // streamed gradients: jac! is the n-vector phi'(x) (and 2 x for the quadratic term), the matrix stays A
            LF_TRY((run_vec<EwDerivF, 0, NoPost>(ctx, N, EwDerivF{x->p, ew->kind ? ew->kind->p : nullptr, const_cast<double*>(Jct->rs), const_cast<double*>(Jct->ru), cons->n_x},
                                                 0u, nullptr, NoPost())));
        } else if (ew->Asp) {
            LF_TRY((run_vec<EwDerivF, 0, NoPost>(ctx, N, EwDerivF{x->p, ew->kind ? ew->kind->p : nullptr, ew->work->p, nullptr, 0}, 0u, nullptr, NoPost())));
            LF_TRY(lfpsqp_spmat_rowscale(ctx, const_cast<lfpsqp_spmat*>(cons->Jsp), ew->Asp, ew->work));
            LF_TRY(lfpsqp_spmat_to_dense(ctx, cons->Jsp, Jct));
        } else if (N > 0) {
            const double* dqw = nullptr;
            LF_TRY(stage_qw(ctx, cons, &dqw));
            hipLaunchKernelGGL(ew_jac_kernel, dim3((unsigned)((N + 2 * kThreads - 1) / (2 * kThreads)), (unsigned)((ml + kEwJacCols - 1) / kEwJacCols)),
                               dim3(kThreads), 0, ctx->stream, ew->A->p, ew->A->ld, Jct->p, Jct->ld, N, ml, x->p, ew->kind ? ew->kind->p : nullptr, dqw,
                               cons->n_x);
            LF_LAUNCH_CHECK(ctx);
        }
    }
    if (cons->has_ball)
        LF_TRY((run_vec<BallColF, 0, NoPost>(ctx, Jct->n, BallColF{x->p, Jct->p + cons->m_lin * Jct->ld, cons->n_x, cons->slack_row}, 0u,
                                             nullptr, NoPost())));
    if (!cval) return 0;                                      // gradients only: the caller holds c(x) already (an accepted retraction returns it)
    return cons_eval(ctx, cons, x, cval);
}

int lfpsqp_constraints_hess_diag(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, const lfpsqp_vec* x, const double* lam, lfpsqp_vec* hx) {
    LF_RANGE("lfpsqp_constraints_hess_diag");
    LF_ARG(ctx, ctx && cons_ok(cons) && x && lam && hx && x->n >= cons->Jct->n && hx->n >= cons->Jct->n && hx->p != x->p);
    const int64_t N = cons->Jct->n;
    const int ml = (int)cons->m_lin;
    const lfpsqp_elementwise* ew = cons->ew;
    double cq = cons->has_ball ? 2.0 * lam[ml] : 0.0;
    if (ew && ew->qw)
        for (int j = 0; j < ml; ++j) cq += 2.0 * ew->qw[j] * lam[j];
    const double* kind = (ew && ew->kind) ? ew->kind->p : nullptr;
    const EwHessE he{x->p, kind, hx->p, cq, cons->n_x};
    if (!kind) {                                               // phi'' = 0: only the constant of the quadratic terms
        if (cq == 0.0) return 0;
        return run_vec<EwHessVecF, 0, NoPost>(ctx, N, EwHessVecF{he, EllRows{nullptr, nullptr, 0, 0, nullptr}}, 0u, nullptr, NoPost());
    }
    LF_TRY(ensure_mvec(ctx, (size_t)ml + 8));
    for (int j = 0; j < ml; ++j) ctx->h_m[j] = lam[j];
    LF_HIP(ctx, hipMemcpyAsync(ctx->d_m, ctx->h_m, sizeof(double) * ml, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));            // h_m is the context's SHARED pinned staging block: the copy must have read it
                                                               // before this call returns and another call writes it (as stage_qw does)
    if (ew->Asp) return run_vec<EwHessVecF, 0, NoPost>(ctx, N, EwHessVecF{he, ell_rows(ew->Asp, ctx->d_m)}, 0u, nullptr, NoPost());
    return run_gemv_n<EwHessE, 0, NoPost>(ctx, ew->A, ml, N, ctx->d_m, he, nullptr, NoPost());
}

int lfpsqp_retract_nr(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m64,
                      const lfpsqp_constraints* cons, lfpsqp_cfun cfun, void* cuser, const lfpsqp_ineq_data* idata,
                      const lfpsqp_vec* xtilde, const lfpsqp_vec* x, lfpsqp_vec* xnew, double tol, int64_t maxiter, double* cval,
                      int* flag, int64_t* iters) {
    LF_RANGE("lfpsqp_retract_nr");
    LF_ARG(ctx, ctx && U && Sigma && Vt && xtilde && x && xnew && cval && flag && iters && m64 >= 1 && U->ncols == m64);
    LF_ARG(ctx, cfun || cons_ok(cons));
    LF_ARG(ctx, xtilde->n == x->n && xnew->n == x->n && xnew->p != x->p && xnew->p != xtilde->p);
    const int m = (int)m64;
    const bool ineq = idata != nullptr;
    LF_ARG(ctx, ineq == (U->Dx != nullptr));
    LF_TRY(ensure_mvec(ctx, (size_t)2 * m + 8));

    if (!cfun && m <= kNRMaxM) {
        // ---- device-resident loop: no host round trip per iteration ------------------------------------------
        const int ml = (int)cons->m_lin;
        const int64_t N = cons->Jct->n;
        const size_t mm = (size_t)m * m;
        // one-stream step: the caller vouches that U->Z == U->A * U->W and A is the matrix c! streams anyway
        const lfpsqp_elementwise* ew = cons->ew;                  // nonlinear class: c! streams the constant A with phi(xnew), not Jct
        const bool quad = cons->has_ball || (ew && ew->qw);
        const int wm = (U->A && U->W && U->A->p == cons->Jct->p && U->A->m <= kNRMaxM) ? (int)U->A->m : 0;
        const int cwd = (wm && !ew) ? onepass_cw(ctx, wm, cons->Jct->ld, N) : 0;
        // ... and with a sparse twin of the linear block (all but <= 4 of A's columns) the step runs on the nonzeros alone
        const lfpsqp_spmat* Ssp = (wm && ml > 0 && cons->Jsp && cons->Jsp->n == N && cons->Jsp->m == ml && wm >= ml && wm - ml <= 4) ? cons->Jsp : nullptr;
        // (non-zero: the generator W and W*ddelta are kept on the device) -- also when only the two-stream kernel applies (nonlinear class, shapes
        // without a one-pass kernel): it then streams Jct with W*ddelta instead of Z with ddelta, so a basis in factored form (U->Z == NULL) works
        const int cw = (Ssp || wm) ? 1 : cwd;
        LF_ARG(ctx, U->Z || wm);
        const size_t wsz = cw ? (((size_t)wm * m + (size_t)wm + 3) & ~(size_t)1) : 0;     // W, W*delta and one scalar (qw . W*delta) behind it
        LF_TRY(ensure_small(ctx, 2 * mm + wsz + 8 * (size_t)m + 256));
        double* dD = ctx->small;
        double* dVt = dD + mm;
        double* dW = dVt + mm;
        double* dwdelta = dW + (size_t)(cw ? wm : 0) * m;
        double* dSig = dW + wsz;
        double* db = dSig + m;
        double* dcval = db + m;
        double* ddelta = dcval + m;
        double* draw = ddelta + m + (m & 1);          // keep the reduce output 16-byte aligned
        if (cw) LF_HIP(ctx, hipMemcpyAsync(dW, U->W, sizeof(double) * (size_t)wm * m, hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipMemcpyAsync(dVt, Vt, sizeof(double) * mm, hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipMemcpyAsync(dSig, Sigma, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
        if (ml > 0) LF_HIP(ctx, hipMemcpyAsync(db, cons->b, sizeof(double) * ml, hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));                 // caller-owned pageable sources
        volatile int64_t* hstat = ctx->h_istat;
        for (int k = 0; k < kNRRing; ++k) hstat[kNRRingOff + k] = 0;
        hstat[I_NR_STATUS] = 0;
        NRSmall sm{dD, dVt, dSig, db, dcval, ddelta, draw, draw + ml, cw ? dW : nullptr, dwdelta, ctx->istat + I_NR_STATUS,
                   ctx->h_istat, m, ml, cons->has_ball ? 1 : 0, cw ? wm : 0, cons->R2, tol, maxiter};
        LF_TRY(stage_qw(ctx, cons, &sm.qw));
        LF_TRY(lfpsqp_vec_copy(ctx, xnew, xtilde));                                             // :116
        if (ineq) LF_TRY(lfpsqp_y_retract(ctx, xnew, x, idata));                                // :118-120
        LF_TRY(cons_raw(ctx, cons, xnew->p, draw));                                             // c!(cval, xnew), raw products
        hipLaunchKernelGGL(nr_small_kernel, dim3(1), dim3(kNRThreads), 0, ctx->stream, sm, 1);
        LF_LAUNCH_CHECK(ctx);
        LF_HIP(ctx, hipEventRecord(ctx->ev_slot[0], ctx->stream));
        NRStepE ep{xnew->p, x->p, ineq ? lfpsqp_half_stride(N) : 0, ineq ? U->sx->p : nullptr, ineq ? U->sy->p : nullptr,
                   ineq ? idata->q->p : nullptr, ineq ? idata->r->p : nullptr, ineq ? idata->s->p : nullptr,
                   ineq ? idata->t->p : nullptr, cons->n_x, cons->slack_row, quad ? 1 : 0, ctx->istat};
        if (ew) {
            ep.kind = ew->kind ? ew->kind->p : nullptr;
            ep.phi_out = ew->Asp ? ew->work->p : nullptr;      // (identity phi with a sparse A: the product reads xnew itself)
            if (ew->Asp && !ep.kind) ep.phi_out = nullptr;
        }
        const lfpsqp_mat* cmat = ew ? ew->A : cons->Jct;          // the matrix of the c! product
        // first product of a two-stream step: U delta = Z delta, or Jct (W delta) when the generator is known
        const lfpsqp_mat* s1 = wm ? cons->Jct : U->Z;
        const int n1 = wm ? wm : m;
        const double* t1 = wm ? dwdelta : ddelta;
        const lfpsqp_spmat* csp = ew ? ew->Asp : Ssp;
        const bool ewo = ew_onepass_ok(ctx, cons);
        int64_t it = 0;
        bool done = false;
        while (!done && it < maxiter) {
            // step `it`: xnew += U delta, y_retract!, raw c! products (one launch) ; Broyden + test + next delta (one workgroup)
            if (Ssp) {
                const NRSparseStepF sf{ep, ell_rows(Ssp, dwdelta), cons->Jct->p + (int64_t)ml * cons->Jct->ld, cons->Jct->ld, dwdelta + ml, wm - ml};
                LF_TRY((run_vec<NRSparseStepF, 1, NoPost>(ctx, N, sf, 0u, draw + ml, NoPost())));
                LF_TRY(spmv_t(ctx, csp, ep.phi_out ? ep.phi_out : xnew->p, draw));
            } else if (ewo && wm == ml) {                          // nonlinear class, dense A, generator known: ONE pass over A per step
                NRStepE e1 = ep;
                e1.ew_one = 1;
                e1.sq = ew->qw ? dwdelta + wm : nullptr;
                if (ineq) LF_TRY((run_onepass<NRStepRow<true>, 1, 1>(ctx, ew->A, ml, ml, N, dwdelta, NRStepRow<true>{e1}, draw, -1, 0, kEwWgPerCu)));
                else LF_TRY((run_onepass<NRStepRow<false>, 1, 1>(ctx, ew->A, ml, ml, N, dwdelta, NRStepRow<false>{e1}, draw, -1, 0, kEwWgPerCu)));
            } else if (ewo) {                                      // ... without the generator: the update over Z, then c! in its own (one-pass) launch
                NRStepE e1 = ep;
                e1.kind = nullptr;                                 // (the update alone: nothing is handed to a second product)
                LF_TRY((run_gemv_nt<NRStepE, 0>(ctx, s1, n1, t1, nullptr, 0, N, e1, draw + ml)));
                LF_TRY(cons_raw(ctx, cons, xnew->p, draw));
            } else if (cwd && ineq) LF_TRY((run_onepass<NRStepRow<true>, 1, 1>(ctx, cons->Jct, wm, ml, N, dwdelta, NRStepRow<true>{ep}, draw, 4)));      // (profiling slot 4)
            else if (cwd) LF_TRY((run_onepass<NRStepRow<false>, 1, 1>(ctx, cons->Jct, wm, ml, N, dwdelta, NRStepRow<false>{ep}, draw, 4)));
            else if (ew && ew->Asp) {                             // sparse A without the generator hint: dense step over Z, sparse c!
                LF_TRY((run_gemv_nt<NRStepE, 0>(ctx, s1, n1, t1, nullptr, 0, N, ep, draw + ml)));
                if (quad) LF_TRY((run_vec<BallF, 1, NoPost>(ctx, N, BallF{xnew->p, cons->n_x, cons->slack_row}, 0u, draw + ml, NoPost())));   // (summed as c! sums it)
                LF_TRY(spmv_t(ctx, csp, ep.phi_out ? ep.phi_out : xnew->p, draw));
            } else if (quad || ew) LF_TRY((run_gemv_nt<NRStepE, 1>(ctx, s1, n1, t1, cmat, ml, N, ep, draw)));
            else LF_TRY((run_gemv_nt<NRStepE, 0>(ctx, s1, n1, t1, cmat, ml, N, ep, draw)));
            hipLaunchKernelGGL(nr_small_kernel, dim3(1), dim3(kNRThreads), 0, ctx->stream, sm, 0);
            LF_LAUNCH_CHECK(ctx);
            LF_HIP(ctx, hipEventRecord(ctx->ev_slot[(it + 1) & 3], ctx->stream));
            // rank-deterministic stop: status after `it` completed steps (published by the previous kernel generation)
            LF_HIP(ctx, hipEventSynchronize(ctx->ev_slot[it & 3]));
            if (hstat[kNRRingOff + ((it + 1) % kNRRing)] != 0) done = true;
            ++it;
        }
        LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, dcval, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < m; ++k) cval[k] = ctx->h_m[k];
        *flag = (int)hstat[I_NR_FLAG];
        *iters = hstat[I_NR_ITER];
        return 0;
    }

    auto eval_c = [&](double* out) -> int {
        if (cfun) {
            int rc = cfun(cuser, xnew, out);
            if (rc != 0) return set_err(ctx, LFPSQP_ERR_ARG, "user c! callback returned %d", rc);
            return 0;
        }
        return cons_eval(ctx, cons, xnew, out);
    };

    LF_TRY(lfpsqp_vec_copy(ctx, xnew, xtilde));                       // :116
    if (ineq) LF_TRY(lfpsqp_y_retract(ctx, xnew, x, idata));          // :118-120
    LF_TRY(eval_c(cval));
    std::vector<double> D((size_t)m * m), tmp(m), tmp2(m), dc(m);
    for (int j = 0; j < m; ++j)                                       // :126-130  D[k,j] = Vt[k,j] / Sigma[k]
        for (int k = 0; k < m; ++k) D[(size_t)j * m + k] = Vt[(size_t)j * m + k] / Sigma[k];
    // staging vector for delta on the device (offset past the slots cons_eval uses)
    lfpsqp_vec tv;
    tv.p = ctx->d_m + round_up(m + 8, 2);
    tv.n = m;
    tv.cap = m;
    double* h_delta = ctx->h_m + round_up(m + 8, 2);

    int64_t i = 0;
    while (i < maxiter) {
        double cmax = 0.0;
        for (int k = 0; k < m; ++k) {                                 // NaN-propagating, like norm(cval, Inf)
            const double a = fabs(cval[k]);
            cmax = (cmax != cmax) ? cmax : ((a != a) ? a : fmax(cmax, a));
        }
        if (cmax < tol) break;                                        // :135  (NaN compares false, like Julia)
        for (int k = 0; k < m; ++k) {                                 // :140  tmp = -D cval
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += D[(size_t)j * m + k] * cval[j];
            tmp[k] = -s;
            h_delta[k] = -s;
        }
        LF_HIP(ctx, hipMemcpyAsync(tv.p, h_delta, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
        if (!cfun && !cons->ew && U->Z) {
            // fused step: xnew += U tmp (:141), y_retract! (:145), and the c! products (:146/148) in one launch
            const int ml = (int)cons->m_lin;
            const int64_t N = cons->Jct->n;
            NRStepE ep{xnew->p, x->p, ineq ? lfpsqp_half_stride(N) : 0, ineq ? U->sx->p : nullptr, ineq ? U->sy->p : nullptr,
                       ineq ? idata->q->p : nullptr, ineq ? idata->r->p : nullptr, ineq ? idata->s->p : nullptr,
                       ineq ? idata->t->p : nullptr, cons->n_x, cons->slack_row, cons->has_ball ? 1 : 0, ctx->istat};
            ctx->h_istat[I_NR_STATUS] = 0;
            LF_HIP(ctx, hipMemsetAsync(ctx->istat + I_NR_STATUS, 0, sizeof(int64_t), ctx->stream));
            LF_TRY((run_gemv_nt<NRStepE, 1>(ctx, U->Z, m, tv.p, cons->Jct, ml, N, ep, ctx->d_m)));
            LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, ctx->d_m, sizeof(double) * (ml + 1), hipMemcpyDeviceToHost, ctx->stream));
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int j = 0; j < ml; ++j) tmp2[j] = ctx->h_m[j] - cons->b[j];
            if (cons->has_ball) tmp2[ml] = ctx->h_m[ml] - cons->R2;
        } else {
            LF_TRY(lfpsqp_q_gemv_n(ctx, U, 1.0, nullptr, &tv, 1.0, xnew));   // :141  xnew += U tmp
            if (ineq) LF_TRY(lfpsqp_y_retract(ctx, xnew, x, idata));      // :144-146
            LF_TRY(eval_c(tmp2.data()));
        }
        for (int k = 0; k < m; ++k) { dc[k] = tmp2[k] - cval[k]; cval[k] = tmp2[k]; }   // :152-153
        for (int k = 0; k < m; ++k) {                                 // :156  tmp2 = D' tmp
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += D[(size_t)k * m + j] * tmp[j];
            tmp2[k] = s;
        }
        for (int k = 0; k < m; ++k) {                                 // :157  tmp = tmp - D dc
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += D[(size_t)j * m + k] * dc[j];
            tmp[k] -= s;
        }
        double den = 0.0;
        for (int k = 0; k < m; ++k) den += tmp2[k] * dc[k];
        const double alpha = 1.0 / den;                               // :159
        for (int j = 0; j < m; ++j)                                   // :160  D += alpha tmp tmp2'
            for (int k = 0; k < m; ++k) D[(size_t)j * m + k] += alpha * tmp[k] * tmp2[j];
        ++i;
    }
    *flag = (i == maxiter) ? 1 : 0;                                   // :171-174
    *iters = i;
    return 0;
}


// How many trial points lfpsqp_retract_nr_batch takes per pass for this basis and these constraints: 16 (the step on the matrix cores: up
// to 132 generator columns / 128 linear constraints), 4 (the VALU form: up to 1024 columns, the wide form of the one-pass kernel from 257 on) or 0 (cannot batch: no generator, sparse
// constraint gradients, the nonlinear class, a shape without the one-stream step).
int lfpsqp_retract_nr_batch_width(const lfpsqp_ctx* ctx, const lfpsqp_basis* U, const lfpsqp_constraints* cons, int* width) {
    if (!ctx || !U || !cons || !width || !cons->Jct) return LFPSQP_ERR_ARG;
    *width = 0;
    if (cons->Jsp || cons->ew || !plain_mat(cons->Jct)) return 0;
    const int ml = (int)cons->m_lin;
    const int wm = (U->A && U->W && U->A->p == cons->Jct->p && U->A->m <= kOnepassMaxCols) ? (int)U->A->m : 0;
    if (!wm || U->ncols > kNRMaxM || !onepass_cw(ctx, wm, cons->Jct->ld, cons->Jct->n)) return 0;
    *width = (nrb_mfma_shape(wm, ml) && ctx->tune_nrb_mfma >= 0) ? kNRBatchMax
             : ((nrb_wide_shape(wm, ml) && ctx->tune_nrb_mfma >= 0) ? kNRBWideTrials : 4);
    return 0;
}

// Batched form of lfpsqp_retract_nr: nb (2..16) independent retractions of the trial points xtilde[b] (all from the same x,
// the same factors and constraints) advance together, one pass over Jct per Newton step for all of them.  Each trial
// has its own Broyden state, convergence test and iteration count and stops on its own (a finished trial keeps its
// iterate); the call returns when all have finished.  Requires the one-stream step (basis generator known,
// device-resident constraints, 4 <= columns <= 256); otherwise LFPSQP_ERR_UNSUPPORTED and the caller retracts one by one.
int lfpsqp_retract_nr_batch(lfpsqp_ctx* ctx, const lfpsqp_basis* U, const double* Sigma, const double* Vt, int64_t m64,
                            const lfpsqp_constraints* cons, const lfpsqp_ineq_data* idata, int nb, const lfpsqp_vec* const* xtilde,
                            const lfpsqp_vec* x, lfpsqp_vec* const* xnew, double tol, int64_t maxiter, double* cval, int* flags,
                            int64_t* iters) {
    LF_RANGE("lfpsqp_retract_nr_batch");
    LF_ARG(ctx, ctx && U && Sigma && Vt && xtilde && x && xnew && cval && flags && iters && m64 >= 1 && U->ncols == m64);
    LF_ARG(ctx, cons_ok(cons) && nb >= 2 && nb <= kNRBatchMax);
    {
        int width = 0;
        LF_TRY(lfpsqp_retract_nr_batch_width(ctx, U, cons, &width));
        if (width == 0) return LFPSQP_ERR_UNSUPPORTED;            // (sparse constraint gradients, the nonlinear class, shapes without the one-stream step)
        if (nb > width) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "batched Newton retraction: %d trials, this shape takes at most %d per pass", nb, width);
    }
    const int m = (int)m64;
    const bool ineq = idata != nullptr;
    LF_ARG(ctx, ineq == (U->Dx != nullptr));
    for (int b = 0; b < nb; ++b) {
        LF_ARG(ctx, xtilde[b] && xnew[b] && xtilde[b]->n == x->n && xnew[b]->n == x->n && xnew[b]->p != x->p && xnew[b]->p != xtilde[b]->p);
        for (int c = 0; c < b; ++c) LF_ARG(ctx, xnew[b]->p != xnew[c]->p);
    }
    const int ml = (int)cons->m_lin;
    const int64_t N = cons->Jct->n;
    const int wm = (U->A && U->W && U->A->p == cons->Jct->p && U->A->m <= kOnepassMaxCols) ? (int)U->A->m : 0;
    if (!wm || m > kNRMaxM || !onepass_cw(ctx, wm, cons->Jct->ld, N) || wm > kOnepassMaxCols)
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "batched Newton retraction needs the one-stream step (generator known, 4..1024 columns)");
    LF_TRY(ensure_mvec(ctx, (size_t)kNRBatchMax * m + 8));            // h_m: the trials' constraint values on their way back
    // instantiated batch widths (a missing trial is born finished): 2 and 4 on the VALU form of the one-pass kernel, 16 on the matrix
    // cores (ctx->tune_nrb_mfma: 1 = the matrix-core form for every batch its shape covers, -1 = never)
    // (from three trials on the matrix-core form is at least as fast without bounds -- 2.0-2.2 ms against 2.0-2.25 for four trials at 1e7 x 128 --
    // and much faster with them: 2.6 against 3.5 ms)
    const bool wide = nrb_wide_shape(wm, ml) && ctx->tune_nrb_mfma >= 0 && (nb > 2 || ctx->tune_nrb_mfma > 0);      // the matrix-core form of 133 .. 528 columns
    const bool mfma = wide || (nrb_mfma_shape(wm, ml) && ctx->tune_nrb_mfma >= 0 && (nb > 2 || ctx->tune_nrb_mfma > 0));
    const int NBk = wide ? kNRBWideTrials : (mfma ? kNRBatchMax : (nb <= 2 ? 2 : 4));
    const size_t mm = (size_t)m * m;
    const int wstride = (int)round_up(wm, 2);
    const int rawn = NBk * ml + NBk;
    LF_TRY(ensure_small(ctx, (size_t)NBk * mm + mm + (size_t)wm * m + (size_t)NBk * wstride + 8 * (size_t)m * (NBk + 1) + rawn + 256));
    double* dD = ctx->small;                           // NBk x (m x m)
    double* dVt = dD + (size_t)NBk * mm;
    double* dW = dVt + mm;
    double* dwdelta = dW + (size_t)wm * m + ((wm * m) & 1);
    double* dSig = dwdelta + (size_t)NBk * wstride;
    double* db = dSig + m + (m & 1);
    double* dcval = db + m + (m & 1);                  // NBk x m
    double* ddelta = dcval + (size_t)NBk * m + ((NBk * m) & 1);
    double* draw = ddelta + (size_t)NBk * m + ((NBk * m) & 1);   // [NBk x ml products ; NBk ball partials]
    LF_HIP(ctx, hipMemcpyAsync(dVt, Vt, sizeof(double) * mm, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipMemcpyAsync(dW, U->W, sizeof(double) * (size_t)wm * m, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipMemcpyAsync(dSig, Sigma, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    if (ml > 0) LF_HIP(ctx, hipMemcpyAsync(db, cons->b, sizeof(double) * ml, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipMemsetAsync(draw, 0, sizeof(double) * rawn, ctx->stream));
    LF_HIP(ctx, hipMemsetAsync(ctx->istat + I_NRB, 0, sizeof(int64_t) * (4 * kNRBatchMax + 1), ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));                     // caller-owned pageable sources
    volatile int64_t* hstat = ctx->h_istat;
    for (int k = 0; k < kNRRing; ++k) hstat[kNRRingOff + k] = 0;
    hstat[I_NR_STATUS] = 0;

    NRSmallB sb;
    sb.nb = NBk;
    sb.all = ctx->istat + I_NRB_ALL;
    sb.hstat = ctx->h_istat;
    lfpsqp_vec* xn[kNRBatchMax];
    for (int b = 0; b < NBk; ++b) {
        const int src = b < nb ? b : 0;               // padding trials alias trial 0 and never run
        xn[b] = xnew[src];
        sb.t[b] = NRSmall{dD + (size_t)b * mm, dVt, dSig, db, dcval + (size_t)b * m, ddelta + (size_t)b * m, draw + (size_t)b * ml,
                          draw + (size_t)NBk * ml + b, dW, dwdelta + (size_t)b * wstride, ctx->istat + I_NRB + 4 * b, nullptr, m, ml,
                          cons->has_ball ? 1 : 0, wm, cons->R2, tol, maxiter};
    }
    const NRStepE ep{xnew[0]->p, x->p, ineq ? lfpsqp_half_stride(N) : 0, ineq ? U->sx->p : nullptr, ineq ? U->sy->p : nullptr,
                     ineq ? idata->q->p : nullptr, ineq ? idata->r->p : nullptr, ineq ? idata->s->p : nullptr,
                     ineq ? idata->t->p : nullptr, cons->n_x, cons->slack_row, cons->has_ball ? 1 : 0, ctx->istat};
    for (int b = 0; b < nb; ++b) {                                                                  // :116-124 per trial
        LF_TRY(lfpsqp_vec_copy(ctx, xnew[b], xtilde[b]));
        if (ineq) LF_TRY(lfpsqp_y_retract(ctx, xnew[b], x, idata));
        if (mfma) continue;                       // c! of all trial points in ONE pass over Jct, below
        if (ml > 0) LF_TRY(run_gemv_t(ctx, cons->Jct, ml, N, PlainVec{xnew[b]->p}, draw + (size_t)b * ml));
        if (cons->has_ball)
            LF_TRY((run_vec<BallF, 1, NoPost>(ctx, N, BallF{xnew[b]->p, cons->n_x, cons->slack_row}, 0u, draw + (size_t)NBk * ml + b, NoPost())));
    }
    if (mfma) {
        // the step kernel with zero coefficients and nothing updated or stored (eval_only): its second product and ball partials ARE c! at the
        // sixteen points -- one pass instead of a GEMV-T per trial
        NRStepE e0 = ep;
        e0.eval_only = 1;
        if (!ctx->d_zeros) {
            LF_HIP(ctx, hipMalloc((void**)&ctx->d_zeros, sizeof(double) * (kOnepassMaxCols + 8)));
            LF_HIP(ctx, hipMemsetAsync(ctx->d_zeros, 0, sizeof(double) * (kOnepassMaxCols + 8), ctx->stream));
        }
        lfpsqp_vec* xe[kNRBatchMax];
        for (int b = 0; b < kNRBatchMax; ++b) xe[b] = xnew[b < nb ? b : 0];
        if (wide && ineq) LF_TRY((nrb_wide_step<true>(ctx, cons, wm, ml, N, ctx->d_zeros, 0, e0, xe, nb, draw)));
        else if (wide) LF_TRY((nrb_wide_step<false>(ctx, cons, wm, ml, N, ctx->d_zeros, 0, e0, xe, nb, draw)));
        else if (ineq) LF_TRY((nrb_mfma_step<true>(ctx, cons, wm, ml, N, ctx->d_zeros, 0, e0, xe, nb, draw)));
        else LF_TRY((nrb_mfma_step<false>(ctx, cons, wm, ml, N, ctx->d_zeros, 0, e0, xe, nb, draw)));
    }
    if (NBk > nb) {                                    // born finished: status 1, flag 1
        const int64_t fin[3] = {1, 0, 1};
        for (int b = nb; b < NBk; ++b)
            LF_HIP(ctx, hipMemcpyAsync(ctx->istat + I_NRB + 4 * b, fin, sizeof(fin), hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    hipLaunchKernelGGL(nr_small_batch_kernel, dim3(NBk), dim3(kNRThreads), 0, ctx->stream, sb, 1);
    hipLaunchKernelGGL(nr_batch_all_kernel, dim3(1), dim3(1), 0, ctx->stream, sb, 1, (int64_t)-1);
    LF_LAUNCH_CHECK(ctx);
    LF_HIP(ctx, hipEventRecord(ctx->ev_slot[0], ctx->stream));
    int64_t it = 0;
    bool done = false;
    while (!done && it < maxiter) {
        if (wide) {
            if (ineq) LF_TRY((nrb_wide_step<true>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, nb, draw)));
            else LF_TRY((nrb_wide_step<false>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, nb, draw)));
        } else if (mfma) {
            if (ineq) LF_TRY((nrb_mfma_step<true>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, nb, draw)));
            else LF_TRY((nrb_mfma_step<false>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, nb, draw)));
        } else if (NBk == 2) {
            if (ineq) LF_TRY((nr_batch_step<true, 2>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, draw)));
            else LF_TRY((nr_batch_step<false, 2>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, draw)));
        } else {
            if (ineq) LF_TRY((nr_batch_step<true, 4>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, draw)));
            else LF_TRY((nr_batch_step<false, 4>(ctx, cons, wm, ml, N, dwdelta, wstride, ep, xn, draw)));
        }
        hipLaunchKernelGGL(nr_small_batch_kernel, dim3(NBk), dim3(kNRThreads), 0, ctx->stream, sb, 0);
        hipLaunchKernelGGL(nr_batch_all_kernel, dim3(1), dim3(1), 0, ctx->stream, sb, 0, it);
        LF_LAUNCH_CHECK(ctx);
        LF_HIP(ctx, hipEventRecord(ctx->ev_slot[(it + 1) & 3], ctx->stream));
        LF_HIP(ctx, hipEventSynchronize(ctx->ev_slot[it & 3]));            // all-done word after `it` completed steps
        if (hstat[kNRRingOff + (it % kNRRing)] != 0) done = true;
        ++it;
    }
    int64_t hist[4 * kNRBatchMax];
    LF_HIP(ctx, hipMemcpyAsync(ctx->h_m, dcval, sizeof(double) * (size_t)nb * m, hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipMemcpyAsync(hist, ctx->istat + I_NRB, sizeof(hist), hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < nb; ++b) {
        for (int k = 0; k < m; ++k) cval[(size_t)b * m + k] = ctx->h_m[(size_t)b * m + k];
        iters[b] = hist[4 * b + 1];
        flags[b] = (int)hist[4 * b + 2];
    }
    return 0;
}

int lfpsqp_retract_pp(lfpsqp_ctx* ctx, const lfpsqp_constraints* cons, lfpsqp_cfun cfun, lfpsqp_jacfun jacfun, void* user,
                      lfpsqp_mat* Jct, int64_t m64, const lfpsqp_ineq_data* idata, lfpsqp_vec* Dx, lfpsqp_vec* Dy, lfpsqp_vec* S,
                      const lfpsqp_vec* xtilde, const lfpsqp_vec* x, lfpsqp_vec* xnew, double mu0, double tol, int64_t maxiter,
                      int64_t maxiter_pcg, const lfpsqp_pp_work* w, double* cval, int* flag_out, int64_t* iters, int64_t* pcg_iters) {
    LF_RANGE("lfpsqp_retract_pp");
    LF_ARG(ctx, ctx && Jct && xtilde && x && xnew && w && cval && flag_out && iters && pcg_iters && m64 >= 1 && m64 <= Jct->m);
    LF_ARG(ctx, (cfun && jacfun) || cons_ok(cons));
    LF_ARG(ctx, w->r && w->p && w->z && w->dx && w->g && w->tmp_m && w->tmp_m->n >= m64);
    const bool ineq = idata != nullptr;
    const int m = (int)m64;
    const int64_t N = Jct->n;
    if (ineq) LF_ARG(ctx, Dx && Dy && S && w->tmp_w && w->h && w->DxS && w->DyS && w->ones && w->zeros && idata->n == N);
    {
        // everything this call will ask of the m-vector staging area, reserved NOW: the pointers taken from it below must survive the Gram
        // passes and the preconditioned solves of the loop (ensure_mvec reallocates when it has to grow)
        const size_t np = ((size_t)m + 127) / 128;
        const size_t gram_doubles = w->precondition ? np * (np + 1) / 2 * 16384 + 64 : 0;
        LF_TRY(ensure_mvec(ctx, std::max((size_t)8 * m + 128, gram_doubles)));
    }
    lfpsqp_vec cdev;                       // replicated device copy of cval (rhs of J' c)
    cdev.p = ctx->d_m + round_up(m + 8, 2);
    cdev.n = m;
    cdev.cap = m;
    double* h_c = ctx->h_m + round_up(m + 8, 2);
    // fulljac in lfpsqp_basis form (src/retractions.jl:324): plain Jct, or [[diag Dx.*S, Jct]; [diag Dy.*S, 0]]
    lfpsqp_basis Jop;
    Jop.Z = Jct;
    Jop.ncols = m;
    Jop.Dx = ineq ? w->DxS : nullptr;
    Jop.Dy = ineq ? w->DyS : nullptr;
    Jop.sx = ineq ? w->ones : nullptr;
    Jop.sy = ineq ? w->zeros : nullptr;
    Jop.A = nullptr;
    Jop.W = nullptr;
    Jop.S = (cons && !cfun && cons->Jsp && cons->Jsp->m == m) ? cons->Jsp : nullptr;    // constant sparse equality block: sparse inner solves

    auto eval_c = [&](double* out) -> int {
        if (cfun) {
            int rc = cfun(user, xnew, out);
            if (rc != 0) return set_err(ctx, LFPSQP_ERR_ARG, "user c! callback returned %d", rc);
            return 0;
        }
        return cons_eval(ctx, cons, xnew, out);
    };
    auto eval_jac = [&]() -> int {
        if (jacfun) {
            int rc = jacfun(user, xnew, Jct, cval);
            if (rc != 0) return set_err(ctx, LFPSQP_ERR_ARG, "user jac! callback returned %d", rc);
            return 0;
        }
        return lfpsqp_constraints_jac(ctx, cons, xnew, Jct, cval);
    };
    auto dist2_of_g = [&](double* out) -> int { return lfpsqp_dot(ctx, w->g, w->g, out); };
    auto nanmax_h = [](double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); };

    int flag = 0;
    LF_TRY(lfpsqp_vec_copy(ctx, xnew, xtilde));                                   // :329
    double mu = mu0;
    int64_t i = 0, pcg_total = 0;
    double hh = 0.0;
    // host buffers of the exact preconditioner (DeviceOptions.pp_precondition): allocated once per retraction, reused by every Gauss-Newton step
    std::vector<double> G, Lc, Kh, Lt, ywork;
    while (i < maxiter) {
        LF_TRY(eval_jac());                                                       // :340 (+ transpose!, :347)
        double curtol = 0.0;
        for (int k = 0; k < m; ++k) curtol = nanmax_h(curtol, fabs(cval[k]));
        if (ineq) {                                                               // :343-353
            LF_TRY(lfpsqp_inequality_gradient(ctx, xnew, idata, Dx, Dy, S, nullptr, nullptr));
            LF_TRY(lfpsqp_vmul(ctx, Dx, S, w->DxS));
            LF_TRY(lfpsqp_vmul(ctx, Dy, S, w->DyS));
            double hmax = 0.0;
            LF_TRY(lfpsqp_calculate_h(ctx, w->h, xnew, idata, &hmax));
            curtol = nanmax_h(curtol, hmax);
            LF_TRY(lfpsqp_dot(ctx, w->h, w->h, &hh));
        }
        if (curtol < tol) break;                                                  // :359
        LF_TRY(lfpsqp_waxpby(ctx, 1.0, xnew, -1.0, xtilde, w->g));                // :364
        double cc = 0.0, gg = 0.0;
        for (int k = 0; k < m; ++k) cc += cval[k] * cval[k];
        LF_TRY(dist2_of_g(&gg));
        const double prev_obj_val = (hh + cc) + mu * gg;                          // :366
        for (int k = 0; k < m; ++k) h_c[k] = cval[k];
        LF_HIP(ctx, hipMemcpyAsync(cdev.p, h_c, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
        LF_TRY(lfpsqp_q_gemv_n(ctx, &Jop, 1.0, ineq ? w->h : nullptr, &cdev, mu, w->g));   // :369  g = fulljac' cvalaug + mu g
        LF_TRY(lfpsqp_vec_fill(ctx, w->dx, 0.0));
        LF_TRY(lfpsqp_vec_copy(ctx, w->r, w->g));
        int pcg_flag = 0;
        int64_t pcg_i = 0;
        bool pre_done = false;
        if (w->precondition && w->q && !Jop.S && (!ineq || (w->i11 && w->i12 && w->i22)) && onepass_cw(ctx, m, Jct->ld, N) != 0) {
            // exact preconditioner of this Gauss-Newton step's operator (DeviceOptions.pp_precondition; lfpsqp_pcg_pre): K = (I + G)^-1 with
            // G = Jct' D0^-1 Jct from one Gram pass over the CURRENT Jct
            if (ineq) LF_TRY((run_vec<D0InvF, 0, NoPost>(ctx, N, D0InvF{w->DxS->p, w->DyS->p, w->i11->p, w->i12->p, w->i22->p, mu}, 0u, nullptr, NoPost())));
            G.resize((size_t)m * m);
            Kh.resize((size_t)m * m);
            LF_TRY(lfpsqp_gram(ctx, Jct, m, ineq ? w->i11 : nullptr, G.data()));
            for (int j = 0; j < m; ++j)
                for (int k = 0; k < m; ++k) {
                    double v = ineq ? G[(size_t)j * m + k] : G[(size_t)j * m + k] / mu;
                    if (j == k) v += 1.0;
                    G[(size_t)j * m + k] = v;
                }
            bool finite = true;
            for (double v : G) finite = finite && (v == v) && fabs(v) < 1e300;
            if (finite && cholesky_lower(m, G, Lc, 0.0)) {
                // K = L^-T L^-1: solve L Y = I (forward), then L' K = Y (backward), column by column.  The columns are independent: shared
                // over the host threads, the forward solve on a transposed copy of L so that both sweeps run down contiguous memory
                // (2 m^3 / 3 flops per Gauss-Newton step: 1.4 Mflop at config 4's m = 129, 0.7 Gflop at m = 1024)
                Lt.resize((size_t)m * m);
                [[maybe_unused]] const int nth = m >= 64 ? small_threads() : 1;
                ywork.resize((size_t)m * (size_t)(nth > 1 ? nth : 1));
                for (int j = 0; j < m; ++j)
                    for (int i2 = j; i2 < m; ++i2) Lt[(size_t)i2 * m + j] = Lc[(size_t)j * m + i2];       // Lt[i, j] row-major = L[i, j]
#pragma omp parallel for if (nth > 1) num_threads(nth) schedule(dynamic, 8)
                for (int c = 0; c < m; ++c) {
                    double* y = ywork.data() + (size_t)m * (size_t)small_thread_id();
                    for (int i2 = 0; i2 < c; ++i2) y[i2] = 0.0;                              // (the unit vector's leading zeros: the backward sweep reads them)
                    for (int i2 = c; i2 < m; ++i2) {
                        const double* li = &Lt[(size_t)i2 * m];
                        double acc = (i2 == c) ? 1.0 : 0.0;
                        for (int k = c; k < i2; ++k) acc -= li[k] * y[k];
                        y[i2] = acc / li[i2];
                    }
                    double* kc = &Kh[(size_t)c * m];
                    for (int i2 = m - 1; i2 >= 0; --i2) {
                        const double* lc = &Lc[(size_t)i2 * m];
                        double acc = y[i2];
                        for (int k = i2 + 1; k < m; ++k) acc -= lc[k] * kc[k];
                        kc[i2] = acc / lc[i2];
                    }
                }
                lfpsqp_pcg_precond pc{Kh.data(), ineq ? w->i11 : nullptr, ineq ? w->i12 : nullptr, ineq ? w->i22 : nullptr, w->q};
                LF_TRY(lfpsqp_pcg_pre(ctx, mu, &Jop, &pc, w->dx, w->r, w->p, w->z, tol, maxiter_pcg, &pcg_flag, &pcg_i));
                pre_done = true;
            }
        }
        if (!pre_done)
            LF_TRY(lfpsqp_pcg(ctx, mu, &Jop, w->dx, w->r, w->p, w->z, w->tmp_w, w->tmp_m, tol, maxiter_pcg, &pcg_flag, &pcg_i));   // :375
        pcg_total += pcg_i;
        if (pcg_flag > 0) { flag = 2; break; }                                    // :377-381
        LF_TRY(lfpsqp_vec_copy(ctx, w->p, xnew));                                 // :384
        double gdx = 0.0;
        LF_TRY(lfpsqp_dot(ctx, w->g, w->dx, &gdx));
        const double ar_dot = -gdx;                                               // :385
        double alpha = 1.0;
        LF_TRY(lfpsqp_axpby(ctx, -alpha, w->dx, 1.0, xnew));                      // :389
        LF_TRY(lfpsqp_waxpby(ctx, 1.0, xnew, -1.0, xtilde, w->g));
        double dist2 = 0.0;
        LF_TRY(dist2_of_g(&dist2));
        LF_TRY(eval_c(cval));                                                     // :392
        if (ineq) {
            LF_TRY(lfpsqp_calculate_h(ctx, w->h, xnew, idata, nullptr));
            LF_TRY(lfpsqp_dot(ctx, w->h, w->h, &hh));
        }
        cc = 0.0;
        for (int k = 0; k < m; ++k) cc += cval[k] * cval[k];                      // :399
        int armijo_count = 0;
        while ((hh + cc) + mu * dist2 > prev_obj_val + 1e-4 * alpha * ar_dot) {   // :403
            alpha /= 2;
            LF_TRY(lfpsqp_waxpby(ctx, 1.0, w->p, -alpha, w->dx, xnew));
            LF_TRY(lfpsqp_waxpby(ctx, 1.0, xnew, -1.0, xtilde, w->g));
            LF_TRY(dist2_of_g(&dist2));
            // BUG-COMPAT :410-417: c! lands in cvalaug and is overwritten by the stale full-step cval; only h and
            // dist2 change.  The discarded c! evaluation is skipped.
            if (ineq) {
                LF_TRY(lfpsqp_calculate_h(ctx, w->h, xnew, idata, nullptr));
                LF_TRY(lfpsqp_dot(ctx, w->h, w->h, &hh));
            }
            if (++armijo_count == 100) { flag = 3; break; }                       // :422-425
        }
        ++i;
        const double nrm = sqrt(hh + cc);
        mu = (mu * 0.1 < nrm) ? mu * 0.1 : nrm;                                   // :431  min(0.1 mu, norm(cvalaug))
    }
    if (i == maxiter) flag = 1;                                                   // :435-437
    *flag_out = flag;
    *iters = i;
    *pcg_iters = pcg_total;
    return 0;
}

}  // extern "C"

