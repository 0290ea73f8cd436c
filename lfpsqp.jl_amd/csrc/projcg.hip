// Projected conjugate gradient on the device -- the reference's projcg!
// (src/projcg.jl:40-121).
//
// DEFAULT: ONE pass over U per iteration (fused iteration, "F" below).  The projection of iteration k,
//   gp = rp - U (U'rp)  (:95-97), needs U twice: U'rp (all rows) and then U*(U'rp).  But
//   rp_{k+1} = g_{k+1} + alpha_{k+1} A d_{k+1},  d_{k+1} = -g_{k+1} + beta_{k+1} d_k   (:93, :99)  is LINEAR in vectors that are
//   known row by row as soon as g_{k+1} = gp exists, so
//       U'rp_{k+1} = t1 + alpha_{k+1} * t3_{k+1},   t1 = U'g_{k+1},  t2 = U'(A g_{k+1}),  t3_{k+1} = U'(A d_{k+1}) = -t2 + beta_{k+1} t3_k
//   and t1, t2 are accumulated by the SAME kernel that forms g_{k+1}, over the row tile it still holds in registers
//   (onepass_kernel, kernels.h).  In the same way the next d'Ad follows from three sums of row-local products,
//       d+'A d+ = g+'A g+ - 2 beta+ g+'A d + beta+^2 d'A d        (d+ = beta+ d - g+),
//   so an iteration needs ONE global reduction (one all-reduce of 2m+5 doubles across ranks).  Per iteration:
//     K1  x += alpha_prev*d (deferred :92) ; d = beta*d - g                      (vec_kernel, no reduction)   :99
//     F   rp = g + alpha A d (registers) ; gp = rp - U*Utr ; g = gp ;
//         partials rp'gp, gp'gp, gp'A gp, gp'A d, d'A d, U'gp, U'(A gp)           (onepass_kernel)            :93-103
//     post  beta :98, rg, nr, convergence / iteration limit :103-111 ; next d'Ad, exits :77-87, alpha :91 ; t3, Utr
//   = 8 n m + 72 n bytes instead of 16 n m + 104 n.  beta, rg, nr and the exit tests are the reference's; U'rp and d'Ad
//   are assembled from exactly computed pieces instead of one product each -- a difference of the same size as a change
//   of summation order (t1 = U'g re-measures the basis component of g every iteration, nothing is a recurrence except
//   t3, which t1 corrects).
//   Used for a diagonal operator A and 4 <= m <= 1024 columns; otherwise (and with LFPSQP_ONEPASS=-1):
//
// FALLBACK: three fused streaming kernels per iteration, two passes over U:
//
//   K1  x += alpha_prev*d (deferred :92) ; d = beta*d - g ; partial d'(A d)   (vec_kernel)     :99, :74-75
//   K2  alpha = rg/dAd ; rp = g + alpha A d formed on the fly ; partial U' rp (gemv_t_kernel)  :91-96
//   K3  gp = rp - U (U' rp), rp recomputed ; g = gp ; partial rp'gp, gp'gp    (gemv_n_kernel)  :97-103
//
// r == g throughout the reference loop (:61, :100-101), so r is not stored; gp, Ad AND rp are
// never materialised inside the loop: measured on MI355X, n-vector STORES issued inside the
// matrix stream cost ~3-4x their bytes (K2 with its x/rp stores 1.83 ms, without 1.60 ms), so
// the iteration is arranged to store only x, d (in the small vector kernel K1) and g (K3):
// 3 n-vector writes + 10 n-vector reads + two passes over U = 16 n m + 104 n bytes
// (SURVEY §8d's fused minimum is 16 n m + 96 n with 4 writes; the reference moves ~27 passes).
// The x-update of iteration k is applied by K1 of iteration k+1 (or by a final flush).  All scalars (alpha, beta, rg, nr), the iteration counter
// and the exit status live in device memory; every kernel starts with a uniform
// "already finished?" test, so the host may enqueue one iteration ahead of the
// status it has seen (no pipeline bubble) and extra launches are no-ops.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "internal.h"
#include "sparse.h"

namespace lfpsqp {

enum { S_DAD = 0, S_RG = 1, S_ALPHA = 2, S_RPGP = 3, S_GPGP = 4, S_BETA = 5, S_NR = 6, S_TOL = 7, S_DD = 8, S_ALPHA_PREV = 9 };
enum { I_STATUS = 0, I_ITER = 1, I_MAXIT = 2 };
enum { ST_RUNNING = 0, ST_CONVERGED = 1, ST_RG_BREAK = 2, ST_NEGCURV = 3, ST_MAXIT = 4 };

struct AOpD {  // A = a0*I + diag(dg)
    double a0;
    const double* dg;
    __device__ __forceinline__ double2 apply(int64_t i, double2 d) const {
        if (dg) {
            const double2 q = ld2(dg + i);
            return make_double2((a0 + q.x) * d.x, (a0 + q.y) * d.y);
        }
        return make_double2(a0 * d.x, a0 * d.y);
    }
};

struct AOpV {  // generic A behind a callback: the product A*v for the ONE vector v in question sits in a device vector
    const double* Av;
    __device__ __forceinline__ double2 apply(int64_t i, double2) const { return ld2(Av + i); }
};

struct AOpLR {  // A = a0*I + diag(dg) + V diag(sigma) V': (A v)_i = (a0 + dg_i) v_i + sum_j V_ij sigma_j (V'v)_j, with V'v (k, device) formed before the launch
    double a0;
    const double* dg;
    const double* V;
    int64_t ldv;
    int k;
    const double *sigma, *vtv;
    __device__ __forceinline__ double2 apply(int64_t i, double2 d) const {
        double2 o = dg ? make_double2((a0 + dg[i]) * d.x, (a0 + dg[i + 1]) * d.y) : make_double2(a0 * d.x, a0 * d.y);
        for (int j = 0; j < k; ++j) {
            const double c = ld_scal(sigma + j) * ld_scal(vtv + j);
            const double2 v = ld2(V + (int64_t)j * ldv + i);
            o.x = fma(v.x, c, o.x);
            o.y = fma(v.y, c, o.y);
        }
        return o;
    }
};

// A = a0*I + diag(dg) + tridiagonal couplings: (A v)_i = (a0 + dg_i) v_i + off_{i-1} v_{i-1} + off_i v_{i+1}  (off_i couples rows i and i+1; off_{n-1}
// is ignored).  lfpsqp_projcg_tridiag / lfpsqp_tridiag_mul.
struct TriD {
    double a0;
    const double* dg;
    const double* off;
    int64_t n;
};
struct TriMulF {   // out = A v (a plain vector kernel: the neighbours come out of the cache lines the row itself brings in)
    TriD A;
    const double* v;
    double* out;
    const int64_t* istat;     // nullptr: always; else only while the solve is running
    __device__ __forceinline__ bool skip() const { return istat != nullptr && ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (!v0) return;
        const double2 vv = ld2(v + i), of = ld2(A.off + i);
        const double2 dd = A.dg ? ld2(A.dg + i) : make_double2(0.0, 0.0);
        const double vm = (i > 0) ? v[i - 1] : 0.0, om = (i > 0) ? A.off[i - 1] : 0.0;
        double o0 = fma(om, vm, (A.a0 + dd.x) * vv.x);
        if (v1) o0 = fma(of.x, vv.y, o0);
        if (!v1) { out[i] = o0; return; }
        double o1 = fma(of.x, vv.x, (A.a0 + dd.y) * vv.y);
        if (i + 2 < A.n) o1 = fma(of.y, v[i + 2], o1);
        st2(out + i, make_double2(o0, o1));
    }
};

// What the fused tridiagonal iteration (PcgFuseTri) needs from the rows' NEIGHBOURS, as two stored vectors, so that the pass itself is row-local
// and may update the residual in place:  ad = A d  and  q_i = off_{i-1} rr_{i-1} + off_i rr_{i+1}  with  rr = g + alpha ad  (the off-diagonal part
// of A rr).  INIT: rr is the stored initial residual (alpha = 0, no direction yet), only q is written.  A vector kernel: six streams of n doubles.
template <bool INIT>
struct TriPrepF {
    TriD A;
    const double* g;          // the residual (INIT: the stored initial residual)
    const double* d;
    double* ad;
    double* q;
    const double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    // entry k of a vector / of the couplings, zero outside the matrix (clamped index + select: no divergent loads)
    __device__ __forceinline__ double at(const double* v, int64_t k) const {
        const int64_t kc = k < 0 ? 0 : (k >= A.n ? A.n - 1 : k);
        const double x = v[kc];
        return (k == kc) ? x : 0.0;
    }
    __device__ __forceinline__ double cpl(int64_t k) const {      // off_k couples rows k and k + 1: 0 <= k < n - 1
        const int64_t kc = k < 0 ? 0 : (k >= A.n ? A.n - 1 : k);
        const double x = A.off[kc];
        return (k >= 0 && k + 1 < A.n) ? x : 0.0;
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (!v0) return;
        const double alpha = INIT ? 0.0 : ld_scal(scal + S_ALPHA);
        // rows i-1 .. i+2 of rr (and of A d): the two rows of this thread and one neighbour on either side
        double o[5], adv[4], rr[4], gv[4], dv[6], axv[4];
        if (i >= 2 && i + 4 < A.n) {                                             // interior (all but the first and the last thread): aligned pair loads
            const double2 o0 = ld2(A.off + i - 2), o1 = ld2(A.off + i);
            o[0] = o0.x; o[1] = o0.y; o[2] = o1.x; o[3] = o1.y; o[4] = A.off[i + 2];
            const double2 g1 = ld2(g + i);
            gv[0] = g[i - 1]; gv[1] = g1.x; gv[2] = g1.y; gv[3] = g[i + 2];
            if (!INIT) {
                const double2 d0 = ld2(d + i - 2), d1 = ld2(d + i), d2 = ld2(d + i + 2);
                dv[0] = d0.x; dv[1] = d0.y; dv[2] = d1.x; dv[3] = d1.y; dv[4] = d2.x; dv[5] = d2.y;
                if (A.dg) {
                    const double2 a1 = ld2(A.dg + i);
                    axv[0] = A.a0 + A.dg[i - 1]; axv[1] = A.a0 + a1.x; axv[2] = A.a0 + a1.y; axv[3] = A.a0 + A.dg[i + 2];
                } else {
                    axv[0] = axv[1] = axv[2] = axv[3] = A.a0;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) o[k] = cpl(i - 2 + k);                   // off_{i-2} .. off_{i+2}
#pragma unroll
            for (int k = 0; k < 4; ++k) gv[k] = at(g, i - 1 + k);
            if (!INIT) {
#pragma unroll
                for (int k = 0; k < 6; ++k) dv[k] = at(d, i - 2 + k);            // d_{i-2} .. d_{i+3}
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t r = i - 1 + k;
                    const int64_t rc = r < 0 ? 0 : (r >= A.n ? A.n - 1 : r);
                    axv[k] = A.a0 + (A.dg ? A.dg[rc] : 0.0);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                            // row i - 1 + k
            adv[k] = INIT ? 0.0 : fma(o[k + 1], dv[k + 2], fma(o[k], dv[k], axv[k] * dv[k + 1]));   // same expression as TriMulF's for every row
            rr[k] = INIT ? gv[k] : fma(alpha, adv[k], gv[k]);                    // :93
        }
        const double q0 = fma(o[2], rr[2], o[1] * rr[0]);
        const double q1 = fma(o[3], rr[3], o[2] * rr[1]);
        if (v1) {
            st2(q + i, make_double2(q0, q1));
            if (!INIT) st2(ad + i, make_double2(adv[1], adv[2]));
        } else {
            q[i] = q0;
            if (!INIT) ad[i] = adv[1];
        }
    }
};

struct StackD {   // stacked (bound-constrained) basis Q = [[diag Dx; diag Dy], [sx.*Z; sy.*Z]]
    int64_t hs;
    const double *Dx, *Dy, *sx, *sy;
};

// ---- K1 -----------------------------------------------------------------------
struct PcgDirF {
    double* d;
    const double* g;
    double* x;
    AOpD A;
    const double* scal;
    const int64_t* istat;
    int first;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        double2 dd = ld2(d + i);
        if (!first) {
            const double alpha = ld_scal(scal + S_ALPHA);               // x += alpha d of the previous iteration (:92)
            double2 xx = ld2(x + i);
            xx = make_double2(fma(alpha, dd.x, xx.x), fma(alpha, dd.y, xx.y));
            if (v1) st2(x + i, xx);
            else if (v0) x[i] = xx.x;
            const double beta = ld_scal(scal + S_BETA);
            const double2 gg = ld2(g + i);
            dd = make_double2(beta * dd.x - gg.x, beta * dd.y - gg.y);   // src/projcg.jl:99
            if (v1) st2(d + i, dd);
            else if (v0) d[i] = dd.x;
        }
        const double2 ad = A.apply(i, dd);                                // :74
        double s = 0.0;
        if (v0) s = dd.x * ad.x;
        if (v1) s = fma(dd.y, ad.y, s);
        red[0] += s;                                                      // :75
    }
};
// generic-operator flow: x += alpha*d of the previous iteration (:92) ; d = beta*d - g (:99) -- A d comes from the callback afterwards
struct PcgDirX {
    double* d;
    const double* g;
    double* x;
    const double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double alpha = ld_scal(scal + S_ALPHA), beta = ld_scal(scal + S_BETA);
        double2 dd = ld2(d + i), xx = ld2(x + i);
        const double2 gg = ld2(g + i);
        xx = make_double2(fma(alpha, dd.x, xx.x), fma(alpha, dd.y, xx.y));
        dd = make_double2(beta * dd.x - gg.x, beta * dd.y - gg.y);
        if (v1) { st2(x + i, xx); st2(d + i, dd); }
        else if (v0) { x[i] = xx.x; d[i] = dd.x; }
    }
};
struct DotPairF {   // d'(A d) from the two vectors (:75)
    const double* a;
    const double* b;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 x = ld2(a + i), y = ld2(b + i);
        double s = 0.0;
        if (v0) s = x.x * y.x;
        if (v1) s = fma(x.y, y.y, s);
        red[0] += s;
    }
};
// The post-ops also publish (status, iteration, nr) into a pinned HOST block with system-scope
// stores, so the host follows the solve without any per-iteration device-to-host copy.  Besides the
// "latest" words there is a RING of per-iteration status words: the host decides whether to stop from
// the entry of ONE fixed iteration (it - 2) after that iteration's event has completed, so the decision --
// and with it the number of enqueued iterations and COLLECTIVE calls -- is identical on every rank no
// matter how far each rank's GPU has run ahead (a "latest status" read would be rank-timing dependent
// and could desynchronise the RCCL call sequence).
constexpr int kRing = 8, kRingOff = 8;
struct HostMirror {
    int64_t* hstat;  // [0] status, [1] iterations started, [kRingOff + (iter % kRing)] status after iteration `iter`
    double* hnr;     // [0] nr
    __device__ __forceinline__ void publish(int64_t status, int64_t iter, double nr) const {
        __hip_atomic_store(hnr, nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hstat + 1, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hstat + kRingOff + (iter % kRing), status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hstat, status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

struct PcgPost1 {  // after d'Ad is final: iteration count, exits, alpha  (:72-91)
    double* scal;
    int64_t* istat;
    HostMirror hm;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void run(double*) const {
        const int64_t it = ld_stat(istat + I_ITER) + 1;
        istat[I_ITER] = it;
        const double dAd = ld_scal(scal + S_DAD), rg = ld_scal(scal + S_RG);
        if (dAd <= 0.0) { istat[I_STATUS] = ST_NEGCURV; hm.publish(ST_NEGCURV, it, ld_scal(scal + S_NR)); }
        else if (rg <= 0.0) { istat[I_STATUS] = ST_RG_BREAK; hm.publish(ST_RG_BREAK, it, ld_scal(scal + S_NR)); }
        else scal[S_ALPHA] = rg / dAd;
    }
};

// ---- K2 (producer of v = rp for U' rp; nothing is stored) -------------------------------
template <class AOP>
struct PcgStepV {
    const double* d;
    const double* g;
    AOP A;
    const double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ double2 rp_at(int64_t r) const {
        const double alpha = ld_scal(scal + S_ALPHA);
        const double2 dd = ld2(d + r), gg = ld2(g + r);
        const double2 ad = A.apply(r, dd);
        return make_double2(fma(alpha, ad.x, gg.x), fma(alpha, ad.y, gg.y));      // :93
    }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 rr = rp_at(r);
        return make_double2(v0 ? rr.x : 0.0, v1 ? rr.y : 0.0);
    }
};

// ---- K3 (consumer of U*Utr) -------------------------------------------------------
template <class AOP>
struct PcgProjE {
    const double* rp;   // stored residual: only the initial projection (init = 1) reads it
    double* g;
    double* d;          // written (d = -g) only by the initial projection
    const int64_t* istat;
    int init;
    PcgStepV<AOP> sv;   // recomputes rp = g + alpha A d inside the loop
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double* red) const {
        const double2 rr = init ? ld2(rp + r) : sv.rp_at(r);
        const double2 gp = make_double2(rr.x - acc.x, rr.y - acc.y);      // :97 (alpha=-1, beta=1)
        if (v1) st2(g + r, gp);
        else if (v0) g[r] = gp.x;
        if (init) {
            const double2 nd = make_double2(-gp.x, -gp.y);                // :62
            if (v1) st2(d + r, nd);
            else if (v0) d[r] = nd.x;
        }
        double s0 = 0.0, s1 = 0.0;
        if (v0) { s0 = rr.x * gp.x; s1 = gp.x * gp.x; }
        if (v1) { s0 = fma(rr.y, gp.y, s0); s1 = fma(gp.y, gp.y, s1); }
        red[0] += s0;                                                     // :98 rp'gp
        red[1] += s1;                                                     // :84/:103 (r == g): rg, nr^2
    }
};
struct PcgPost3 {  // beta, next rg, nr, convergence / iteration-limit exits (:98-111, :71)
    double* scal;
    int64_t* istat;
    int init;
    HostMirror hm;
    const double* src;   // [rp'gp, gp'gp] (scal + S_RPGP for the two-pass kernels, the tail of the fused kernel's output otherwise)
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void run(double*) const {
        const double rpgp = ld_scal(src), gpgp = ld_scal(src + 1);
        if (init) {
            scal[S_RG] = gpgp;
            int64_t st0 = ST_RUNNING;
            if (ld_stat(istat + I_MAXIT) <= 0) { istat[I_STATUS] = ST_MAXIT; st0 = ST_MAXIT; }
            hm.publish(st0, 0, INFINITY);
            return;
        }
        scal[S_BETA] = rpgp / ld_scal(scal + S_RG);
        scal[S_RG] = gpgp;
        const double nr = sqrt(gpgp);
        scal[S_NR] = nr;
        const int64_t it = ld_stat(istat + I_ITER);
        int64_t st = ST_RUNNING;
        if (nr < ld_scal(scal + S_TOL)) st = ST_CONVERGED;
        else if (it >= ld_stat(istat + I_MAXIT)) st = ST_MAXIT;
        if (st != ST_RUNNING) istat[I_STATUS] = st;
        hm.publish(st, it, nr);
    }
};

// ---- F: the fused iteration (one pass over U) -----------------------------------------------------------
// Row functor of onepass_kernel: N-product acc = U[row, :] . Utr arrives, the row's residual is projected and stored,
// and the two T-vectors (gp, A gp) go back into the same tile.  INIT: the initial projection (:58-62): rp is the
// stored residual and d = -g is written.  ST = stacked (bound) form, cf. PcgProjES.
template <bool ST, bool INIT>
struct PcgFuseE {
    const double* rp;   // stored initial residual (INIT only)
    const double* g;    // the residual this iteration reads ...
    double* gout;       // ... and where the projected one goes: the two alternate between iterations (see lfpsqp_projcg)
    double* d;
    AOpD A;
    const double* scal;
    const int64_t* istat;
    StackD k;           // stacked only
    static constexpr bool kSplitRed = true;
    struct Uni { double alpha; };
    struct Row { double gx, dx, ax, gy, dy, ay, Dx, Dy, sx, sy; };   // INIT: gx / gy carry the stored residual, dx / dy are unused
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const { return Uni{INIT ? 0.0 : uniform_f64(ld_scal(scal + S_ALPHA))}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.gx = INIT ? at(rp, o) : at(g, o);
        w.dx = INIT ? 0.0 : at(d, o);
        w.ax = A.a0 + (A.dg ? at(A.dg, o) : 0.0);
        if (ST) {
            w.gy = INIT ? at(rp + k.hs, o) : at(g + k.hs, o);
            w.dy = INIT ? 0.0 : at(d + k.hs, o);
            w.ay = A.a0 + (A.dg ? at(A.dg + k.hs, o) : 0.0);
            w.Dx = at(k.Dx, o); w.Dy = at(k.Dy, o); w.sx = at(k.sx, o); w.sy = at(k.sy, o);
        } else {
            w.gy = w.dy = w.ay = w.Dx = w.Dy = w.sx = w.sy = 0.0;
        }
        return w;
    }
    // reductions (logical order in the kernel's output): rp'gp, gp'gp (:98, :84/:103) and the three sums from which the NEXT
    // iteration's d'Ad follows without a second global reduction:
    //     d+ = beta d - gp  =>  d+'A d+ = gp'A gp - 2 beta gp'A d + beta^2 d'A d   (all three direct).
    // The first four are gp times {rp, gp, A gp, A d}: lane group h takes the h-th of them in running sum 0; d'A d is running sum 1
    // of group 0 (kSplitRed: 2 running sums per lane instead of 5).
    // staged form (onepass_kernel STG): the projected residual of the plain iteration goes to the workgroup's LDS slot of the row
    // instead of memory; the kernel stores whole bursts of them through stage_out()
    static constexpr int kStageStreams = INIT ? 0 : (ST ? 2 : 1);          // (stacked: the x and the y half of the residual)
    __device__ __forceinline__ double* stage_out(int sv) const { return sv == 0 ? gout : gout + k.hs; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[2], double (&red)[2]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                                 const Row& w, double (&v)[2], double (&red)[2], double* slot, int sstride) const {
        const double acc = accv[0];
        const bool st = valid && owner, cnt = valid && lead;
        const int h = (int)((threadIdx.x >> 2) & 3u);
        if (!ST) {
            const double ad = w.ax * w.dx;
            const double rr = INIT ? w.gx : fma(u.alpha, ad, w.gx);                  // :93 (same expression as PcgStepV::rp_at)
            const double gp = rr - acc;                                              // :97
            const double ag = w.ax * gp;
            if (st) {
                if (slot) *slot = gp;
                else put(gout, o, gp);
                if (INIT) put(d, o, -gp);                                            // :62
            }
            const double f = (h == 0) ? rr : ((h == 1) ? gp : ((h == 2) ? ag : ad));
            if (cnt) {
                red[0] += gp * f;                                                    // :98, :84 / :103, g'Ag, g'Ad
                if (h == 0) red[1] += w.dx * ad;                                     // d'Ad
            }
            v[0] = valid ? gp : 0.0;
            v[1] = valid ? ag : 0.0;
        } else {
            const double adx = w.ax * w.dx, ady = w.ay * w.dy;
            const double rx = INIT ? w.gx : fma(u.alpha, adx, w.gx);
            const double ry = INIT ? w.gy : fma(u.alpha, ady, w.gy);
            const double ww = w.Dx * rx + w.Dy * ry;                                 // diagonal block of Q'rp
            const double gx = rx - fma(w.sx, acc, w.Dx * ww);
            const double gy = ry - fma(w.sy, acc, w.Dy * ww);
            const double agx = w.ax * gx, agy = w.ay * gy;
            if (st) {
                if (slot) { slot[0] = gx; slot[sstride] = gy; }
                else { put(gout, o, gx); put(gout + k.hs, o, gy); }
                if (INIT) { put(d, o, -gx); put(d + k.hs, o, -gy); }
            }
            const double fx = (h == 0) ? rx : ((h == 1) ? gx : ((h == 2) ? agx : adx));
            const double fy = (h == 0) ? ry : ((h == 1) ? gy : ((h == 2) ? agy : ady));
            if (cnt) {
                red[0] += fx * gx + fy * gy;
                if (h == 0) red[1] += w.dx * adx + w.dy * ady;
            }
            v[0] = valid ? (w.sx * gx + w.sy * gy) : 0.0;                            // the Z-block of Q'g
            v[1] = valid ? (w.sx * agx + w.sy * agy) : 0.0;                          // ... of Q'(A g)
        }
    }
};

// ---- F for A = a0 I + diag(dg) + V diag(sigma) V'  (k <= 8 columns): the same single pass ---------------------------------------------------
// The low-rank term is not row-local -- (A v)_i needs V'v -- but everything the iteration multiplies A with is known through k-vectors:
//   (A d)_i     = D_i d_i + sum_j V_ij c_j,  c = sigma .* (V'd)   -- V'd follows the recurrence of d: V'd+ = beta V'd - V'gp  (post-op);
//   U'(A gp)    = U'(D gp) + (U'V) (sigma .* (V'gp))              -- U'V (m x k) once per solve, V'gp: k more reduction terms of the pass;
//   gp'A gp     = gp'D gp + sum_j sigma_j (V'gp)_j^2,  gp'A d = sum_i gp_i (A d)_i,  d'A d = sum_i d_i (A d)_i   (row-local products).
// Cost next to the diagonal form: k doubles per row (8 k of 8 m + 72 bytes) and k reduction terms; a general Hessian of this shape no longer pays
// two passes per iteration through the callback path (lfpsqp_projcg_op).
constexpr int kLRMax = 8;
template <bool INIT>
struct PcgFuseLR {
    const double* rp;
    const double* g;
    double* gout;
    double* d;
    AOpD A;
    const double* scal;
    const int64_t* istat;
    const double* V;      // n x k, column-major
    int64_t ldv;
    int k;
    const double* vdc;    // device, kLRMax: sigma .* (V'd) of the current direction (zeros beyond k; written by the post-op)
    static constexpr bool kSplitRed = true;
    static constexpr bool kNoRowScale = true;           // (no matrix views: the generator of a nonlinear class has a diagonal Hessian term)
    struct Uni { double alpha; double c[kLRMax]; };
    struct Row { double gx, dx, ax, v0, v1, v2, v3, v4, v5, v6, v7; };      // (named fields: an array member keeps the record in scratch memory)
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const {
        Uni u;
        u.alpha = INIT ? 0.0 : uniform_f64(ld_scal(scal + S_ALPHA));
#pragma unroll
        for (int j = 0; j < kLRMax; ++j) u.c[j] = (!INIT && j < k) ? uniform_f64(ld_scal(vdc + j)) : 0.0;
        return u;
    }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.gx = INIT ? at(rp, o) : at(g, o);
        w.dx = INIT ? 0.0 : at(d, o);
        w.ax = A.a0 + (A.dg ? at(A.dg, o) : 0.0);
        // (unconditional: a branch around a load keeps the whole row record in scratch memory; the columns beyond k re-read column 0 -- the same
        // cache line -- and meet zero coefficients, their sums are never read)
        auto col = [&](int j) { return at(V + (int64_t)(j < k ? j : 0) * ldv, o); };
        w.v0 = col(0); w.v1 = col(1); w.v2 = col(2); w.v3 = col(3); w.v4 = col(4); w.v5 = col(5); w.v6 = col(6); w.v7 = col(7);
        return w;
    }
    static constexpr int kStageStreams = INIT ? 0 : 1;
    __device__ __forceinline__ double* stage_out(int) const { return gout; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[2], double (&red)[4]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    // reductions, logical order: rp'gp, gp'gp, gp'D gp, gp'A d | d'A d, gp'V_0, gp'V_1, gp'V_2 | gp'V_3 .. gp'V_6 | gp'V_7 -- lane group h of a row
    // takes logical sum 4 s + h in running sum s (kSplitRed)
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                                 const Row& w, double (&v)[2], double (&red)[4], double* slot, int) const {
        const int h = (int)((threadIdx.x >> 2) & 3u);
        double lr = w.v0 * u.c[0];
        lr = fma(w.v1, u.c[1], lr); lr = fma(w.v2, u.c[2], lr); lr = fma(w.v3, u.c[3], lr);
        lr = fma(w.v4, u.c[4], lr); lr = fma(w.v5, u.c[5], lr); lr = fma(w.v6, u.c[6], lr); lr = fma(w.v7, u.c[7], lr);
        const double ad = fma(w.ax, w.dx, lr);                                   // (A d)_i
        const double rr = INIT ? w.gx : fma(u.alpha, ad, w.gx);                  // :93
        const double gp = rr - accv[0];                                          // :97
        const double ag = w.ax * gp;                                             // the row-local part of A gp
        if (valid && owner) {
            if (slot) *slot = gp;
            else put(gout, o, gp);
            if (INIT) put(d, o, -gp);                                            // :62
        }
        if (valid && lead) {
            red[0] += gp * ((h == 0) ? rr : ((h == 1) ? gp : ((h == 2) ? ag : ad)));
            red[1] += (h == 0) ? w.dx * ad : gp * ((h == 1) ? w.v0 : ((h == 2) ? w.v1 : w.v2));
            red[2] += gp * ((h == 0) ? w.v3 : ((h == 1) ? w.v4 : ((h == 2) ? w.v5 : w.v6)));
            red[3] += (h == 0) ? gp * w.v7 : 0.0;
        }
        v[0] = valid ? gp : 0.0;
        v[1] = valid ? ag : 0.0;
    }
};

// ---- F for A = a0 I + diag(dg) + tridiagonal couplings: the same single pass ---------------------------------------------------------------
// (A gp)_i needs gp_{i+-1}, which other lanes, waves and workgroups are still computing -- but gp = rr - U t with t (= Utr) known BEFORE the
// pass and rr = g + alpha A d made of STORED vectors:
//   U'(A gp)  = U'(A rr) - (U'A U) t          -- the second product carries (A rr)_i = ax_i rr_i + q_i, with q (the neighbours' part) and A d
//                                                prepared by a vector kernel (TriPrepF), the m x m matrix M = U'A U formed once per solve
//                                                (tri_reduced_operator below);
//   gp'A gp   = rr'A rr - 2 t'(U'A rr) + t'M t  (post-op; the three terms are of the size of |A| |rr|^2, and rr differs from gp by what ONE
//                                                step alpha A d left in the range of U: no cancellation beyond a digit or two);
//   gp'A d, d'A d: row-local with A d stored.
// The row record has five doubles instead of three, the matrix stream is untouched, the residual is updated in place as in the diagonal form.
template <bool INIT>
struct PcgFuseTri {
    const double* rp;
    const double* g;
    double* gout;
    double* d;
    const double* ad;     // A d of the current direction       (TriPrepF; unused by INIT)
    const double* q;      // off-diagonal part of A rr          (TriPrepF)
    AOpD A;               // the diagonal of the operator
    const double* scal;
    const int64_t* istat;
    static constexpr bool kSplitRed = true;
    static constexpr bool kNoRowScale = true;
    struct Uni { double alpha; };
    struct Row { double gx, dx, ax, ad, q; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const { return Uni{INIT ? 0.0 : uniform_f64(ld_scal(scal + S_ALPHA))}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.gx = INIT ? at(rp, o) : at(g, o);
        w.dx = INIT ? 0.0 : at(d, o);
        w.ax = A.a0 + (A.dg ? at(A.dg, o) : 0.0);
        w.ad = INIT ? 0.0 : at(ad, o);
        w.q = at(q, o);
        return w;
    }
    static constexpr int kStageStreams = INIT ? 0 : 1;
    __device__ __forceinline__ double* stage_out(int) const { return gout; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[2], double (&red)[2]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    // reductions, logical order: rp'gp, gp'gp, rr'A rr, gp'A d | d'A d  (kSplitRed, as PcgFuseE)
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                                 const Row& w, double (&v)[2], double (&red)[2], double* slot, int) const {
        const int h = (int)((threadIdx.x >> 2) & 3u);
        const double rr = INIT ? w.gx : fma(u.alpha, w.ad, w.gx);                // :93 (TriPrepF formed the neighbours' rr by the same fma)
        const double gp = rr - accv[0];                                          // :97
        const double ar = fma(w.ax, rr, w.q);                                    // (A rr)_i
        if (valid && owner) {
            if (slot) *slot = gp;
            else put(gout, o, gp);
            if (INIT) put(d, o, -gp);                                            // :62
        }
        if (valid && lead) {
            red[0] += (h == 2) ? rr * ar : gp * ((h == 0) ? rr : ((h == 1) ? gp : w.ad));
            if (h == 0) red[1] += w.dx * w.ad;                                   // d'Ad
        }
        v[0] = valid ? gp : 0.0;
        v[1] = valid ? ar : 0.0;
    }
};

// The one post-op of the fused iteration, after the (all-reduced) sums T = [t1 (m); t2 (m); rp'gp; gp'gp; g'Ag; g'Ad; d'Ad]
// of kernel F are final.  End of iteration it (:98-111): beta, rg, nr, convergence / iteration limit.  Start of iteration
// it+1 (:72-91): d+'A d+ from the three direct sums, negative-curvature / rg exits, alpha.  Then (all threads)
// Utr = U'rp+ = t1 + alpha * t3,  t3 = U'(A d+) = -t2 + beta * t3   (first iteration: d0 = -g0, so t3 = -t2).
struct PcgPostF {
    const double* T;
    double* t3;
    double* Utr;
    double* scal;
    int64_t* istat;
    int m, init;
    HostMirror hm;
    // basis in factored form with a dense generator, U = A W (lfpsqp_basis.Z == NULL): kernel F streamed A and left the RAW sums
    // Traw = [A'gp (wm); A'(A_op gp) (wm); 5 scalars]; this kernel applies the small factor -- T = [W'raw1; W'raw2; scalars] first, and
    // uA = W Utr (the coefficients of F's next first product) last.  W == nullptr: materialised basis, nothing of this runs.
    const double* W = nullptr;    // wm x m, column-major (device)
    const double* Traw = nullptr;
    double* Tw = nullptr;         // == T, writable
    double* uA = nullptr;
    int wm = 0;
    int pre = 0;                  // the sums in T are final already (LFPSQP_PROJCG_START_PROJECTED: lfpsqp_tangent_step applied W'): no conversion of Traw
    // operator with a low-rank term V diag(sigma) V' (PcgFuseLR): k > 0; the sums carry V'gp behind the five scalars
    int k = 0;
    const double* UtV = nullptr;  // m x k, column-major (device): U'V
    const double* sigma = nullptr;   // k (device)
    double* vdraw = nullptr;      // kLRMax (device): V'd
    double* vdc = nullptr;        // kLRMax (device): sigma .* (V'd), what the kernel multiplies the rows of V with
    // tridiagonal operator (PcgFuseTri): M = U'A U (m x m, column-major, device); the pass left U'(A rr) in t2's place and rr'A rr in g'A g's
    const double* triM = nullptr;
    __device__ __forceinline__ int nsums() const { return k > 0 ? 5 + kLRMax : 5; }
};
// init = 2: RESUME after an iteration-limit exit (LFPSQP_PROJCG_RESUME): the end-of-iteration part already ran in the previous
// call; the limit has been raised, so the start-of-next-iteration part runs now (x was flushed by that call: alpha_prev = 0).
// y[j] = sum_k W[j*ld + k*sk] x[k], j < rows, k < cols, by the whole workgroup: `groups` threads share an output (the k-range dealt round
// robin), partial sums meet in LDS in a fixed order.  x read by ld_scal (written by an earlier kernel) or from LDS (xs != nullptr).
template <int NY>
__device__ __forceinline__ void small_wdot(const double* W, size_t ld, size_t sk, int rows, int cols, const double* const (&x)[NY], const double* xs,
                                           double (&y)[NY], double* scratch /* NY * blockDim.x */) {
    int rp = 32;
    while (rp < rows) rp <<= 1;
    const int groups = (int)blockDim.x / rp > 0 ? (int)blockDim.x / rp : 1;
    const int j = threadIdx.x % rp, g = threadIdx.x / rp;
    double acc[NY];
#pragma unroll
    for (int q = 0; q < NY; ++q) acc[q] = 0.0;
    if (j < rows && g < groups)
        for (int k = g; k < cols; k += groups) {
            const double w = W[(size_t)j * ld + (size_t)k * sk];
#pragma unroll
            for (int q = 0; q < NY; ++q) acc[q] = fma(w, xs ? xs[k] : ld_scal(x[q] + k), acc[q]);
        }
#pragma unroll
    for (int q = 0; q < NY; ++q) scratch[q * blockDim.x + threadIdx.x] = acc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NY; ++q) {
        double sum = 0.0;
        if (g == 0 && j < rows)
            for (int gg = 0; gg < groups; ++gg) sum += scratch[q * blockDim.x + gg * rp + j];
        y[q] = sum;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void pcg_post_kernel(PcgPostF u) {
    if (u.init == 2) {
        if (threadIdx.x == 0) u.istat[I_STATUS] = ST_RUNNING;
        __syncthreads();
    } else if (ld_stat(u.istat + I_STATUS) != ST_RUNNING) return;
    __shared__ double sh[2];
    __shared__ int go;
    __shared__ double s_utr[kOnepassMaxCols];
    __shared__ double s_part[2 * 1024];
    if (u.W && u.init != 2 && !u.pre) {                                // (a resumed solve finds T as the previous call converted it)
        // [t1; t2] = W' [raw1; raw2]: output j = column j of W (contiguous) against the raw sums.  m <= 1024 = blockDim: one output per thread slot
        const double* const xin[2] = {u.Traw, u.Traw + u.wm};
        double yo[2];
        small_wdot<2>(u.W, (size_t)u.wm, 1, u.m, u.wm, xin, nullptr, yo, s_part);
        if ((int)threadIdx.x < u.m) {
            int rp = 32;
            while (rp < u.m) rp <<= 1;
            if ((int)threadIdx.x / rp == 0) { u.Tw[threadIdx.x] = yo[0]; u.Tw[u.m + threadIdx.x] = yo[1]; }
        }
        if ((int)threadIdx.x < u.nsums()) u.Tw[2 * u.m + threadIdx.x] = ld_scal(u.Traw + 2 * u.wm + threadIdx.x);
        __threadfence();
        __syncthreads();
    }
    __shared__ double s_mu[kOnepassMaxCols];
    __shared__ double s_tri[2];
    if (u.triM) {
        // y = M t with t = Utr AS THE PASS USED IT (it is overwritten at the end of this kernel); M is symmetric: row j read down column j.
        // t and U'(A rr) go through LDS first, the two inner products t'(U'A rr) and t'y are summed by the whole workgroup in a fixed order
        // (one thread reading 2 m scalars from memory was 50 us of latency per iteration at m = 128)
        for (int j = threadIdx.x; j < u.m; j += blockDim.x) s_utr[j] = ld_scal(u.Utr + j);
        __syncthreads();
        double c1 = 0.0, c2 = 0.0;
        for (int j = threadIdx.x; j < u.m; j += blockDim.x) {
            double a = 0.0;
            for (int l = 0; l < u.m; ++l) a = fma(u.triM[(size_t)l * u.m + j], s_utr[l], a);
            s_mu[j] = a;
            c1 = fma(ld_scal(u.T + u.m + j), s_utr[j], c1);
            c2 = fma(a, s_utr[j], c2);
        }
        s_part[threadIdx.x] = c1;
        s_part[1024 + threadIdx.x] = c2;
        __syncthreads();
        for (int w = (int)blockDim.x >> 1; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                s_part[threadIdx.x] += s_part[threadIdx.x + w];
                s_part[1024 + threadIdx.x] += s_part[1024 + threadIdx.x + w];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) { s_tri[0] = s_part[0]; s_tri[1] = s_part[1024]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double* S = u.T + 2 * u.m;
        const double rpgp = ld_scal(S), gpgp = ld_scal(S + 1), gAd = ld_scal(S + 3), dAd = ld_scal(S + 4);
        double gAg = ld_scal(S + 2);
        if (u.triM) gAg = (gAg - 2.0 * s_tri[0]) + s_tri[1];          // gp'A gp = rr'A rr - 2 t'(U'A rr) + t'M t
        for (int j = 0; j < u.k; ++j) {                                 // + sum_j sigma_j (V'gp)_j^2
            const double vg = ld_scal(S + 5 + j);
            gAg = fma(ld_scal(u.sigma + j) * vg, vg, gAg);
        }
        double beta = 0.0, dAd_next = gAg;                              // d0 = -g0
        int64_t it = 0;
        int64_t st = ST_RUNNING;
        double nr = INFINITY;                                           // src/projcg.jl:69
        if (u.init == 1) {
            u.scal[S_RG] = gpgp;
            if (ld_stat(u.istat + I_MAXIT) <= 0) st = ST_MAXIT;
            u.hm.publish(st, 0, INFINITY);
        } else if (u.init == 2) {
            beta = ld_scal(u.scal + S_BETA);
            nr = ld_scal(u.scal + S_NR);
            it = ld_stat(u.istat + I_ITER);
            if (it >= ld_stat(u.istat + I_MAXIT)) st = ST_MAXIT;
            u.hm.publish(st, it, nr);
            dAd_next = fma(beta * beta, dAd, fma(-2.0 * beta, gAd, gAg));
        } else {
            beta = rpgp / ld_scal(u.scal + S_RG);                       // :98
            u.scal[S_BETA] = beta;
            u.scal[S_RG] = gpgp;
            nr = sqrt(gpgp);                                            // :103
            u.scal[S_NR] = nr;
            it = ld_stat(u.istat + I_ITER);
            if (nr < ld_scal(u.scal + S_TOL)) st = ST_CONVERGED;        // :107
            else if (it >= ld_stat(u.istat + I_MAXIT)) st = ST_MAXIT;   // :71
            u.hm.publish(st, it, nr);
            dAd_next = fma(beta * beta, dAd, fma(-2.0 * beta, gAd, gAg));
        }
        if (st == ST_RUNNING) {                                         // iteration it+1 starts (:72-91)
            u.istat[I_ITER] = it + 1;
            u.scal[S_DAD] = dAd_next;
            if (dAd_next <= 0.0) st = ST_NEGCURV;                       // :77
            else if (gpgp <= 0.0) st = ST_RG_BREAK;                     // :84-87 (r == g)
            else {
                u.scal[S_ALPHA_PREV] = (u.init == 2) ? 0.0 : ld_scal(u.scal + S_ALPHA);   // still owed to x (deferred :92)
                u.scal[S_ALPHA] = gpgp / dAd_next;                      // :91
            }
            if (st != ST_RUNNING) u.hm.publish(st, it + 1, nr);
        }
        if (st != ST_RUNNING) u.istat[I_STATUS] = st;
        go = (st == ST_RUNNING);
        sh[0] = gpgp / dAd_next;
        sh[1] = beta;
        if (go)
            for (int j = 0; j < u.k; ++j) {                             // V'd+ = beta V'd - V'gp  (d+ = beta d - gp; d0 = -g0)
                const double vg = ld_scal(S + 5 + j);
                const double vd = (u.init == 1) ? -vg : fma(beta, u.vdraw[j], -vg);
                u.vdraw[j] = vd;
                u.vdc[j] = ld_scal(u.sigma + j) * vd;
            }
    }
    __syncthreads();
    if (!go) return;
    const double alpha = sh[0], beta = sh[1];
    for (int j = threadIdx.x; j < u.m; j += blockDim.x) {
        const double t1 = ld_scal(u.T + j);
        double t2 = ld_scal(u.T + u.m + j);
        if (u.triM) t2 -= s_mu[j];                                      // U'(A gp) = U'(A rr) - M t
        for (int l = 0; l < u.k; ++l)                                   // U'(A gp) = U'(D gp) + (U'V) (sigma .* (V'gp))
            t2 = fma(u.UtV[(size_t)l * u.m + j], ld_scal(u.sigma + l) * ld_scal(u.T + 2 * u.m + 5 + l), t2);
        const double t3 = (u.init == 1) ? -t2 : fma(beta, u.t3[j], -t2);
        u.t3[j] = t3;
        const double ut = fma(alpha, t3, t1);
        u.Utr[j] = ut;
        if (u.W) s_utr[j] = ut;
    }
    if (u.W) {                                                          // uA = W Utr: output k = row k of W (stride wm between its entries)
        __syncthreads();
        const double* const xin[1] = {nullptr};
        double yo[1];
        small_wdot<1>(u.W, 1, (size_t)u.wm, u.wm, u.m, xin, s_utr, yo, s_part);
        int rp = 32;
        while (rp < u.wm) rp <<= 1;
        if ((int)threadIdx.x < u.wm && (int)threadIdx.x / rp == 0) u.uA[threadIdx.x] = yo[0];
    }
}
// the vector part of an iteration start in the fused flow: x += alpha_prev*d (deferred :92) ; d = beta*d - g (:99)
// LFPSQP_K1_NT: 1 = both vectors K1 writes leave through streaming stores (not kept dirty in the L2 for a write-back that would fall into
// the next F's read stream), 2 = x only, 0 = plain stores.  Same buffers, two builds (tools/gpu_k1nt_ab.sh): 1 makes K1 itself 3.6 % faster
// (0.0713 -> 0.0687 ms) and F 0.4-0.7 % faster on slow allocation pairs, nothing on fast ones; 2 changes K1 only.
#ifndef LFPSQP_K1_NT
#define LFPSQP_K1_NT 1
#endif
struct PcgDirG {
    double* d;
    const double* g;
    double* x;
    const double* scal;
    const int64_t* istat;
    int force;            // run regardless of the status, d only (negative-curvature exit needs d+)
    __device__ __forceinline__ bool skip() const { return !force && ld_stat(istat + I_STATUS) != ST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        double2 dd = ld2(d + i);
        if (!force) {
            const double alpha = ld_scal(scal + S_ALPHA_PREV);
            double2 xx = ld2(x + i);
            xx = make_double2(fma(alpha, dd.x, xx.x), fma(alpha, dd.y, xx.y));
            if (v1) { if (LFPSQP_K1_NT >= 1) st2_stream(x + i, xx); else st2(x + i, xx); }
            else if (v0) x[i] = xx.x;
        }
        const double beta = ld_scal(scal + S_BETA);
        const double2 gg = ld2(g + i);
        dd = make_double2(beta * dd.x - gg.x, beta * dd.y - gg.y);
        if (v1) { if (LFPSQP_K1_NT == 1) st2_stream(d + i, dd); else st2(d + i, dd); }
        else if (v0) d[i] = dd.x;
    }
};

// ---- setup / teardown functors --------------------------------------------------
template <class AOP>
struct ResidualV {  // v = sgn*(A x - b), optionally stored   (:56-57 with sgn=+1, :115-116 with sgn=-1)
    const double* x;
    const double* b;
    double* out;  // may be null
    AOP A;
    double sgn;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 xx = ld2(x + r), bb = ld2(b + r);
        const double2 ax = A.apply(r, xx);
        const double2 rr = make_double2(sgn * (ax.x - bb.x), sgn * (ax.y - bb.y));
        if (out) {
            if (v1) st2(out + r, rr);
            else if (v0) out[r] = rr.x;
        }
        return make_double2(v0 ? rr.x : 0.0, v1 ? rr.y : 0.0);
    }
};
struct InitState {
    double* scal;
    int64_t* istat;
    double tol;
    int64_t maxit;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void run(double*) const {
        for (int k = 0; k < 16; ++k) scal[k] = 0.0;
        scal[S_TOL] = tol;
        scal[S_NR] = INFINITY;                    // src/projcg.jl:69
        istat[I_STATUS] = ST_RUNNING;
        istat[I_ITER] = 0;
        istat[I_MAXIT] = maxit;
    }
};
struct FlushXF {  // the x-update of the last completed iteration (:92)
    double* x;
    const double* d;
    const double* scal;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double alpha = ld_scal(scal + S_ALPHA);
        const double2 dd = ld2(d + i);
        double2 xx = ld2(x + i);
        xx = make_double2(fma(alpha, dd.x, xx.x), fma(alpha, dd.y, xx.y));
        if (v1) st2(x + i, xx);
        else if (v0) x[i] = xx.x;
    }
};
struct SumSqF {
    const double* x;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(x + i);
        double s = 0.0;
        if (v0) s = a.x * a.x;
        if (v1) s = fma(a.y, a.y, s);
        red[0] += s;
    }
};
struct NormalizeIntoF {  // x = d / sqrt(dd)    (:79)
    double* x;
    const double* d;
    const double* dd;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double nrm = sqrt(ld_scal(dd));
        const double2 a = ld2(d + i);
        const double2 o = make_double2(a.x / nrm, a.y / nrm);
        if (v1) st2(x + i, o);
        else if (v0) x[i] = o.x;
    }
};

// ---- stacked (bound-constrained) variants: Q = [[diag Dx; diag Dy], [sx.*Z; sy.*Z]] -------
// One workgroup row r of the N x M matrix Z serves BOTH halves of the 2N-vectors, so the
// whole projection costs one pass over Z where the reference streams a 2N x M factor.
template <class AOP>
struct PcgStepVS {
    PcgStepV<AOP> p;
    StackD k;
    __device__ __forceinline__ bool skip() const { return p.skip(); }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 rx = p.load(r, v0, v1);                 // x-half of rp (recomputed, not stored)
        const double2 ry = p.load(r + k.hs, v0, v1);          // y-half
        const double2 ax = ld2(k.sx + r), ay = ld2(k.sy + r);
        return make_double2(ax.x * rx.x + ay.x * ry.x, ax.y * rx.y + ay.y * ry.y);
    }
};
template <class AOP>
struct ResidualVS {
    ResidualV<AOP> p;
    StackD k;
    double* wout;   // optional: the diagonal block Dx.*rx + Dy.*ry of Q'r (the first N entries of lambda)
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 rx = p.load(r, v0, v1);
        const double2 ry = p.load(r + k.hs, v0, v1);
        const double2 ax = ld2(k.sx + r), ay = ld2(k.sy + r);
        if (wout) {
            const double2 dx = ld2(k.Dx + r), dy = ld2(k.Dy + r);
            const double2 ww = make_double2(dx.x * rx.x + dy.x * ry.x, dx.y * rx.y + dy.y * ry.y);
            if (v1) st2(wout + r, ww);
            else if (v0) wout[r] = ww.x;
        }
        return make_double2(ax.x * rx.x + ay.x * ry.x, ax.y * rx.y + ay.y * ry.y);
    }
};
template <class AOP>
struct PcgProjES {
    PcgProjE<AOP> p;
    StackD k;
    __device__ __forceinline__ bool skip() const { return p.skip(); }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double* red) const {
        const double2 rx = p.init ? ld2(p.rp + r) : p.sv.rp_at(r);
        const double2 ry = p.init ? ld2(p.rp + r + k.hs) : p.sv.rp_at(r + k.hs);
        const double2 dx = ld2(k.Dx + r), dy = ld2(k.Dy + r), ax = ld2(k.sx + r), ay = ld2(k.sy + r);
        const double2 ww = make_double2(dx.x * rx.x + dy.x * ry.x, dx.y * rx.y + dy.y * ry.y);   // diagonal block of Q'rp
        // gp = rp - Q Q'rp on the two halves
        const double2 gx = make_double2(rx.x - fma(ax.x, acc.x, dx.x * ww.x), rx.y - fma(ax.y, acc.y, dx.y * ww.y));
        const double2 gy = make_double2(ry.x - fma(ay.x, acc.x, dy.x * ww.x), ry.y - fma(ay.y, acc.y, dy.y * ww.y));
        if (v1) { st2(p.g + r, gx); st2(p.g + r + k.hs, gy); }
        else if (v0) { p.g[r] = gx.x; p.g[r + k.hs] = gy.x; }
        if (p.init) {
            if (v1) { st2(p.d + r, make_double2(-gx.x, -gx.y)); st2(p.d + r + k.hs, make_double2(-gy.x, -gy.y)); }
            else if (v0) { p.d[r] = -gx.x; p.d[r + k.hs] = -gy.x; }
        }
        double s0 = 0.0, s1 = 0.0;
        if (v0) { s0 = rx.x * gx.x + ry.x * gy.x; s1 = gx.x * gx.x + gy.x * gy.x; }
        if (v1) { s0 += rx.y * gx.y + ry.y * gy.y; s1 += gx.y * gx.y + gy.y * gy.y; }
        red[0] += s0;
        red[1] += s1;
    }
};

}  // namespace lfpsqp

using namespace lfpsqp;

// A is either the device-resident diagonal form (A) or a callback (opf, with the work vector Av its products land in)
// ---- the basis in factored form on the nonzeros: U = A W, A = [S | up to 4 dense columns] ---------------------------------------
// U'v = W'(A'v): the producer's vector v is materialised (SpStoreV), A'v by the sparse product (+ a small dense GEMV-T for the extra
// columns), W' by one workgroup (sp_basis_small, sparse.hip).  U t = A (W t): u = W t by the same small kernel, then a row pass over the ELL
// entries feeds the consumer functor what the dense GEMV-N would have fed it (SpConsumeE).
struct SpPlainV {   // GEMV-T producer: the materialised vector itself
    const double* v;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 a = ld2(v + r);
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};
template <class V>
struct SpStoreV {
    V v;
    double* tmp;
    __device__ __forceinline__ bool skip() const { return v.skip(); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 a = v.load(i, v0, v1);
        if (v1) st2(tmp + i, a);
        else if (v0) tmp[i] = a.x;
    }
};
template <class E>
struct SpConsumeE {
    E e;
    EllRows R;                      // R.t = u (first S.m entries)
    const double* xcol;             // dense extra columns of A
    int64_t ldx;
    const double* ux;               // u[S.m ..]
    int nx;
    __device__ __forceinline__ bool skip() const { return e.skip(); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        double2 acc = R.acc(i);
        for (int j = 0; j < nx; ++j) {
            const double w = ld_scal(ux + j);
            const double2 c = ld2(xcol + (int64_t)j * ldx + i);
            acc.x = fma(c.x, w, acc.x);
            acc.y = fma(c.y, w, acc.y);
        }
        e.apply(i, acc, v0, v1, red);
    }
};
// the conditions projcg_impl puts on a basis without Z (DF: dense generator, fused iteration; SF: on the nonzeros of the sparse twin)
int lfpsqp_factored_basis_supported(const lfpsqp_ctx* ctx, const lfpsqp_mat* A, const lfpsqp_spmat* SA, int* yes) {
    if (!ctx || !A || !yes) return LFPSQP_ERR_ARG;
    if (SA)
        *yes = (A->m >= 1 && A->m <= kOnepassMaxCols && SA->n == A->n && SA->m >= 1 && A->m >= SA->m && A->m - SA->m <= 4) ? 1 : 0;
    else
        *yes = (A->m <= kOnepassMaxCols && onepass_cw(ctx, (int)A->m, A->ld, A->n) != 0) ? 1 : 0;
    return 0;
}

// ---- lfpsqp_projcg_tridiag: the reduced operator M = Z'A Z of a tridiagonal A, once per solve ----------------------------------------------------
// With s_i = sign(off_i) and R_i = Z_i + s_i Z_{i+1} (rows of Z):  off_i (Z_i'Z_{i+1} + Z_{i+1}'Z_i) = |off_i| (R_i'R_i - Z_i'Z_i - Z_{i+1}'Z_{i+1}), so
//     Z'A Z = R' diag(|off|) R + Z' diag(c) Z,    c_i = a0 + dg_i - |off_i| - |off_{i-1}|,
// i.e. weighted Gram matrices on the matrix cores -- the kernel of the tangent set-up, which forms the rows of R in registers on their way to
// LDS (gram_kernel SHIFT): two passes over Z, no scratch matrix.  The first term is a sum of squares whatever the signs of the couplings (for a
// discrete Laplacian, off = -1, it is the whole of M: no cancellation); c is non-negative for a diagonally dominant A, otherwise its negative
// part costs a third pass.
__global__ __launch_bounds__(256) void tri_weights_kernel(TriD A, double* __restrict__ wabs, double* __restrict__ sgn, double* __restrict__ cpos, double* __restrict__ cneg,
                                                          int64_t npad, double* __restrict__ anyneg) {
    bool neg = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += (int64_t)gridDim.x * 256) {
        double wa = 0.0, sg = 1.0, cp = 0.0, cn = 0.0;
        if (i < A.n) {
            const double o = (i + 1 < A.n) ? A.off[i] : 0.0;
            wa = fabs(o);
            sg = (o < 0.0) ? -1.0 : 1.0;
            const double wm = (i > 0) ? fabs(A.off[i - 1]) : 0.0;
            const double c = A.a0 + (A.dg ? A.dg[i] : 0.0) - wa - wm;
            if (c >= 0.0) cp = c;
            else { cn = -c; neg = true; }
            if (c != c) cp = c;                               // (NaN data: let the Gram pass report it)
        }
        wabs[i] = wa; sgn[i] = sg; cpos[i] = cp; cneg[i] = cn;
    }
    if (neg) *anyneg = 1.0;
}
static int ensure_tri(lfpsqp_ctx* ctx, size_t doubles) {
    if (doubles <= ctx->tri_cap) return 0;
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_tri) LF_HIP(ctx, hipFree(ctx->d_tri));
    ctx->d_tri = nullptr;
    ctx->tri_cap = 0;
    LF_HIP(ctx, hipMalloc((void**)&ctx->d_tri, doubles * sizeof(double)));
    ctx->tri_cap = doubles;
    return 0;
}
// Mh (m x m, column-major, host) = U'A U for U = Z[:, :mc] (W == nullptr, m == mc) or U = Z[:, :mc] W (W: mc x m, host)
static int tri_reduced_operator(lfpsqp_ctx* ctx, const lfpsqp_mat* Z, int mc, const TriD& A, const double* W, int m, std::vector<double>& Mh) {
    const int64_t n = A.n, npad = round_up((n > 0 ? n : 1) + 1, kPadRows);
    LF_TRY(ensure_tri(ctx, 4 * (size_t)npad + 8));
    double* wabs = ctx->d_tri;
    double* sgn = wabs + npad;
    double* cpos = sgn + npad;
    double* cneg = cpos + npad;
    double* anyneg = cneg + npad;
    LF_HIP(ctx, hipMemsetAsync(anyneg, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(tri_weights_kernel, dim3((int)std::min<int64_t>((npad + 255) / 256, 4096)), dim3(256), 0, ctx->stream, A, wabs, sgn, cpos, cneg, npad, anyneg);
    LF_LAUNCH_CHECK(ctx);
    double hneg = 0.0;
    LF_HIP(ctx, hipMemcpyAsync(&hneg, anyneg, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const lfpsqp_mat Zp = Z->plain();
    std::vector<double> G, G2((size_t)mc * mc);
    LF_TRY(gram_shifted(ctx, &Zp, mc, wabs, sgn, G));
    lfpsqp_vec wv;
    wv.n = n; wv.cap = npad;
    wv.p = cpos;
    LF_TRY(lfpsqp_gram(ctx, &Zp, mc, &wv, G2.data()));
    for (size_t k = 0; k < G.size(); ++k) G[k] += G2[k];
    if (hneg != 0.0) {
        wv.p = cneg;
        LF_TRY(lfpsqp_gram(ctx, &Zp, mc, &wv, G2.data()));
        for (size_t k = 0; k < G.size(); ++k) G[k] -= G2[k];
    }
    if (!W) { Mh = G; return 0; }
    // M = W' G W
    std::vector<double> GW((size_t)mc * m);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < mc; ++i) {
            double a = 0.0;
            for (int k = 0; k < mc; ++k) a = fma(G[(size_t)k * mc + i], W[(size_t)j * mc + k], a);      // G symmetric: row i down column i
            GW[(size_t)j * mc + i] = a;
        }
    Mh.assign((size_t)m * m, 0.0);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i <= j; ++i) {
            double a = 0.0;
            for (int k = 0; k < mc; ++k) a = fma(W[(size_t)i * mc + k], GW[(size_t)j * mc + k], a);
            Mh[(size_t)j * m + i] = Mh[(size_t)i * m + j] = a;
        }
    return 0;
}

static int projcg_impl(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_diag_op* A, lfpsqp_opfun opf, void* ouser,
                       lfpsqp_vec* Av, const lfpsqp_basis* U, const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit,
                       int64_t n_global, int flags, const lfpsqp_projcg_work* work, int64_t* iters, double* nr,
                       const lfpsqp_lowrank_op* LRop = nullptr, const lfpsqp_tridiag_op* TRop = nullptr) {
    const lfpsqp_diag_op no_diag = {0.0, nullptr};
    lfpsqp_diag_op lr_diag = {0.0, nullptr};
    if (LRop) { lr_diag.a0 = LRop->a0; lr_diag.dg = LRop->dg; A = &lr_diag; }
    if (TRop) { lr_diag.a0 = TRop->a0; lr_diag.dg = TRop->dg; A = &lr_diag; }
    if (opf) A = &no_diag;
    LF_ARG(ctx, ctx && x && A && U && b && work && iters && nr);
    // (a tridiagonal operator may start from the state lfpsqp_tangent_step left -- LFPSQP_PROJCG_START_GIVEN: r0 and U'r0 do not involve A -- but not
    // from its folded initial projection, whose sums were formed with the diagonal alone)
    LF_ARG(ctx, !(opf || TRop) || (Av && Av->n == b->n && !(flags & (LFPSQP_PROJCG_RESUME | (TRop ? 0 : LFPSQP_PROJCG_START_GIVEN) | LFPSQP_PROJCG_START_PROJECTED))));
    LF_ARG(ctx, !((flags & LFPSQP_PROJCG_RESUME) && (flags & (LFPSQP_PROJCG_START_GIVEN | LFPSQP_PROJCG_START_PROJECTED))));
    LF_ARG(ctx, !((flags & LFPSQP_PROJCG_START_GIVEN) && (flags & LFPSQP_PROJCG_START_PROJECTED)));
    LF_ARG(ctx, work->g && work->d && work->rp && work->Utr);
    const lfpsqp_ctx::ProjcgResume rs = ctx->pcg_resume;      // (taken before this call's own workspace requests invalidate it)
    const bool stacked = U->Dx != nullptr;
    const int64_t nv = b->n;                       // length of the n-vectors (hs + N when stacked)
    const int m = (int)U->ncols;
    LF_ARG(ctx, x->n == nv && work->g->n == nv && work->d->n == nv && work->rp->n == nv);
    // basis in factored form with a DENSE generator (U->Z == NULL, U = [sx; sy] .* (A W)): the fused iteration streams A (FINDINGS.md 5.3)
    const bool DF = m > 0 && !U->Z && U->A && U->W && !U->SA && m <= U->A->m && U->A->m <= kOnepassMaxCols;
    const bool SF = m > 0 && !U->Z && U->A && U->W && U->SA;       // ... or on the nonzeros of its sparse twin: no Z either
    LF_ARG(ctx, m >= 0 && (m == 0 || ((DF || SF || (U->Z && m <= U->Z->m)) && work->Utr->n >= m)));
    LF_ARG(ctx, !A->dg || A->dg->n == nv);
    LF_ARG(ctx, n_global >= (stacked ? 0 : nv));
    int64_t N = nv, hs = 0;                        // rows of Z
    if (stacked) {
        LF_ARG(ctx, U->Dy && U->sx && U->sy);
        N = U->Dx->n;
        hs = lfpsqp_half_stride(N);
        LF_ARG(ctx, nv == hs + N && U->Dy->n == N && U->sx->n == N && U->sy->n == N && (m == 0 || ((DF || SF) ? U->A->n : U->Z->n) == N));
        if (c) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "stacked basis with c != 0 (never used by optimize, src/optimize.jl:368)");
        LF_ARG(ctx, !(flags & LFPSQP_PROJCG_WANT_LAMBDA) || (lambda && lambda->n >= N + m));
    } else {
        LF_ARG(ctx, m == 0 || ((DF || SF) ? U->A->n : U->Z->n) == nv);
        LF_ARG(ctx, !c || c->n >= m);
        LF_ARG(ctx, !(flags & LFPSQP_PROJCG_WANT_LAMBDA) || (lambda && lambda->n >= m));
    }

    const AOpD Ad{A->a0, A->dg ? A->dg->p : nullptr};
    double* scal = ctx->scal;
    int64_t* istat = ctx->istat;
    double* g = work->g->p;
    double* d = work->d->p;
    double* rp = work->rp->p;
    double* Utr = work->Utr->p;
    const lfpsqp_mat* Z = m > 0 ? (DF ? U->A : U->Z) : nullptr;      // the matrix the kernels stream
    const int mc = DF ? (int)U->A->m : m;                            // ... and how many of its columns
    const StackD sk = stacked ? StackD{hs, U->Dx->p, U->Dy->p, U->sx->p, U->sy->p} : StackD{0, nullptr, nullptr, nullptr, nullptr};
    // loop bound min(maxit, n + m) with the reference's n = length(b), m = length(c)   (src/projcg.jl:43-44,71)
    const int64_t m_ref = stacked ? n_global / 2 + m : m;
    int64_t maxit_eff = maxit < n_global + m_ref ? maxit : n_global + m_ref;
    if (maxit_eff < 0) maxit_eff = 0;

    // basis in factored form on the nonzeros (plain basis, generator and its sparse twin known)?
    const lfpsqp_spmat* SA = nullptr;
    int wm = 0;
    double *dWs = nullptr, *tA = nullptr, *uA = nullptr;
    double* tmpN = rp;                             // scratch for the materialised producer vector: rp, or (stacked: rp holds both halves) a buffer
    if (m > 0 && m <= kOnepassMaxCols && U->SA && U->A && plain_mat(U->A) && U->W && U->SA->n == N && U->A->n == N && U->SA->m >= 1 && U->A->m >= U->SA->m &&
        U->A->m - U->SA->m <= 4 && !(c && m > 0)) {
        SA = U->SA;
        wm = (int)U->A->m;
        // (the scratch n-vector is read by whole tiles, like every n-vector of the library: padded to kPadRows, SpPlainV::load)
        LF_TRY(ensure_small(ctx, (size_t)wm * m + 2 * (size_t)wm + 64 + (stacked ? (size_t)round_up(N, kPadRows) + 2 : 0)));
        dWs = ctx->small;
        tA = dWs + (((size_t)wm * m + 1) & ~(size_t)1);
        uA = tA + ((wm + 1) & ~1);
        if (stacked) tmpN = uA + ((wm + 1) & ~1);
        LF_HIP(ctx, hipMemcpyAsync(dWs, U->W, sizeof(double) * (size_t)wm * m, hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));                 // U->W is caller-owned pageable memory
    }
    if (SF && !SA) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "basis in factored form on the nonzeros (Z == NULL, SA given): shape not covered; materialise Z");
    const int nxs = SA ? wm - (int)SA->m : 0;
    // t_out = U' v for the producer functor v (materialised in tmp): W' ([S | X]' v)
    auto sp_gemv_t = [&](auto vf, double* tmp, double* t_out, double* u_out) -> int {
        using V = decltype(vf);
        LF_TRY((run_vec<SpStoreV<V>, 0, NoPost>(ctx, N, SpStoreV<V>{vf, tmp}, 0u, nullptr, NoPost())));
        LF_TRY(spmv_t(ctx, SA, tmp, tA));
        if (nxs > 0) {
            lfpsqp_mat view = *U->A;
            view.p = U->A->p + (int64_t)SA->m * U->A->ld;
            view.m = nxs;
            LF_TRY(run_gemv_t(ctx, &view, nxs, N, SpPlainV{tmp}, tA + SA->m));
        }
        return sp_basis_small(ctx, dWs, wm, m, tA, t_out, u_out);
    };

    // fused iteration (one pass over U)?  Needs a tile shape for m columns and 32-bit lane offsets
    const bool fused = !SA && !opf && m > 0 && onepass_cw(ctx, mc, Z->ld, N) != 0;
    if (DF && !fused)
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "basis in factored form (Z == NULL) needs the fused iteration: diagonal operator, 4 .. 1024 generator "
                                                    "columns; materialise Z = A W for this shape");
    if (DF && c) LF_TRY(lfpsqp_q_gemv_n(ctx, U, 1.0, nullptr, c, 0.0, x));       // x = U c (:55), before the small factor settles in ctx->small
    double *T12 = nullptr, *t3 = nullptr, *Traw = nullptr, *uDF = nullptr, *dWf = nullptr;
    // operator with a low-rank term (lfpsqp_projcg_lowrank): fused iteration only
    const int kLR = LRop ? (int)LRop->k : 0;
    const int ns = kLR > 0 ? 5 + kLRMax : 5;             // scalar sums of a pass
    if (LRop) {
        LF_ARG(ctx, LRop->V && plain_mat(LRop->V) && kLR >= 1 && kLR <= kLRMax && LRop->V->m >= kLR && LRop->V->n == nv && (!LRop->dg || LRop->dg->n == nv));
        if (!fused || stacked || (flags & (LFPSQP_PROJCG_RESUME | LFPSQP_PROJCG_START_GIVEN | LFPSQP_PROJCG_START_PROJECTED)))
            return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "lfpsqp_projcg_lowrank: needs the one-pass iteration over a plain dense basis (4 .. 1024 columns), no RESUME / START_GIVEN");
    }
    // tridiagonal operator (lfpsqp_projcg_tridiag): fused iteration only; M = U'A U first (its Gram passes use the scratch areas reserved below)
    std::vector<double> triMh;
    TriD Atri{0.0, nullptr, nullptr, 0};
    if (TRop) {
        LF_ARG(ctx, TRop->off && TRop->off->n == nv && (!TRop->dg || TRop->dg->n == nv));
        if (!fused || stacked || !plain_mat(Z) || ctx->comm_active())
            return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "lfpsqp_projcg_tridiag: needs the one-pass iteration over a plain dense basis (4 .. 1024 columns, no matrix "
                                                        "view, no bounds) on a single rank (the couplings would cross the shard boundaries); use lfpsqp_projcg_op");
        Atri = TriD{TRop->a0, TRop->dg ? TRop->dg->p : nullptr, TRop->off->p, nv};
        LF_TRY(tri_reduced_operator(ctx, Z, mc, Atri, DF ? U->W : nullptr, m, triMh));
    }
    double* dTriM = nullptr;
    double *lrUtV = nullptr, *lrSig = nullptr, *lrVdraw = nullptr, *lrVdc = nullptr, *lrVtv = nullptr;
    if (fused) {
        LF_TRY(ensure_mvec(ctx, (size_t)3 * m + (DF ? 3 * (size_t)mc + 16 : 0) + 2 * kLRMax + 24 + (kLR > 0 ? (size_t)m * kLRMax + 4 * kLRMax + 8 : 0)));
        T12 = ctx->d_m;                                  // [t1 (m); t2 (m); rp'gp; gp'gp; g'Ag; g'Ad; d'Ad (; V'gp)]
        t3 = ctx->d_m + round_up(2 * m + ns, 2);
        double* tail = t3 + round_up(m, 2);
        if (DF) {
            Traw = tail;                                 // the kernel's raw sums over the generator's columns
            uDF = Traw + round_up(2 * mc + ns, 2);       // W Utr: the coefficients of the first product
            tail = uDF + round_up(mc, 2);
        }
        if (kLR > 0) {
            lrUtV = tail; lrSig = lrUtV + (size_t)m * kLRMax; lrVdraw = lrSig + kLRMax; lrVdc = lrVdraw + kLRMax; lrVtv = lrVdc + kLRMax;
        }
        if (DF || TRop) {
            const size_t wsz = DF ? (size_t)round_up((int64_t)mc * m, 2) : 0;
            LF_TRY(ensure_small(ctx, wsz + (TRop ? (size_t)m * m : 0) + 64));
            if (DF) {
                dWf = ctx->small;
                LF_HIP(ctx, hipMemcpyAsync(dWf, U->W, sizeof(double) * (size_t)mc * m, hipMemcpyHostToDevice, ctx->stream));
            }
            if (TRop) {
                dTriM = ctx->small + wsz;
                LF_HIP(ctx, hipMemcpyAsync(dTriM, triMh.data(), sizeof(double) * (size_t)m * m, hipMemcpyHostToDevice, ctx->stream));
            }
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));             // U->W is caller-owned pageable memory (and triMh a local)
        }
    }
    // t_out = U'v for the producer v: W'(A'v) (raw products parked in Traw); u_out (optional) = W t_out
    auto df_gemv_t = [&](auto vf, double* t_out, double* u_out) -> int {
        LF_TRY(run_gemv_t(ctx, Z, mc, N, vf, Traw));
        return sp_basis_small(ctx, dWf, mc, m, Traw, t_out, u_out);
    };
    const HostMirror hm{ctx->h_istat, ctx->h_scal};
    volatile int64_t* hstat = ctx->h_istat;
    hstat[0] = ST_RUNNING;
    hstat[1] = 0;
    for (int k = 0; k < kRing; ++k) hstat[kRingOff + k] = ST_RUNNING;
    // the user's operator: Av = A * v, queued on the context's stream (or synchronous) by the callback
    auto apply_op = [&](const lfpsqp_vec* v) -> int {
        const int rc = opf(ouser, v, Av);
        if (rc != 0) return set_err(ctx, LFPSQP_ERR_ARG, "operator callback returned %d", rc);
        return 0;
    };
    const AOpV Aop{(opf || TRop) ? Av->p : nullptr};
    auto residual_with = [&](auto aop, double sgn, double* store, double* t_out) -> int {
        using AOP = decltype(aop);
        const ResidualV<AOP> rv{x->p, b->p, store, aop, sgn};
        if (SA && stacked) return sp_gemv_t(ResidualVS<AOP>{rv, sk, store ? nullptr : lambda->p}, tmpN, t_out, store ? uA : nullptr);
        if (DF && stacked) return df_gemv_t(ResidualVS<AOP>{rv, sk, store ? nullptr : lambda->p}, t_out, store ? uDF : nullptr);
        if (DF) return df_gemv_t(rv, t_out, store ? uDF : nullptr);
        if (stacked) return run_gemv_t(ctx, Z, m, N, ResidualVS<AOP>{rv, sk, store ? nullptr : lambda->p}, t_out);
        if (SA) return sp_gemv_t(rv, tmpN, t_out, store ? uA : nullptr);    // (plain basis: rp doubles as the scratch vector)
        return run_gemv_t(ctx, Z, m, N, rv, t_out);
    };
    auto launch_residual = [&](double sgn, double* store, double* t_out) -> int {
        if (kLR > 0) {                                                        // V'x first (k columns of V: a thin pass), then the residual with it
            LF_TRY(run_gemv_t(ctx, LRop->V, kLR, N, SpPlainV{x->p}, lrVtv));
            return residual_with(AOpLR{Ad.a0, Ad.dg, LRop->V->p, LRop->V->ld, kLR, lrSig, lrVtv}, sgn, store, t_out);
        }
        if (TRop) {                                                           // Av = A x by the stencil kernel, then as a stored product
            LF_TRY((run_vec<TriMulF, 0, NoPost>(ctx, nv, TriMulF{Atri, x->p, Av->p, nullptr}, 0u, nullptr, NoPost())));
            return residual_with(Aop, sgn, store, t_out);
        }
        if (!opf) return residual_with(Ad, sgn, store, t_out);
        LF_TRY(apply_op(x));                                                  // Av = A x
        return residual_with(Aop, sgn, store, t_out);
    };
    auto k2_with = [&](auto aop) -> int {
        using AOP = decltype(aop);
        const PcgStepV<AOP> sv{d, g, aop, scal, istat};
        if (SA && stacked) return sp_gemv_t(PcgStepVS<AOP>{sv, sk}, tmpN, Utr, uA);
        if (stacked) return run_gemv_t(ctx, Z, m, N, PcgStepVS<AOP>{sv, sk}, Utr, 1);
        if (SA) return sp_gemv_t(sv, tmpN, Utr, uA);
        return run_gemv_t(ctx, Z, m, N, sv, Utr, 1);
    };
    auto launch_k2 = [&]() -> int { return opf ? k2_with(Aop) : k2_with(Ad); };
    auto k3_with = [&](auto aop, int init) -> int {
        using AOP = decltype(aop);
        const PcgProjE<AOP> pe{rp, g, d, istat, init, PcgStepV<AOP>{d, g, aop, scal, istat}};
        const PcgPost3 post{scal, istat, init, hm, scal + S_RPGP};
        if (SA && stacked) {                                              // u = W Utr was formed together with Utr (sp_gemv_t)
            using E = PcgProjES<AOP>;
            const SpConsumeE<E> ce{E{pe, sk}, ell_rows(SA, uA), U->A->p + (int64_t)SA->m * U->A->ld, U->A->ld, uA + SA->m, nxs};
            return run_vec<SpConsumeE<E>, 2, PcgPost3>(ctx, N, ce, 0u, scal + S_RPGP, post, init ? -1 : 2);
        }
        if (stacked) return run_gemv_n<PcgProjES<AOP>, 2, PcgPost3>(ctx, Z, m, N, Utr, PcgProjES<AOP>{pe, sk}, scal + S_RPGP, post, init ? -1 : 2);
        if (SA) {
            using E = PcgProjE<AOP>;
            const SpConsumeE<E> ce{pe, ell_rows(SA, uA), U->A->p + (int64_t)SA->m * U->A->ld, U->A->ld, uA + SA->m, nxs};
            return run_vec<SpConsumeE<E>, 2, PcgPost3>(ctx, nv, ce, 0u, scal + S_RPGP, post, init ? -1 : 2);
        }
        return run_gemv_n<PcgProjE<AOP>, 2, PcgPost3>(ctx, Z, m, N, Utr, pe, scal + S_RPGP, post, init ? -1 : 2);
    };
    auto launch_k3 = [&](int init) -> int { return opf ? k3_with(Aop, init) : k3_with(Ad, init); };

    // The fused kernel reads the residual of a row a tile ahead and stores the projected one a tile later.  On a box in the slow
    // state of FINDINGS.md §6, removing EITHER that load or that store of the same line made the kernel 13 % faster (1.95 -> 1.70 ms),
    // which suggested alternating two buffers between iterations (work->g and work->rp, free after the initial projection) so
    // that the kernel never stores to lines it has just loaded.  Measured on a fast-state box that costs 4-5 % (1.76 against
    // 1.70 ms: the store no longer hits a line the L2 already holds), so it is OFF by default; lfpsqp_ctx_set_residual_buffers(ctx, 1)
    // (or LFPSQP_GPING=1) turns it on (same bits either way).  gcur = the buffer holding the current g.
    const bool kPing = ctx->tune_gping == 1;
    // (the tridiagonal iteration keeps a vector of its own in rp after the initial projection: no alternation there)
    double* gbuf[2] = {g, (fused && kPing && !TRop) ? rp : g};
    int gcur = 0;
    auto launch_fused = [&](int init) -> int {
        const int slot = init ? -1 : 3;
        const double* gin = gbuf[gcur];
        double* gout = init ? gbuf[0] : gbuf[gcur ^ 1];
        const double* tin = DF ? uDF : Utr;               // coefficients of the first product over the streamed matrix's mc columns
        double* Tout = DF ? Traw : T12;
        // tridiagonal: the neighbours' contributions first (a vector kernel), into Av while rp still holds the initial residual, into rp afterwards
        if (TRop && init) {
            LF_TRY((run_vec<TriPrepF<true>, 0, NoPost>(ctx, nv, TriPrepF<true>{Atri, rp, nullptr, nullptr, Av->p, scal, istat}, 0u, nullptr, NoPost())));
            LF_TRY((run_onepass<PcgFuseTri<true>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseTri<true>{rp, gin, gout, d, nullptr, Av->p, Ad, scal, istat}, Tout, slot)));
        } else if (TRop) {
            LF_TRY((run_vec<TriPrepF<false>, 0, NoPost>(ctx, nv, TriPrepF<false>{Atri, gin, d, Av->p, rp, scal, istat}, 0u, nullptr, NoPost())));
            LF_TRY((run_onepass<PcgFuseTri<false>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseTri<false>{rp, gin, gout, d, Av->p, rp, Ad, scal, istat}, Tout, slot)));
        }
        else if (kLR > 0 && init) LF_TRY((run_onepass<PcgFuseLR<true>, 2, 5 + kLRMax>(ctx, Z, mc, mc, N, tin, PcgFuseLR<true>{rp, gin, gout, d, Ad, scal, istat, LRop->V->p, LRop->V->ld, kLR, lrVdc}, Tout, slot)));
        else if (kLR > 0) LF_TRY((run_onepass<PcgFuseLR<false>, 2, 5 + kLRMax>(ctx, Z, mc, mc, N, tin, PcgFuseLR<false>{rp, gin, gout, d, Ad, scal, istat, LRop->V->p, LRop->V->ld, kLR, lrVdc}, Tout, slot)));
        else if (stacked && init) LF_TRY((run_onepass<PcgFuseE<true, true>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseE<true, true>{rp, gin, gout, d, Ad, scal, istat, sk}, Tout, slot)));
        else if (stacked) LF_TRY((run_onepass<PcgFuseE<true, false>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseE<true, false>{rp, gin, gout, d, Ad, scal, istat, sk}, Tout, slot)));
        else if (init) LF_TRY((run_onepass<PcgFuseE<false, true>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseE<false, true>{rp, gin, gout, d, Ad, scal, istat, sk}, Tout, slot)));
        else LF_TRY((run_onepass<PcgFuseE<false, false>, 2, 5>(ctx, Z, mc, mc, N, tin, PcgFuseE<false, false>{rp, gin, gout, d, Ad, scal, istat, sk}, Tout, slot)));
        if (!init) gcur ^= 1;
        PcgPostF pf{T12, t3, Utr, scal, istat, m, init, hm, dWf, Traw, T12, uDF, DF ? mc : 0};
        if (kLR > 0) { pf.k = kLR; pf.UtV = lrUtV; pf.sigma = lrSig; pf.vdraw = lrVdraw; pf.vdc = lrVdc; }
        pf.triM = dTriM;
        hipLaunchKernelGGL(pcg_post_kernel, dim3(1), dim3(DF ? 1024 : 256), 0, ctx->stream, pf);
        LF_LAUNCH_CHECK(ctx);
        return 0;
    };

    // LFPSQP_PROJCG_RESUME: carry on from the state the previous call left when it stopped at its iteration limit
    const bool resume = (flags & LFPSQP_PROJCG_RESUME) != 0;
    int64_t it_base = 0;                              // device iteration number = it_base + host iteration + 1
    if (resume) {
        if (!fused || !rs.valid || rs.x != x->p || rs.g != g || rs.d != d || rs.Z != Z->p || rs.m != m || rs.nv != nv || rs.dg != Ad.dg ||
            rs.a0 != Ad.a0 || rs.b != b->p || rs.n_global != n_global || rs.epoch != ctx->launch_epoch)
            return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "LFPSQP_PROJCG_RESUME: no resumable projcg state for these arguments "
                           "(needs the one-pass iteration, a previous call with the same x, A, U, b, work that stopped at its iteration limit, and no library call that queued device work in between)");
        it_base = rs.iters;
        gcur = (rs.gcur_is_rp && gbuf[1] == rp) ? 1 : 0;
        if (rs.gcur_is_rp != (gcur == 1)) return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "LFPSQP_PROJCG_RESUME: the buffer scheme changed between the calls");
        maxit_eff = (it_base + maxit < n_global + m_ref) ? it_base + maxit : n_global + m_ref;
        hstat[1] = it_base;
        const int64_t lim = maxit_eff;
        LF_HIP(ctx, hipMemcpyAsync(istat + I_MAXIT, &lim, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));            // (source on the stack)
        hipLaunchKernelGGL(pcg_post_kernel, dim3(1), dim3(DF ? 1024 : 256), 0, ctx->stream, PcgPostF{T12, t3, Utr, scal, istat, m, 2, hm, dWf, Traw, T12, uDF, DF ? mc : 0});
        LF_LAUNCH_CHECK(ctx);
    }
    ctx->pcg_resume.valid = false;
    if (!resume) {
        hipLaunchKernelGGL((post_kernel<InitState>), dim3(1), dim3(1), 0, ctx->stream, scal, InitState{scal, istat, tol, maxit_eff});
        LF_LAUNCH_CHECK(ctx);

        // x = U c (:55); c == NULL is the all-zero c of optimize
        if (c && m > 0) {
            if (!DF) LF_TRY(lfpsqp_gemv_n(ctx, Z, m, 1.0, c, 0.0, x));      // (factored basis: done above)
        } else {
            LF_TRY(lfpsqp_vec_fill(ctx, x, 0.0));
        }
        if (kLR > 0) {
            // sigma (host, NULL = ones) and U'V (m x k): one GEMV-T per column of V, once per solve; V'd starts at zero
            for (int j = 0; j < kLRMax; ++j) ctx->h_m[j] = (j < kLR) ? (LRop->sigma ? LRop->sigma[j] : 1.0) : 0.0;
            LF_HIP(ctx, hipMemcpyAsync(lrSig, ctx->h_m, sizeof(double) * kLRMax, hipMemcpyHostToDevice, ctx->stream));
            LF_HIP(ctx, hipStreamSynchronize(ctx->stream));            // (h_m is the context's shared staging block)
            LF_HIP(ctx, hipMemsetAsync(lrVdraw, 0, sizeof(double) * 3 * kLRMax, ctx->stream));       // V'd, sigma .* V'd, V'x
            for (int l = 0; l < kLR; ++l) {
                const SpPlainV col{LRop->V->p + (int64_t)l * LRop->V->ld};
                if (DF) LF_TRY(df_gemv_t(col, lrUtV + (size_t)l * m, nullptr));
                else LF_TRY(run_gemv_t(ctx, Z, m, N, col, lrUtV + (size_t)l * m));
            }
        }
        // r = A x - b (kept in rp), Utr = U' r, g = r - U Utr, d = -g, rg = g'g   (:56-62)
        if (flags & LFPSQP_PROJCG_START_PROJECTED) {
            // lfpsqp_tangent_step (LFPSQP_TANGENT_INIT_PROJCG) has already made the initial projection in its own pass: g = g0 and d = -g0 are in
            // `work`, the sums [U'g0; U'(A g0); r0'g0; g0'g0; g0'A g0; 0; 0] wait behind the generator's column count in work->Utr
            if (!(DF && fused && !c && kLR == 0) || work->Utr->n < (int64_t)mc + 2 * m + 5)
                return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "LFPSQP_PROJCG_START_PROJECTED: basis in factored form over a dense generator, c == NULL, work->Utr of >= A.m + 2 ncols + 5 entries");
            LF_HIP(ctx, hipMemcpyAsync(T12, Utr + mc, sizeof(double) * (2 * (size_t)m + 5), hipMemcpyDeviceToDevice, ctx->stream));
            PcgPostF pf{T12, t3, Utr, scal, istat, m, 1, hm, dWf, Traw, T12, uDF, mc};
            pf.pre = 1;
            hipLaunchKernelGGL(pcg_post_kernel, dim3(1), dim3(1024), 0, ctx->stream, pf);
            LF_LAUNCH_CHECK(ctx);
        } else if (flags & LFPSQP_PROJCG_START_GIVEN) {
            // the caller's previous pass left r0 = -b in rp and U'r0 in Utr (lfpsqp_tangent_step): no residual pass; the first product of the
            // initial projection needs its coefficients over the generator's columns, W Utr
            if (!(DF && fused && !c))
                return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "LFPSQP_PROJCG_START_GIVEN: basis in factored form over a dense generator, c == NULL");
            LF_TRY(factored_w_times_t(ctx, dWf, mc, m, Utr, uDF));
        } else {
            LF_TRY(launch_residual(1.0, rp, Utr));
        }
        if (flags & LFPSQP_PROJCG_START_PROJECTED) { /* the initial projection is done */ }
        else if (fused) LF_TRY(launch_fused(1));
        else LF_TRY(launch_k3(1));
    }

    int64_t it = 0;
    bool done = false;
    const int gcur_start = gcur;
    const int64_t it_end = maxit_eff - it_base;
    while (!done && it < it_end) {
        if (fused) {
            // one global reduction per iteration: the post-op of F has already done the exits and alpha of this iteration
            if (it > 0 || resume) LF_TRY((run_vec<PcgDirG, 0, NoPost>(ctx, nv, PcgDirG{d, gbuf[gcur], x->p, scal, istat, 0}, 0u, nullptr, NoPost(), 0)));
            LF_TRY(launch_fused(0));
        } else if (opf) {
            // generic operator: the direction update, then the user's product A d, then d'(A d); the two passes over U read A d
            if (it > 0) LF_TRY((run_vec<PcgDirX, 0, NoPost>(ctx, nv, PcgDirX{d, g, x->p, scal, istat}, 0u, nullptr, NoPost(), 0)));
            LF_TRY(apply_op(work->d));
            LF_TRY((run_vec<DotPairF, 1, PcgPost1>(ctx, nv, DotPairF{d, Av->p, istat}, 0u, scal + S_DAD, PcgPost1{scal, istat, hm})));
            LF_TRY(launch_k2());
            LF_TRY(launch_k3(0));
        } else {
            LF_TRY((run_vec<PcgDirF, 1, PcgPost1>(ctx, nv, PcgDirF{d, g, x->p, Ad, scal, istat, it == 0 ? 1 : 0}, 0u, scal + S_DAD,
                                                  PcgPost1{scal, istat, hm}, 0)));
            LF_TRY(launch_k2());
            LF_TRY(launch_k3(0));
        }
        // throttle: stay at most two iterations ahead of the GPU; stop on the status of iteration it-2
        // (device iteration number it-1), which is final once its event has completed -- see HostMirror
        LF_HIP(ctx, hipEventRecord(ctx->ev_slot[it & 3], ctx->stream));
        if (it >= 2) {
            LF_HIP(ctx, hipEventSynchronize(ctx->ev_slot[(it - 2) & 3]));
            if (hstat[kRingOff + ((it_base + it - 1) % kRing)] != ST_RUNNING) done = true;
        }
        ++it;
    }
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t status = hstat[0];
    *iters = hstat[1];
    *nr = *(volatile double*)ctx->h_scal;
    if (fused) {
        // which buffer holds g: the host toggled once per QUEUED iteration, the device once per EXECUTED one (launches behind the
        // exit are no-ops).  An iteration-limit / convergence exit comes after the kernel of iteration *iters, the exits at an
        // iteration start (negative curvature, rg <= 0) before it.
        const int64_t executed = ((status == ST_CONVERGED || status == ST_MAXIT) ? *iters : *iters - 1) - it_base;
        gcur = (gbuf[1] != gbuf[0]) ? (int)((gcur_start + (executed > 0 ? executed : 0)) & 1) : 0;
    }

    // the x-update of the last COMPLETED iteration is still pending (K1 of the next one would have applied it; in the fused
    // flow the exits of an iteration start are taken before its K1, so they leave it pending too)
    if ((status == ST_CONVERGED || status == ST_MAXIT || (fused && status == ST_RG_BREAK && *iters > 1)) && *iters > it_base)
        LF_TRY((run_vec<FlushXF, 0, NoPost>(ctx, nv, FlushXF{x->p, d, scal}, 0u, nullptr, NoPost())));
    if (fused && status == ST_MAXIT && *iters > 0 && kLR == 0 && !TRop)
        ctx->pcg_resume = lfpsqp_ctx::ProjcgResume{true, x->p, g, d, Z->p, m, nv, *iters, gcur == 1, Ad.dg, b->p, Ad.a0, n_global, ctx->launch_epoch};
    if (status == ST_NEGCURV) {   // :77-82
        if (fused && *iters > 1)  // d+ = beta d - g of the iteration that found the negative curvature was not formed yet
            LF_TRY((run_vec<PcgDirG, 0, NoPost>(ctx, nv, PcgDirG{d, gbuf[gcur], x->p, scal, istat, 1}, 0u, nullptr, NoPost())));
        LF_TRY((run_vec<SumSqF, 1, NoPost>(ctx, nv, SumSqF{d}, 0u, scal + S_DD, NoPost())));
        LF_TRY((run_vec<NormalizeIntoF, 0, NoPost>(ctx, nv, NormalizeIntoF{x->p, d, scal + S_DD}, 0u, nullptr, NoPost())));
        if (lambda) LF_TRY(lfpsqp_vec_fill(ctx, lambda, NAN));
        *nr = INFINITY;
    } else if (flags & LFPSQP_PROJCG_WANT_LAMBDA) {   // :115-118  lambda = Q'(b - A x): [w (N); t (m)] when stacked
        if (stacked) LF_TRY(launch_residual(-1.0, nullptr, lambda->p + N));
        else if (m > 0) LF_TRY(launch_residual(-1.0, nullptr, lambda->p));
    }
    if (ctx->profiling) {
        LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        prof_collect(ctx);
    }
    return 0;
}

// The placement probe: the fused kernel F itself, on zero vectors (alpha = 0, g = d = 0: every store writes the zero it loaded), so what is
// timed is exactly the access pattern whose speed depends on where the matrix and the vectors were allocated (FINDINGS.md 6).
int lfpsqp::placement_probe(lfpsqp_ctx* ctx, const lfpsqp_mat* M, int ncols, double* g, double* d, double* a, int reps, double* ms) {
    *ms = -1.0;
    if (!M || M->n <= 0 || ncols < 4 || ncols > M->m || onepass_cw(ctx, ncols, M->ld, M->n) == 0) return 0;
    LF_TRY(ensure_mvec(ctx, (size_t)3 * ncols + 24));               // (also invalidates any resumable projcg state: scal is rewritten below)
    double* T12 = ctx->d_m;
    double* Utr = ctx->d_m + round_up(2 * ncols + 5, 2);           // zeros: the first product vanishes
    LF_HIP(ctx, hipMemsetAsync(ctx->d_m, 0, sizeof(double) * ((size_t)3 * ncols + 24), ctx->stream));
    hipLaunchKernelGGL((post_kernel<InitState>), dim3(1), dim3(1), 0, ctx->stream, ctx->scal, InitState{ctx->scal, ctx->istat, 0.0, (int64_t)1 << 40});
    LF_LAUNCH_CHECK(ctx);
    const StackD sk{0, nullptr, nullptr, nullptr, nullptr};
    const PcgFuseE<false, false> f{g, g, g, d, AOpD{0.0, a}, ctx->scal, ctx->istat, sk};
    // The trial is LOCAL: no collective inside it.  How many candidates a rank tries depends on its own shard size and free memory, so a
    // trial that all-reduced its (meaningless) sums would leave the ranks with different numbers of collectives in flight.
    const Comm::Kind saved = ctx->comm.kind;
    ctx->comm.kind = Comm::NONE;
    int rc = 0;
    for (int k = 0; k < reps + 1 && rc == 0; ++k) {
        if (k == 1) rc = hipEventRecord(ctx->ev_t0, ctx->stream) == hipSuccess ? 0 : LFPSQP_ERR_HIP;
        if (rc == 0) rc = run_onepass<PcgFuseE<false, false>, 2, 5>(ctx, M, ncols, ncols, M->n, Utr, f, T12, -1);
    }
    ctx->comm.kind = saved;
    LF_TRY(rc);
    LF_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
    LF_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
    float t = 0.f;
    LF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev_t0, ctx->ev_t1));
    *ms = (double)t / reps;
    return 0;
}

extern "C" int lfpsqp_projcg(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_diag_op* A, const lfpsqp_basis* U,
                             const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit, int64_t n_global, int flags,
                             const lfpsqp_projcg_work* work, int64_t* iters, double* nr) {
    LF_RANGE("lfpsqp_projcg");
    LF_ARG(ctx, ctx && A);
    return projcg_impl(ctx, x, lambda, A, nullptr, nullptr, nullptr, U, b, c, tol, maxit, n_global, flags, work, iters, nr);
}

extern "C" int lfpsqp_projcg_lowrank(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_lowrank_op* A, const lfpsqp_basis* U,
                                     const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit, int64_t n_global, int flags,
                                     const lfpsqp_projcg_work* work, int64_t* iters, double* nr) {
    LF_RANGE("lfpsqp_projcg_lowrank");
    LF_ARG(ctx, ctx && A);
    if (A->k == 0) {
        const lfpsqp_diag_op dop{A->a0, A->dg};
        return projcg_impl(ctx, x, lambda, &dop, nullptr, nullptr, nullptr, U, b, c, tol, maxit, n_global, flags, work, iters, nr);
    }
    return projcg_impl(ctx, x, lambda, nullptr, nullptr, nullptr, nullptr, U, b, c, tol, maxit, n_global, flags, work, iters, nr, A);
}

extern "C" int lfpsqp_projcg_tridiag(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, const lfpsqp_tridiag_op* A, lfpsqp_vec* Av, const lfpsqp_basis* U,
                                     const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit, int64_t n_global, int flags,
                                     const lfpsqp_projcg_work* work, int64_t* iters, double* nr) {
    LF_RANGE("lfpsqp_projcg_tridiag");
    LF_ARG(ctx, ctx && A && Av);
    return projcg_impl(ctx, x, lambda, nullptr, nullptr, nullptr, Av, U, b, c, tol, maxit, n_global, flags, work, iters, nr, nullptr, A);
}

extern "C" int lfpsqp_tridiag_mul(lfpsqp_ctx* ctx, const lfpsqp_tridiag_op* A, const lfpsqp_vec* v, lfpsqp_vec* out) {
    LF_ARG(ctx, ctx && A && A->off && v && out && v != out && v->p != out->p && out->n == v->n && A->off->n == v->n && (!A->dg || A->dg->n == v->n));
    if (ctx->comm_active())        // (a rank sees its own rows only: the couplings across the shard boundaries would silently drop out)
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "lfpsqp_tridiag_mul: one rank only (no halo exchange between row shards)");
    return run_vec<TriMulF, 0, NoPost>(ctx, v->n, TriMulF{TriD{A->a0, A->dg ? A->dg->p : nullptr, A->off->p, v->n}, v->p, out->p, nullptr}, 0u, nullptr, NoPost());
}

extern "C" int lfpsqp_projcg_op(lfpsqp_ctx* ctx, lfpsqp_vec* x, lfpsqp_vec* lambda, lfpsqp_opfun A, void* user, lfpsqp_vec* Av,
                                const lfpsqp_basis* U, const lfpsqp_vec* b, const lfpsqp_vec* c, double tol, int64_t maxit,
                                int64_t n_global, int flags, const lfpsqp_projcg_work* work, int64_t* iters, double* nr) {
    LF_RANGE("lfpsqp_projcg_op");
    LF_ARG(ctx, ctx && A && Av);
    return projcg_impl(ctx, x, lambda, nullptr, A, user, Av, U, b, c, tol, maxit, n_global, flags, work, iters, nr);
}

extern "C" int lfpsqp_ctx_stream(lfpsqp_ctx* ctx, void** stream) {
    LF_ARG(ctx, ctx && stream);
    *stream = (void*)ctx->stream;
    return 0;
}
