// pcg! of the ProjPenalty retraction (reference src/retractions.jl:179-246) fused on the device:
//
//   P1  p = r + beta*p ; [w = Dx.*px + Dy.*py] ; partial J p          (gemv_t_kernel)  :209-221
//   P2  z = J'(J p) + mu*p ; partial p'z                              (gemv_n_kernel)  :220-227
//   P3  x += alpha*p ; r -= alpha*z ; partial r'r                     (vec_kernel)     :232-235
//
// z = M!(r) = r (no_precondition, the only preconditioner on the reference's live path, :374-375), so
// rho = r'r is the previous iteration's residual norm squared and costs nothing.  Scalars / status in device
// memory, published to the pinned host block.
//
// DEFAULT (4 <= m <= 1024): ONE pass over Jct per iteration, same construction as the fused projcg iteration
// (projcg.hip, onepass_kernel): with tmp = J p known, the kernel forms z = J'tmp + mu p row by row and, over the tile
// it still holds, accumulates u = J z and s = J r.  The next iteration's J p follows from linearity,
//     J p+ = J r+ + beta J p,   J r+ = J r - alpha J z = s - alpha u,
// so the separate pass for J p (P1) disappears after the first iteration:
//   F   p = r + beta*p (stored) ; z = J'tmp + mu*p (stored) ; partials p'z, J z, J r     (onepass_kernel)
//   P3  x += alpha*p ; r -= alpha*z ; partial r'r                                        (vec_kernel)
//   T   tmp = (s - alpha*u) + beta*tmp                                                   (m-vector kernel)
// = 8 n m + 80 n bytes instead of 16 n m + 88 n.  alpha, beta, rho and the stopping test are computed exactly as in
// the reference; s = J r is re-measured every iteration, so the recurrence for tmp does not drift.
// Stores inside the matrix stream are the expensive part of F (FINDINGS.md 5.2: 160 MB of writes cost as much as 1.4 GB of reads), so F
// stores ONE vector, z; the direction p = r + beta*p, which it forms in registers, is formed again -- by the same fma -- and stored by P3,
// a plain vector kernel that reads p and r anyway:
//   F   z = J'tmp + mu*(r + beta*p) (stored) ; partials p'z, J z, J r
//   P3  p = r + beta*p (stored) ; x += alpha*p ; r -= alpha*z ; partial r'r
#include <math.h>

#include "internal.h"
#include "sparse.h"

namespace lfpsqp {

enum { P_RHO = 16, P_BETA = 17, P_ALPHA = 18, P_PZ = 19, P_RR = 20, P_NRES = 21, P_TOL = 22 };
enum { IP_STATUS = 4, IP_ITER = 5, IP_MAXIT = 6 };
enum { PST_RUNNING = 0, PST_DONE = 1 };

constexpr int kPRing = 8, kPRingOff = 16;   // per-iteration status ring (see HostMirror in projcg.hip)
struct PcgHostMirror {
    int64_t* hstat;  // [IP_STATUS], [IP_ITER], [kPRingOff + (iter % kPRing)]
    __device__ __forceinline__ void publish(int64_t status, int64_t iter) const {
        __hip_atomic_store(hstat + IP_ITER, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hstat + kPRingOff + (iter % kPRing), status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hstat + IP_STATUS, status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

struct PInit {
    double* scal;
    int64_t* istat;
    double tol;
    int64_t maxit;
    PcgHostMirror hm;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void run(double*) const {   // after rr0 = r'r : first rho, beta (rho_prev = 1, :203,212-216)
        const double rr = ld_scal(scal + P_RR);
        scal[P_RHO] = rr;
        scal[P_BETA] = rr / 1.0;
        scal[P_NRES] = INFINITY;                            // :202
        scal[P_TOL] = tol;
        istat[IP_ITER] = 0;
        istat[IP_MAXIT] = maxit;
        const int64_t st = (maxit > 0) ? PST_RUNNING : PST_DONE;   // while norm_res > tol && i < maxiter, norm_res = Inf
        istat[IP_STATUS] = st;
        hm.publish(st, 0);
    }
};

struct P1V {  // p = r + beta p (stored); returns the new p
    double* p;
    const double* r;
    const double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ double2 load(int64_t i, bool v0, bool v1) const {
        const double beta = ld_scal(scal + P_BETA);
        const double2 rr = ld2(r + i), pp = ld2(p + i);
        const double2 o = make_double2(rr.x + beta * pp.x, rr.y + beta * pp.y);   // :217
        if (v1) st2(p + i, o);
        else if (v0) p[i] = o.x;
        return make_double2(v0 ? o.x : 0.0, v1 ? o.y : 0.0);
    }
};
struct PStack {
    int64_t hs;
    const double *Dx, *Dy, *sx, *sy;
    double* w;
};
struct P1VS {
    P1V b;
    PStack k;
    __device__ __forceinline__ bool skip() const { return b.skip(); }
    __device__ __forceinline__ double2 load(int64_t i, bool v0, bool v1) const {
        const double2 px = b.load(i, v0, v1), py = b.load(i + k.hs, v0, v1);
        const double2 dx = ld2(k.Dx + i), dy = ld2(k.Dy + i), ax = ld2(k.sx + i), ay = ld2(k.sy + i);
        const double2 ww = make_double2(dx.x * px.x + dy.x * py.x, dx.y * px.y + dy.y * py.y);
        if (v1) st2(k.w + i, ww);
        else if (v0) k.w[i] = ww.x;
        return make_double2(ax.x * px.x + ay.x * py.x, ax.y * px.y + ay.y * py.y);
    }
};

struct P2E {  // z = acc + mu p ; p'z
    const double* p;
    double* z;
    double mu;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, double2 acc, bool v0, bool v1, double* red) const {
        const double2 pp = ld2(p + i);
        const double2 zz = make_double2(fma(mu, pp.x, acc.x), fma(mu, pp.y, acc.y));   // :222  z = J'tmp + mu z, z = p
        if (v1) st2(z + i, zz);
        else if (v0) z[i] = zz.x;
        double s = 0.0;
        if (v0) s = pp.x * zz.x;
        if (v1) s = fma(pp.y, zz.y, s);
        red[0] += s;
    }
};
struct P2ES {
    P2E b;
    PStack k;
    __device__ __forceinline__ bool skip() const { return b.skip(); }
    __device__ __forceinline__ void apply(int64_t i, double2 acc, bool v0, bool v1, double* red) const {
        const double2 ww = ld2(k.w + i), dx = ld2(k.Dx + i), dy = ld2(k.Dy + i), ax = ld2(k.sx + i), ay = ld2(k.sy + i);
        b.apply(i, make_double2(fma(ax.x, acc.x, dx.x * ww.x), fma(ax.y, acc.y, dx.y * ww.y)), v0, v1, red);
        b.apply(i + k.hs, make_double2(fma(ay.x, acc.x, dy.x * ww.x), fma(ay.y, acc.y, dy.y * ww.y)), v0, v1, red);
    }
};
struct PPost2 {
    double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void run(double*) const { scal[P_ALPHA] = ld_scal(scal + P_RHO) / ld_scal(scal + P_PZ); }   // :227
};

struct P3F {  // [p = r + beta p ;] x += alpha p ; r -= alpha z ; r'r
    double* x;
    double* r;
    double* p;
    const double* z;
    const double* scal;
    const int64_t* istat;
    int pupd;             // the one-pass iteration left the direction update to this kernel (:217, the expression of PcgInnerE)
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double alpha = ld_scal(scal + P_ALPHA);
        double2 pp = ld2(p + i);
        const double2 zz = ld2(z + i);
        double2 xx = ld2(x + i), rr = ld2(r + i);
        if (pupd) {
            const double beta = ld_scal(scal + P_BETA);
            pp = make_double2(fma(beta, pp.x, rr.x), fma(beta, pp.y, rr.y));
            if (v1) st2(p + i, pp);
            else if (v0) p[i] = pp.x;
        }
        xx = make_double2(fma(alpha, pp.x, xx.x), fma(alpha, pp.y, xx.y));     // :232
        rr = make_double2(fma(-alpha, zz.x, rr.x), fma(-alpha, zz.y, rr.y));   // :233
        if (v1) { st2(x + i, xx); st2(r + i, rr); }
        else if (v0) { x[i] = xx.x; r[i] = rr.x; }
        double s = 0.0;
        if (v0) s = rr.x * rr.x;
        if (v1) s = fma(rr.y, rr.y, s);
        red[0] += s;
    }
};
struct PPost3 {
    double* scal;
    int64_t* istat;
    PcgHostMirror hm;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void run(double*) const {
        const double rr = ld_scal(scal + P_RR);
        const double nres = sqrt(rr);                       // :235
        scal[P_NRES] = nres;
        const int64_t it = ld_stat(istat + IP_ITER) + 1;    // :237
        istat[IP_ITER] = it;
        const double rho_prev = ld_scal(scal + P_RHO);      // next iteration: rho = z'r = r'r, beta = rho / rho_prev (:212-216)
        scal[P_RHO] = rr;
        scal[P_BETA] = rr / rho_prev;
        int64_t st = PST_RUNNING;
        if (!(nres > ld_scal(scal + P_TOL)) || it >= ld_stat(istat + IP_MAXIT)) st = PST_DONE;   // :207 (NaN stops the loop)
        if (st != PST_RUNNING) istat[IP_STATUS] = st;
        hm.publish(st, it);
    }
};

// ---- one-pass iteration ------------------------------------------------------------------------------------
template <bool ST>
struct PcgInnerE {
    const double* p;
    const double* r;
    double* z;
    double mu;
    const double* scal;
    const int64_t* istat;
    int first;          // first iteration: p (= r) was already formed by the P1 pass that computed tmp = J p
    PStack k;
    static constexpr bool kSplitRed = false;
    struct Uni { double beta; };
    struct Row { double px, rx, py, ry, Dx, Dy, sx, sy; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    static __device__ __forceinline__ void put(double* base, uint32_t o, double v) {
        *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + o) = v;
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const { return Uni{uniform_f64(ld_scal(scal + P_BETA))}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.px = at(p, o); w.rx = at(r, o);
        if (ST) {
            w.py = at(p + k.hs, o); w.ry = at(r + k.hs, o);
            w.Dx = at(k.Dx, o); w.Dy = at(k.Dy, o); w.sx = at(k.sx, o); w.sy = at(k.sy, o);
        } else {
            w.py = w.ry = w.Dx = w.Dy = w.sx = w.sy = 0.0;
        }
        return w;
    }
    // staged form (onepass_kernel STG): z (stacked: its two halves) waits in LDS and is stored in bursts
    static constexpr int kStageStreams = ST ? 2 : 1;
    __device__ __forceinline__ double* stage_out(int sv) const { return sv == 0 ? z : z + k.hs; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[2], double (&red)[1]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool, const Uni& u,
                                                 const Row& w, double (&v)[2], double (&red)[1], double* slot, int sstride) const {
        const double acc = accv[0];
        const bool st = valid && owner;
        if (!ST) {
            const double pn = first ? w.px : fma(u.beta, w.px, w.rx);                // :217 (stored by P3: same fma)
            const double zz = fma(mu, pn, acc);                                      // :222
            if (st) {
                if (slot) *slot = zz;
                else put(z, o, zz);
                red[0] += pn * zz;                                                   // :226
            }
            v[0] = valid ? zz : 0.0;
            v[1] = valid ? w.rx : 0.0;
        } else {
            const double pnx = first ? w.px : fma(u.beta, w.px, w.rx);
            const double pny = first ? w.py : fma(u.beta, w.py, w.ry);
            const double ww = w.Dx * pnx + w.Dy * pny;                               // diagonal block of J p
            const double zx = fma(mu, pnx, fma(w.sx, acc, w.Dx * ww));
            const double zy = fma(mu, pny, fma(w.sy, acc, w.Dy * ww));
            if (st) {
                if (slot) { slot[0] = zx; slot[sstride] = zy; }
                else { put(z, o, zx); put(z + k.hs, o, zy); }
                red[0] += pnx * zx + pny * zy;
            }
            v[0] = valid ? (w.sx * zx + w.sy * zy) : 0.0;                            // Z-block of J z
            v[1] = valid ? (w.sx * w.rx + w.sy * w.ry) : 0.0;                        // ... of J r
        }
    }
};
struct PPost2F {   // alpha = rho / p'z, p'z at the tail of the fused kernel's output
    double* scal;
    const int64_t* istat;
    const double* pz;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void run(double*) const { scal[P_ALPHA] = ld_scal(scal + P_RHO) / ld_scal(pz); }   // :227
};
// tmp = (s - alpha*u) + beta*tmp  (J p of the next iteration), from the previous fused kernel's [u ; s]
struct PcgTmp {
    const double* US;
    double* tmp;
    const double* scal;
    const int64_t* istat;
    int m;
};
__global__ __launch_bounds__(256) void pcg_tmp_kernel(PcgTmp u) {
    if (ld_stat(u.istat + IP_STATUS) != PST_RUNNING) return;
    const double alpha = ld_scal(u.scal + P_ALPHA), beta = ld_scal(u.scal + P_BETA);
    for (int j = threadIdx.x; j < u.m; j += 256) {
        const double jr = fma(-alpha, ld_scal(u.US + j), ld_scal(u.US + u.m + j));
        u.tmp[j] = fma(beta, u.tmp[j], jr);
    }
}
// ---- sparse operator (lfpsqp_basis.S): the two products of an iteration stream the nonzeros ----------------------------
//   P1s  p = r + beta*p (stored) [stacked: w = Dx.*px + Dy.*py, v = sx.*px + sy.*py]     (vec_kernel)
//        tmp = S'(p | v)                                                                   (spmv_t: chunked CSC, fixed order)
//   P2s  z = S*tmp [+ the diagonal terms] + mu*p ; partial p'z                             (vec_kernel over the ELL rows)
//   P3   as above
template <class P1>
struct P1SparseF {        // the producers of the dense kernels, run row by row; the vector handed to S' is stored for its gather
    P1 e;
    double* vz;           // stacked: where v goes (the x-half of z, free until P2s); plain: nullptr (p itself is gathered)
    __device__ __forceinline__ bool skip() const { return e.skip(); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 v = e.load(i, v0, v1);
        if (vz) {
            if (v1) st2(vz + i, v);
            else if (v0) vz[i] = v.x;
        }
    }
};
template <class P2>
struct P2SparseF {        // the consumers of the dense kernels fed from the ELL rows
    P2 e;
    EllRows E;
    __device__ __forceinline__ bool skip() const { return e.skip(); }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const { e.apply(i, E.acc(i), v0, v1, red); }
};

struct RRF {
    const double* r;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double2 a = ld2(r + i);
        double s = 0.0;
        if (v0) s = a.x * a.x;
        if (v1) s = fma(a.y, a.y, s);
        red[0] += s;
    }
};

// ---- pcg! with an EXACT preconditioner (lfpsqp_pcg_pre) -------------------------------------------------------------------------
// The operator of the inner solve is  A_f = D0 + E E'  with E = [Jct; 0] and D0 = mu I (plain) or, with bounds, mu I plus the 2 x 2
// blocks [a^2, ab; ab, b^2] of the inequality rows (a = Dx.*S, b = Dy.*S: src/inequality_helper.jl:215-271).  Its inverse is
//     M^-1 = D0^-1 - D0^-1 E K E' D0^-1,     K = (I + E' D0^-1 E)^-1  (m x m),
// and for the plain operator that IS the reference's proj_precondition! (src/retractions.jl:248-257: z = (r - U diag(s^2/(mu+s^2)) U'r)/mu
// with U = Jct W, i.e. K = mu W diag(s^2/(mu+s^2)) W').  The caller hands over K and the rows of D0^-1; an iteration of
// src/retractions.jl:207-238 is then three passes over Jct and one vector kernel -- and one iteration is what an exact M takes:
//   S   sP = Jct'(D0^-1 r)_x, measured DIRECTLY (gemv_t_kernel).  (A recurrence sP+ = sP - alpha Jct'(D0^-1 q)_x would save this pass, but r
//       shrinks by ten orders of magnitude in one iteration of an exact preconditioner and the recurrence cancels catastrophically: the
//       second iteration then gains one digit instead of ten -- measured: 6e-11 -> 5e-12 against the oracle's 5e-11 -> 9e-22.)
//   Z   z = M^-1 r:  first product Jct c1 with c1 = K sP; second product Jct' z_x (-> J z); partial z'r            :209-213   (onepass_kernel)
//   Q   q = A_f p for p = z + beta p:  first product Jct (J p) with J p = J z + beta J p; partial p'q               :216-227   (onepass_kernel)
//   P3  p = z + beta p (stored) ; x += alpha p ; r -= alpha q ; partial r'r                                          :217, :232-235
struct PrecScale {
    const double *i11, *i12, *i22;   // rows of D0^-1 = [i11 i12; i12 i22] (stacked operator); nullptr: plain operator, D0^-1 = inv_mu
    double inv_mu;
};
enum { PM_OUT1 = 0 };
template <bool ST>
struct PrecZE {
    const double* r;
    double* z;
    PrecScale ps;
    int64_t hs;
    const int64_t* istat;
    static constexpr bool kSplitRed = false;
    using Uni = NoUni;
    struct Row { double rx, ry, a11, a12, a22; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const { return Uni{}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.rx = at(r, o);
        if (ST) { w.ry = at(r + hs, o); w.a11 = at(ps.i11, o); w.a12 = at(ps.i12, o); w.a22 = at(ps.i22, o); }
        else { w.ry = w.a12 = w.a22 = 0.0; w.a11 = ps.inv_mu; }
        return w;
    }
    static constexpr int kStageStreams = ST ? 2 : 1;
    __device__ __forceinline__ double* stage_out(int sv) const { return sv == 0 ? z : z + hs; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[1], double (&red)[1]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool, const Uni&,
                                                 const Row& w, double (&v)[1], double (&red)[1], double* slot, int sstride) const {
        const double acc = accv[0];
        const bool st = valid && owner;
        const double ux = ST ? fma(w.a12, w.ry, w.a11 * w.rx) : w.a11 * w.rx;           // (D0^-1 r)_x
        const double zx = fma(-w.a11, acc, ux);                                          // z = D0^-1 r - D0^-1 E c1
        double zy = 0.0;
        if (ST) zy = fma(-w.a12, acc, fma(w.a22, w.ry, w.a12 * w.rx));
        if (st) {
            if (slot) { slot[0] = zx; if (ST) slot[sstride] = zy; }
            else { *reinterpret_cast<double*>(reinterpret_cast<char*>(z) + o) = zx; if (ST) *reinterpret_cast<double*>(reinterpret_cast<char*>(z + hs) + o) = zy; }
            red[0] += ST ? fma(zy, w.ry, zx * w.rx) : zx * w.rx;                         // rho = z'r   :213
        }
        v[0] = valid ? zx : 0.0;                                                         // -> J z  (fulljac's column block sees the x half)
    }
};
template <bool ST>
struct PrecQE {
    const double* z;
    const double* p;
    double* q;
    double mu;
    PrecScale ps;
    PStack k;            // fulljac's diagonal block (Dx.*S, Dy.*S); hs
    const double* scal;
    const int64_t* istat;
    static constexpr bool kSplitRed = false;
    struct Uni { double beta; };
    struct Row { double zx, zy, px, py, a, b, a11, a12; };
    static __device__ __forceinline__ double at(const double* base, uint32_t o) {
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + o);
    }
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ Uni uniform() const { return Uni{uniform_f64(ld_scal(scal + P_BETA))}; }
    __device__ __forceinline__ Row fetch(uint32_t o) const {
        Row w;
        w.zx = at(z, o); w.px = at(p, o);
        if (ST) {
            w.zy = at(z + k.hs, o); w.py = at(p + k.hs, o); w.a = at(k.Dx, o); w.b = at(k.Dy, o); w.a11 = at(ps.i11, o); w.a12 = at(ps.i12, o);
        } else {
            w.zy = w.py = w.a = w.b = w.a12 = 0.0; w.a11 = ps.inv_mu;
        }
        return w;
    }
    static constexpr int kStageStreams = ST ? 2 : 1;
    __device__ __forceinline__ double* stage_out(int sv) const { return sv == 0 ? q : q + k.hs; }
    __device__ __forceinline__ void apply(int64_t row, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool lead, const Uni& u,
                                          const Row& w, double (&v)[1], double (&red)[1]) const {
        apply_staged(row, o, accv, valid, owner, lead, u, w, v, red, nullptr, 0);
    }
    __device__ __forceinline__ void apply_staged(int64_t, uint32_t o, const double (&accv)[1], bool valid, bool owner, bool, const Uni& u,
                                                 const Row& w, double (&v)[1], double (&red)[1], double* slot, int sstride) const {
        const double acc = accv[0];
        const bool st = valid && owner;
        const double pnx = fma(u.beta, w.px, w.zx);                                      // :217 (stored by P3: the same fma)
        double qx, qy = 0.0, pny = 0.0;
        if (!ST) {
            qx = fma(mu, pnx, acc);                                                      // :222
        } else {
            pny = fma(u.beta, w.py, w.zy);
            const double ww = w.a * pnx + w.b * pny;                                     // diagonal block of J p
            qx = fma(mu, pnx, fma(w.a, ww, acc));
            qy = fma(mu, pny, w.b * ww);
        }
        if (st) {
            if (slot) { slot[0] = qx; if (ST) slot[sstride] = qy; }
            else { *reinterpret_cast<double*>(reinterpret_cast<char*>(q) + o) = qx; if (ST) *reinterpret_cast<double*>(reinterpret_cast<char*>(q + k.hs) + o) = qy; }
            red[0] += ST ? fma(pny, qy, pnx * qx) : pnx * qx;                            // p'q   :226
        }
        v[0] = 0.0;                                                                      // (no second product needed: sP is measured directly)
    }
};
struct PrecP3F {  // p = z + beta p (stored) ; x += alpha p ; r -= alpha q ; r'r
    double* x;
    double* r;
    double* p;
    const double* z;
    const double* q;
    const double* scal;
    const int64_t* istat;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        const double alpha = ld_scal(scal + P_ALPHA), beta = ld_scal(scal + P_BETA);
        double2 pp = ld2(p + i);
        const double2 zz = ld2(z + i), qq = ld2(q + i);
        double2 xx = ld2(x + i), rr = ld2(r + i);
        pp = make_double2(fma(beta, pp.x, zz.x), fma(beta, pp.y, zz.y));                 // :217
        xx = make_double2(fma(alpha, pp.x, xx.x), fma(alpha, pp.y, xx.y));               // :232
        rr = make_double2(fma(-alpha, qq.x, rr.x), fma(-alpha, qq.y, rr.y));             // :233
        if (v1) { st2(p + i, pp); st2(x + i, xx); st2(r + i, rr); }
        else if (v0) { p[i] = pp.x; x[i] = xx.x; r[i] = rr.x; }
        double s = 0.0;
        if (v0) s = rr.x * rr.x;
        if (v1) s = fma(rr.y, rr.y, s);
        red[0] += s;
    }
};
struct PrecInit {       // rho_prev = 1 (:203), status
    double* scal;
    int64_t* istat;
    double tol;
    int64_t maxit;
    PcgHostMirror hm;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void run(double*) const {
        scal[P_RHO] = 1.0;
        scal[P_NRES] = INFINITY;
        scal[P_TOL] = tol;
        istat[IP_ITER] = 0;
        istat[IP_MAXIT] = maxit;
        const int64_t st = (maxit > 0) ? PST_RUNNING : PST_DONE;
        istat[IP_STATUS] = st;
        hm.publish(st, 0);
    }
};
struct PrecPost3 {
    double* scal;
    int64_t* istat;
    PcgHostMirror hm;
    __device__ __forceinline__ bool skip() const { return ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ void run(double*) const {
        const double nres = sqrt(ld_scal(scal + P_RR));     // :235
        scal[P_NRES] = nres;
        const int64_t it = ld_stat(istat + IP_ITER) + 1;    // :237
        istat[IP_ITER] = it;
        int64_t st = PST_RUNNING;
        if (!(nres > ld_scal(scal + P_TOL)) || it >= ld_stat(istat + IP_MAXIT)) st = PST_DONE;   // :207
        if (st != PST_RUNNING) istat[IP_STATUS] = st;
        hm.publish(st, it);
    }
};
// the replicated m-vector steps between the passes (one workgroup):
//   mode 1 (after Z): rho = z'r, beta = rho / rho_prev (:212-216), J p = J z + beta J p
//   mode 2 (after Q): alpha = rho / p'q (:227)
//   mode 0 / 3 (after S): c1 = K sP  (3: only while the solve is running)
struct PrecSmall {
    const double* out1;   // [J z (m) ; z'r]
    const double* out2;   // [- (m) ; p'q]
    double* tp;           // J p
    double* sP;
    double* c1;
    const double* K;      // m x m, column-major (symmetric)
    double* scal;
    const int64_t* istat;
    int m;
};
__global__ __launch_bounds__(1024) void prec_small_kernel(PrecSmall s, int mode) {
    if (mode != 0 && ld_stat(s.istat + IP_STATUS) != PST_RUNNING) return;
    __shared__ double sp[kOnepassMaxCols];
    const int m = s.m, tid = threadIdx.x;
    if (mode == 1) {
        const double rho = ld_scal(s.out1 + m), rho_prev = ld_scal(s.scal + P_RHO);
        const double beta = rho / rho_prev;
        for (int j = tid; j < m; j += 1024) s.tp[j] = fma(beta, s.tp[j], ld_scal(s.out1 + j));
        __syncthreads();
        if (tid == 0) { s.scal[P_RHO] = rho; s.scal[P_BETA] = beta; }
        return;
    }
    if (mode == 2) {
        if (tid == 0) s.scal[P_ALPHA] = ld_scal(s.scal + P_RHO) / ld_scal(s.out2 + m);
        return;
    }
    for (int j = tid; j < m; j += 1024) sp[j] = ld_scal(s.sP + j);
    __syncthreads();
    // c1 = K sP: a group of 16 lanes per row (K symmetric: row j = column j, read along the column)
    const int g = tid >> 4, l = tid & 15;
    for (int j = g; j < m; j += 64) {
        double a = 0.0;
        for (int k = l; k < m; k += 16) a = fma(s.K[(size_t)j * m + k], sp[k], a);
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
        if (l == 0) s.c1[j] = a;
    }
}
template <bool ST>
struct PrecS0V {     // producer of sP = Jct'(D0^-1 r)_x
    const double* r;
    PrecScale ps;
    int64_t hs;
    const int64_t* istat;     // nullptr: always (the start of a solve); else only while the solve is running
    __device__ __forceinline__ bool skip() const { return istat != nullptr && ld_stat(istat + IP_STATUS) != PST_RUNNING; }
    __device__ __forceinline__ double2 load(int64_t i, bool v0, bool v1) const {
        const double2 rx = ld2(r + i);
        double2 u;
        if (ST) {
            const double2 ry = ld2(r + hs + i), a11 = ld2(ps.i11 + i), a12 = ld2(ps.i12 + i);
            u = make_double2(fma(a12.x, ry.x, a11.x * rx.x), fma(a12.y, ry.y, a11.y * rx.y));
        } else {
            u = make_double2(ps.inv_mu * rx.x, ps.inv_mu * rx.y);
        }
        return make_double2(v0 ? u.x : 0.0, v1 ? u.y : 0.0);
    }
};

}  // namespace lfpsqp

using namespace lfpsqp;

template <bool ST>
static int pcg_pre_impl(lfpsqp_ctx* ctx, double mu, const lfpsqp_basis* Jop, const lfpsqp_pcg_precond* P, lfpsqp_vec* x, lfpsqp_vec* r, lfpsqp_vec* p,
                        lfpsqp_vec* z, double tol, int64_t maxiter, int* flag, int64_t* iters) {
    const int m = (int)Jop->ncols;
    const int64_t nv = r->n;
    const int64_t N = ST ? Jop->Dx->n : nv, hs = ST ? lfpsqp_half_stride(N) : 0;
    const lfpsqp_mat* A = Jop->Z;
    double* scal = ctx->scal;
    int64_t* istat = ctx->istat;
    const PcgHostMirror hm{ctx->h_istat};
    volatile int64_t* hstat = ctx->h_istat;
    hstat[IP_STATUS] = PST_RUNNING;
    hstat[IP_ITER] = 0;
    for (int k = 0; k < kPRing; ++k) hstat[kPRingOff + k] = PST_RUNNING;
    const size_t mm = (size_t)m * m;
    LF_TRY(ensure_small(ctx, mm + 8));
    LF_TRY(ensure_mvec(ctx, (size_t)6 * m + 64));
    double* dK = ctx->small;
    double* out1 = ctx->d_m;                                   // m + 1
    double* out2 = out1 + round_up(2 * m + 1, 2) + 2;          // m + 1
    double* tp = out2 + round_up(m + 1, 2) + 2;
    double* sP = tp + round_up(m, 2);
    double* c1 = sP + round_up(m, 2);
    LF_HIP(ctx, hipMemcpyAsync(dK, P->K, sizeof(double) * mm, hipMemcpyHostToDevice, ctx->stream));
    LF_HIP(ctx, hipMemsetAsync(tp, 0, sizeof(double) * m, ctx->stream));
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));            // P->K is caller-owned pageable memory
    const PrecScale ps{ST ? P->i11->p : nullptr, ST ? P->i12->p : nullptr, ST ? P->i22->p : nullptr, 1.0 / mu};
    const PStack sk = ST ? PStack{hs, Jop->Dx->p, Jop->Dy->p, Jop->sx->p, Jop->sy->p, nullptr} : PStack{0, nullptr, nullptr, nullptr, nullptr, nullptr};
    const PrecSmall sm{out1, out2, tp, sP, c1, dK, scal, istat, m};
    LF_TRY(lfpsqp_vec_fill(ctx, p, 0.0));                                                        // :204
    hipLaunchKernelGGL((post_kernel<PrecInit>), dim3(1), dim3(1), 0, ctx->stream, scal, PrecInit{scal, istat, tol, maxiter, hm});
    LF_LAUNCH_CHECK(ctx);
    LF_TRY(run_gemv_t(ctx, A, m, N, PrecS0V<ST>{r->p, ps, hs, nullptr}, sP));                    // sP = Jct'(D0^-1 r)_x
    hipLaunchKernelGGL(prec_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, sm, 0);          // c1 = K sP
    LF_LAUNCH_CHECK(ctx);
    int64_t it = 0;
    bool done = maxiter <= 0;
    while (!done && it < maxiter) {
        if (it > 0) {                                          // sP of the updated residual, measured directly (a no-op once the solve has stopped)
            LF_TRY(run_gemv_t(ctx, A, m, N, PrecS0V<ST>{r->p, ps, hs, istat}, sP, 4));
            hipLaunchKernelGGL(prec_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, sm, 3);
            LF_LAUNCH_CHECK(ctx);
        }
        LF_TRY((run_onepass<PrecZE<ST>, 1, 1>(ctx, A, m, m, N, c1, PrecZE<ST>{r->p, z->p, ps, hs, istat}, out1, 5)));
        hipLaunchKernelGGL(prec_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, sm, 1);
        LF_LAUNCH_CHECK(ctx);
        LF_TRY((run_onepass<PrecQE<ST>, 1, 1>(ctx, A, m, m, N, tp, PrecQE<ST>{z->p, p->p, P->q->p, mu, ps, sk, scal, istat}, out2, 5)));
        hipLaunchKernelGGL(prec_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, sm, 2);
        LF_LAUNCH_CHECK(ctx);
        LF_TRY((run_vec<PrecP3F, 1, PrecPost3>(ctx, nv, PrecP3F{x->p, r->p, p->p, z->p, P->q->p, scal, istat}, 0u, scal + P_RR,
                                               PrecPost3{scal, istat, hm}, 6)));
        LF_HIP(ctx, hipEventRecord(ctx->ev_slot[it & 3], ctx->stream));
        if (it >= 2) {
            LF_HIP(ctx, hipEventSynchronize(ctx->ev_slot[(it - 2) & 3]));
            if (hstat[kPRingOff + ((it - 1) % kPRing)] != PST_RUNNING) done = true;
        }
        ++it;
    }
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *iters = hstat[IP_ITER];
    *flag = (*iters == maxiter) ? 1 : 0;                                                         // :240-243
    if (ctx->profiling) prof_collect(ctx);
    return 0;
}

extern "C" int lfpsqp_pcg_pre(lfpsqp_ctx* ctx, double mu, const lfpsqp_basis* Jop, const lfpsqp_pcg_precond* P, lfpsqp_vec* x, lfpsqp_vec* r,
                              lfpsqp_vec* p, lfpsqp_vec* z, double tol, int64_t maxiter, int* flag, int64_t* iters) {
    LF_RANGE("lfpsqp_pcg_pre");
    LF_ARG(ctx, ctx && Jop && P && P->K && P->q && x && r && p && z && flag && iters && mu > 0.0);
    const bool stacked = Jop->Dx != nullptr;
    const int m = (int)Jop->ncols;
    const int64_t nv = r->n;
    LF_ARG(ctx, x->n == nv && p->n == nv && z->n == nv && P->q->n == nv && m >= 1 && Jop->Z && m <= Jop->Z->m);
    int64_t N = nv;
    if (stacked) {
        LF_ARG(ctx, Jop->Dy && Jop->sx && Jop->sy && P->i11 && P->i12 && P->i22);
        N = Jop->Dx->n;
        LF_ARG(ctx, nv == lfpsqp_half_stride(N) + N && Jop->Dy->n == N && Jop->Z->n == N && P->i11->n == N && P->i12->n == N && P->i22->n == N);
    } else {
        LF_ARG(ctx, Jop->Z->n == nv);
    }
    if (Jop->S || !onepass_cw(ctx, m, Jop->Z->ld, N))
        return set_err(ctx, LFPSQP_ERR_UNSUPPORTED, "preconditioned pcg! needs the one-pass kernels over a dense Jct (4 .. 1024 columns); use lfpsqp_pcg");
    return stacked ? pcg_pre_impl<true>(ctx, mu, Jop, P, x, r, p, z, tol, maxiter, flag, iters)
                   : pcg_pre_impl<false>(ctx, mu, Jop, P, x, r, p, z, tol, maxiter, flag, iters);
}

extern "C" int lfpsqp_pcg(lfpsqp_ctx* ctx, double mu, const lfpsqp_basis* Jop, lfpsqp_vec* x, lfpsqp_vec* r, lfpsqp_vec* p, lfpsqp_vec* z,
                          lfpsqp_vec* tmp_w, lfpsqp_vec* tmp_m, double tol, int64_t maxiter, int* flag, int64_t* iters) {
    LF_RANGE("lfpsqp_pcg");
    LF_ARG(ctx, ctx && Jop && x && r && p && z && tmp_m && flag && iters);
    const bool stacked = Jop->Dx != nullptr;
    const int m = (int)Jop->ncols;
    const int64_t nv = r->n;
    LF_ARG(ctx, x->n == nv && p->n == nv && z->n == nv && m >= 0 && (m == 0 || (Jop->Z && m <= Jop->Z->m)) && tmp_m->n >= m);
    int64_t N = nv, hs = 0;
    if (stacked) {
        LF_ARG(ctx, Jop->Dy && Jop->sx && Jop->sy && tmp_w);
        N = Jop->Dx->n;
        hs = lfpsqp_half_stride(N);
        LF_ARG(ctx, nv == hs + N && Jop->Dy->n == N && Jop->sx->n == N && Jop->sy->n == N && tmp_w->n == N && (m == 0 || Jop->Z->n == N));
    } else {
        LF_ARG(ctx, m == 0 || Jop->Z->n == nv);
    }
    double* scal = ctx->scal;
    int64_t* istat = ctx->istat;
    const lfpsqp_mat* Z = m > 0 ? Jop->Z : nullptr;
    const PcgHostMirror hm{ctx->h_istat};
    volatile int64_t* hstat = ctx->h_istat;
    hstat[IP_STATUS] = PST_RUNNING;
    hstat[IP_ITER] = 0;
    for (int k = 0; k < kPRing; ++k) hstat[kPRingOff + k] = PST_RUNNING;
    const PStack sk = stacked ? PStack{hs, Jop->Dx->p, Jop->Dy->p, Jop->sx->p, Jop->sy->p, tmp_w->p} : PStack{0, nullptr, nullptr, nullptr, nullptr, nullptr};

    LF_TRY(lfpsqp_vec_fill(ctx, p, 0.0));                                                        // :204
    LF_TRY((run_vec<RRF, 1, PInit>(ctx, nv, RRF{r->p}, 0u, scal + P_RR, PInit{scal, istat, tol, maxiter, hm})));
    const lfpsqp_spmat* S = Jop->S;
    const bool sparse = S && m > 0 && S->m == m && S->n == N;
    const bool fused = !sparse && m > 0 && onepass_cw(ctx, m, Z->ld, N) != 0;
    double* US = nullptr;                            // [u = J z (m) ; s = J r (m) ; p'z]
    if (fused) {
        LF_TRY(ensure_mvec(ctx, (size_t)2 * m + 16));
        US = ctx->d_m;
    }
    int64_t it = 0;
    bool done = maxiter <= 0;
    while (!done && it < maxiter) {
        if (sparse) {
            const P1V p1{p->p, r->p, scal, istat};
            const P2E p2{p->p, z->p, mu, istat};
            if (stacked) {
                LF_TRY((run_vec<P1SparseF<P1VS>, 0, NoPost>(ctx, N, P1SparseF<P1VS>{P1VS{p1, sk}, z->p}, 0u, nullptr, NoPost(), 4)));
                LF_TRY(spmv_t(ctx, S, z->p, tmp_m->p));
                LF_TRY((run_vec<P2SparseF<P2ES>, 1, PPost2>(ctx, N, P2SparseF<P2ES>{P2ES{p2, sk}, ell_rows(S, tmp_m->p)}, 0u, scal + P_PZ,
                                                            PPost2{scal, istat}, 5)));
            } else {
                LF_TRY((run_vec<P1SparseF<P1V>, 0, NoPost>(ctx, nv, P1SparseF<P1V>{p1, nullptr}, 0u, nullptr, NoPost(), 4)));
                LF_TRY(spmv_t(ctx, S, p->p, tmp_m->p));
                LF_TRY((run_vec<P2SparseF<P2E>, 1, PPost2>(ctx, nv, P2SparseF<P2E>{p2, ell_rows(S, tmp_m->p)}, 0u, scal + P_PZ, PPost2{scal, istat}, 5)));
            }
        } else if (!fused || it == 0) {
            const P1V p1{p->p, r->p, scal, istat};
            if (stacked) LF_TRY(run_gemv_t(ctx, Z, m, N, P1VS{p1, sk}, tmp_m->p, 4));
            else LF_TRY(run_gemv_t(ctx, Z, m, N, p1, tmp_m->p, 4));
        } else {
            hipLaunchKernelGGL(pcg_tmp_kernel, dim3(1), dim3(256), 0, ctx->stream, PcgTmp{US, tmp_m->p, scal, istat, m});
            LF_LAUNCH_CHECK(ctx);
        }
        if (sparse) {
            // (both products done above)
        } else if (fused) {
            if (stacked) {
                const PcgInnerE<true> fe{p->p, r->p, z->p, mu, scal, istat, it == 0 ? 1 : 0, sk};
                LF_TRY((run_onepass<PcgInnerE<true>, 2, 1>(ctx, Z, m, m, N, tmp_m->p, fe, US, 5)));
            } else {
                const PcgInnerE<false> fe{p->p, r->p, z->p, mu, scal, istat, it == 0 ? 1 : 0, sk};
                LF_TRY((run_onepass<PcgInnerE<false>, 2, 1>(ctx, Z, m, m, N, tmp_m->p, fe, US, 5)));
            }
            hipLaunchKernelGGL((post_kernel<PPost2F>), dim3(1), dim3(1), 0, ctx->stream, scal, PPost2F{scal, istat, US + 2 * m});
            LF_LAUNCH_CHECK(ctx);
        } else {
            const P2E p2{p->p, z->p, mu, istat};
            if (stacked) LF_TRY((run_gemv_n<P2ES, 1, PPost2>(ctx, Z, m, N, tmp_m->p, P2ES{p2, sk}, scal + P_PZ, PPost2{scal, istat}, 5)));
            else LF_TRY((run_gemv_n<P2E, 1, PPost2>(ctx, Z, m, N, tmp_m->p, p2, scal + P_PZ, PPost2{scal, istat}, 5)));
        }
        LF_TRY((run_vec<P3F, 1, PPost3>(ctx, nv, P3F{x->p, r->p, p->p, z->p, scal, istat, (fused && it > 0) ? 1 : 0}, 0u, scal + P_RR,
                                         PPost3{scal, istat, hm}, 6)));
        // rank-deterministic stop: the status of iteration it-2 (device iteration number it-1), after its event
        LF_HIP(ctx, hipEventRecord(ctx->ev_slot[it & 3], ctx->stream));
        if (it >= 2) {
            LF_HIP(ctx, hipEventSynchronize(ctx->ev_slot[(it - 2) & 3]));
            if (hstat[kPRingOff + ((it - 1) % kPRing)] != PST_RUNNING) done = true;
        }
        ++it;
    }
    LF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *iters = hstat[IP_ITER];
    *flag = (*iters == maxiter) ? 1 : 0;                                                         // :240-243
    if (ctx->profiling) prof_collect(ctx);
    return 0;
}
