// Bound-manifold operators on the device: the elementwise pieces of
// src/inequality_helper.jl (InequalityData, generate_initial_y!, calculate_h!,
// inequality_gradient!) and src/retractions.jl:451-500 (y_retract!), plus the two mul!
// methods of InequalityDecompProject (Q'v and Q[w;t]) on the stacked layout.
#include <math.h>

#include "internal.h"
#include "sparse.h"

namespace lfpsqp {

struct IneqD {  // device view of lfpsqp_ineq_data
    const double *q, *r, *s, *t;
    int64_t n;
};
static inline IneqD view(const lfpsqp_ineq_data* id) { return IneqD{id->q->p, id->r->p, id->s->p, id->t->p, id->n}; }

// All functors below work on row pairs (i, i+1) of the N-long halves; y-half = base + hs.

struct BuildIneqF {  // src/inequality_helper.jl:54-82
    const double *xl, *xu;
    double *q, *r, *s, *t;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void one(int64_t i) const {
        const double l = xl[i], u = xu[i];
        const bool linf = isinf(l), uinf = isinf(u);
        double qq = 0.0, rr = 0.0, ss = 0.0, tt = 0.0;
        if (linf && uinf) {
        } else if (!linf && uinf) { rr = l; ss = -1.0; tt = l; }
        else if (linf && !uinf) { rr = u; ss = 1.0; tt = u; }
        else { qq = 1.0; rr = (u + l) / 2; ss = 1.0; tt = (u - l) * (u - l) / 4; }
        q[i] = qq; r[i] = rr; s[i] = ss; t[i] = tt;
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0) one(i);
        if (v1) one(i + 1);
    }
};

struct InitialYF {  // :92-109
    double* x;  // stacked: y at x + hs
    int64_t hs;
    IneqD id;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void one(int64_t i) const {
        const double xi = x[i], qq = id.q[i], rr = id.r[i], ss = id.s[i], tt = id.t[i];
        double y;
        if (ss == 0.0) y = xi;                                              // line
        else if (qq == 0.0) y = sqrt(fmax(-(xi - tt) / ss, 0.0)) + rr;      // parabola
        else y = sqrt(fmax(tt - (xi - rr) * (xi - rr), 0.0)) + rr;          // circle
        x[hs + i] = y;
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0) one(i);
        if (v1) one(i + 1);
    }
};

__device__ __forceinline__ double h_of(double x, double y, double q, double r, double s, double t) {   // :118-119
    return q * ((x - r) * (x - r)) + (1.0 - q * q) * x + s * ((y - r) * (y - r)) - (1.0 - s * s) * y - t;
}

struct CalcHF {
    const double* x;
    int64_t hs;
    IneqD id;
    double* h;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double* red) const {
        if (v0) {
            const double v = h_of(x[i], x[hs + i], id.q[i], id.r[i], id.s[i], id.t[i]);
            h[i] = v;
            red[0] = nanmax(red[0], fabs(v));
        }
        if (v1) {
            const double v = h_of(x[i + 1], x[hs + i + 1], id.q[i + 1], id.r[i + 1], id.s[i + 1], id.t[i + 1]);
            h[i + 1] = v;
            red[0] = nanmax(red[0], fabs(v));
        }
    }
};

struct IneqGradF {  // :125-141
    const double* x;
    int64_t hs;
    IneqD id;
    double *Dx, *Dy, *S, *sx, *sy;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void one(int64_t i) const {
        const double qq = id.q[i], rr = id.r[i], ss = id.s[i];
        double dx = 2.0 * qq * (x[i] - rr) + (qq == 0.0 ? 1.0 : 0.0);
        double dy = 2.0 * ss * (x[hs + i] - rr) - (ss == 0.0 ? 1.0 : 0.0);
        const double nrm = sqrt(dx * dx + dy * dy);
        dx /= nrm;
        dy /= nrm;
        Dx[i] = dx; Dy[i] = dy; S[i] = nrm;
        if (sx) { sx[i] = dy * dy; sy[i] = -dx * dy; }
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0) one(i);
        if (v1) one(i + 1);
    }
};

// y_retract! for one variable (src/retractions.jl:459-497)
__device__ __forceinline__ void y_retract_one(double& xn, double& yn, double xo, double yo, double qq, double rr, double ss, double tt) {
    if (ss == 0.0) {            // line :463
        xn = yn;
    } else if (qq == 0.0) {     // parabola :464-486
        const double g1 = -ss, g2 = -2.0 * (yo - rr);
        const double ng = sqrt(g1 * g1 + g2 * g2);
        const double ux = xo - xn + g1 / ng;
        const double uy = yo - yn + g2 / ng;
        const double a = ss * (uy * uy);
        const double b = ux + 2.0 * ss * (yn - rr) * uy;
        const double c = xn + ss * ((yn - rr) * (yn - rr)) - rr;
        const double a1 = -b / (2.0 * a);
        const double a2 = sqrt(b * b - 4.0 * a * c) / (2.0 * a);
        const double gam = fmin(a1 + a2, a1 - a2);
        xn += gam * ux;
        yn += gam * uy;
    } else {                    // circle :487-496
        const double c = rr, rho = sqrt(tt);
        const double dist = sqrt((xn - c) * (xn - c) + (yn - c) * (yn - c));
        const double y2 = c + rho * (yn - c) / dist;
        const double x2 = c + rho * (xn - c) / dist;
        yn = y2;
        xn = x2;
    }
}

struct YRetractF {
    double* xn;       // stacked new point (overwritten)
    const double* xo; // stacked base point
    int64_t hs;
    IneqD id;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void one(int64_t i) const {
        double a = xn[i], b = xn[hs + i];
        y_retract_one(a, b, xo[i], xo[hs + i], id.q[i], id.r[i], id.s[i], id.t[i]);
        xn[i] = a;
        xn[hs + i] = b;
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0) one(i);
        if (v1) one(i + 1);
    }
};

// ---- Q'v and Q[w;t] ------------------------------------------------------------
struct QtV {  // producer: v_eff = sx.*vx + sy.*vy ; stores w = Dx.*vx + Dy.*vy
    const double* v;
    int64_t hs;
    const double *Dx, *Dy, *sx, *sy;
    double* w;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 vx = ld2(v + r), vy = ld2(v + hs + r);
        const double2 dx = ld2(Dx + r), dy = ld2(Dy + r), ax = ld2(sx + r), ay = ld2(sy + r);
        const double2 ww = make_double2(dx.x * vx.x + dy.x * vy.x, dx.y * vx.y + dy.y * vy.y);
        if (v1) st2(w + r, ww);
        else if (v0) w[r] = ww.x;
        const double2 e = make_double2(ax.x * vx.x + ay.x * vy.x, ax.y * vx.y + ay.y * vy.y);
        return make_double2(v0 ? e.x : 0.0, v1 ? e.y : 0.0);
    }
};

struct QApplyE {  // y = alpha*[Dx.*w + sx.*acc ; Dy.*w + sy.*acc] + beta*y
    double* y;
    int64_t hs;
    const double *Dx, *Dy, *sx, *sy;
    const double* w;  // may be null
    double alpha, beta;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double*) const {
        const double2 ax = ld2(sx + r), ay = ld2(sy + r);
        double2 ox = make_double2(ax.x * acc.x, ax.y * acc.y);
        double2 oy = make_double2(ay.x * acc.x, ay.y * acc.y);
        if (w) {
            const double2 ww = ld2(w + r), dx = ld2(Dx + r), dy = ld2(Dy + r);
            ox.x += dx.x * ww.x; ox.y += dx.y * ww.y;
            oy.x += dy.x * ww.x; oy.y += dy.y * ww.y;
        }
        ox.x *= alpha; ox.y *= alpha; oy.x *= alpha; oy.y *= alpha;
        if (beta != 0.0) {
            const double2 yx = ld2(y + r), yy = ld2(y + hs + r);
            ox.x += beta * yx.x; ox.y += beta * yx.y;
            oy.x += beta * yy.x; oy.y += beta * yy.y;
        }
        if (v1) { st2(y + r, ox); st2(y + hs + r, oy); }
        else if (v0) { y[r] = ox.x; y[hs + r] = oy.x; }
    }
};

struct LambdaYE {  // lamy = (w - Dx.*acc) ./ S
    const double *Dx, *S, *w;
    double* out;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t r, double2 acc, bool v0, bool v1, double*) const {
        const double2 dx = ld2(Dx + r), ss = ld2(S + r), ww = ld2(w + r);
        // reference order: lamy = Jct*lam ; lamy *= -Dx/S ; lamy += w/S
        const double2 o = make_double2(acc.x * (-1.0 * dx.x / ss.x) + ww.x / ss.x, acc.y * (-1.0 * dx.y / ss.y) + ww.y / ss.y);
        if (v1) st2(out + r, o);
        else if (v0) out[r] = o.x;
    }
};

// e = |Dy| .* dx - Dx .* sgn(Dy) .* dy: the stacked vector d = [dx; dy] as ONE right-hand column of the weighted Gram pass (weights Dy.^2):
// Jct'(sqrt(w2) .* e) = Jct'(Dy.^2 .* dx - Dx.*Dy .* dy) = Jct'(sx .* dx + sy .* dy), the m-part of Q'd (src/inequality_helper.jl:197-212)
struct IneqRhsF {
    const double* d;
    int64_t hs;
    const double *Dx, *Dy;
    double* e;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        const double2 dx = ld2(d + i), dy = ld2(d + hs + i), gx = ld2(Dx + i), gy = ld2(Dy + i);
        const double2 o = make_double2(fabs(gy.x) * dx.x - gx.x * copysign(gy.x != 0.0 ? 1.0 : 0.0, gy.x) * dy.x,
                                       fabs(gy.y) * dx.y - gx.y * copysign(gy.y != 0.0 ? 1.0 : 0.0, gy.y) * dy.y);
        if (v1) st2(e + i, o);
        else if (v0) e[i] = o.x;
    }
};

struct AugDiagF {
    const double *hx, *lamy;
    IneqD id;
    double* a;
    int64_t hs;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void one(int64_t i) const {
        const double ly = lamy[i];
        a[i] = hx[i] + 2.0 * ly * id.q[i];
        a[hs + i] = 2.0 * ly * id.s[i];
    }
    __device__ __forceinline__ void apply(int64_t i, bool v0, bool v1, double*) const {
        if (v0) one(i);
        if (v1) one(i + 1);
    }
};

}  // namespace lfpsqp

using namespace lfpsqp;

static bool ineq_ok(const lfpsqp_ineq_data* id) {
    return id && id->q && id->r && id->s && id->t && id->q->n == id->n && id->r->n == id->n && id->s->n == id->n && id->t->n == id->n;
}

extern "C" {

int64_t lfpsqp_half_stride(int64_t N) { return round_up(N > 0 ? N : 1, kPadRows); }

int lfpsqp_ineq_data_build(lfpsqp_ctx* ctx, const lfpsqp_vec* xl, const lfpsqp_vec* xu, lfpsqp_vec* q, lfpsqp_vec* r, lfpsqp_vec* s,
                           lfpsqp_vec* t) {
    LF_ARG(ctx, ctx && xl && xu && q && r && s && t);
    const int64_t N = xl->n;
    if (xu->n != N) return set_err(ctx, LFPSQP_ERR_ARG, "xl and xu are of different lengths");   // src/inequality_helper.jl:42-44
    LF_ARG(ctx, q->n == N && r->n == N && s->n == N && t->n == N);
    return run_vec<BuildIneqF, 0, NoPost>(ctx, N, BuildIneqF{xl->p, xu->p, q->p, r->p, s->p, t->p}, 0u, nullptr, NoPost());
}

int lfpsqp_generate_initial_y(lfpsqp_ctx* ctx, lfpsqp_vec* xaug, const lfpsqp_ineq_data* id) {
    LF_ARG(ctx, ctx && xaug && ineq_ok(id));
    const int64_t hs = lfpsqp_half_stride(id->n);
    LF_ARG(ctx, xaug->n == hs + id->n);
    return run_vec<InitialYF, 0, NoPost>(ctx, id->n, InitialYF{xaug->p, hs, view(id)}, 0u, nullptr, NoPost());
}

int lfpsqp_calculate_h(lfpsqp_ctx* ctx, lfpsqp_vec* h, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id, double* hmax) {
    LF_ARG(ctx, ctx && h && xaug && ineq_ok(id) && h->n >= id->n);
    const int64_t hs = lfpsqp_half_stride(id->n);
    LF_ARG(ctx, xaug->n == hs + id->n);
    LF_TRY((run_vec<CalcHF, 1, NoPost>(ctx, id->n, CalcHF{xaug->p, hs, view(id), h->p}, 1u, ctx->scal + 32, NoPost())));
    if (hmax) return read_back(ctx, ctx->scal + 32, hmax, 1);
    return 0;
}

int lfpsqp_inequality_gradient(lfpsqp_ctx* ctx, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id, lfpsqp_vec* Dx, lfpsqp_vec* Dy,
                               lfpsqp_vec* S, lfpsqp_vec* sx, lfpsqp_vec* sy) {
    LF_ARG(ctx, ctx && xaug && ineq_ok(id) && Dx && Dy && S && Dx->n == id->n && Dy->n == id->n && S->n == id->n);
    LF_ARG(ctx, (!sx && !sy) || (sx && sy && sx->n == id->n && sy->n == id->n));
    const int64_t hs = lfpsqp_half_stride(id->n);
    LF_ARG(ctx, xaug->n == hs + id->n);
    return run_vec<IneqGradF, 0, NoPost>(ctx, id->n, IneqGradF{xaug->p, hs, view(id), Dx->p, Dy->p, S->p, sx ? sx->p : nullptr, sy ? sy->p : nullptr},
                                         0u, nullptr, NoPost());
}

int lfpsqp_calculate_lambda_y(lfpsqp_ctx* ctx, const lfpsqp_mat* Jct, int64_t ncols, const lfpsqp_vec* lam, const lfpsqp_vec* Dx,
                              const lfpsqp_vec* S, const lfpsqp_vec* w, lfpsqp_vec* lamy) {
    LF_ARG(ctx, ctx && Jct && lam && Dx && S && w && lamy && ncols >= 0 && ncols <= Jct->m && lam->n >= ncols);
    const int64_t N = Jct->n;
    LF_ARG(ctx, Dx->n == N && S->n == N && w->n == N && lamy->n == N);
    return run_gemv_n<LambdaYE, 0, NoPost>(ctx, Jct, (int)ncols, N, lam->p, LambdaYE{Dx->p, S->p, w->p, lamy->p}, nullptr, NoPost());
}

int lfpsqp_augmented_diag(lfpsqp_ctx* ctx, const lfpsqp_vec* hx, const lfpsqp_vec* lamy, const lfpsqp_ineq_data* id, lfpsqp_vec* a) {
    LF_ARG(ctx, ctx && hx && lamy && a && ineq_ok(id) && hx->n == id->n && lamy->n == id->n);
    const int64_t hs = lfpsqp_half_stride(id->n);
    LF_ARG(ctx, a->n == hs + id->n);
    return run_vec<AugDiagF, 0, NoPost>(ctx, id->n, AugDiagF{hx->p, lamy->p, view(id), a->p, hs}, 0u, nullptr, NoPost());
}

int lfpsqp_ineq_rhs(lfpsqp_ctx* ctx, const lfpsqp_vec* daug, const lfpsqp_vec* Dx, const lfpsqp_vec* Dy, lfpsqp_vec* e) {
    LF_ARG(ctx, ctx && daug && Dx && Dy && e);
    const int64_t N = Dx->n, hs = lfpsqp_half_stride(N);
    LF_ARG(ctx, Dy->n == N && daug->n == hs + N && e->n >= N && e->p != daug->p);
    return run_vec<IneqRhsF, 0, NoPost>(ctx, N, IneqRhsF{daug->p, hs, Dx->p, Dy->p, e->p}, 0u, nullptr, NoPost());
}

int lfpsqp_y_retract(lfpsqp_ctx* ctx, lfpsqp_vec* xnewaug, const lfpsqp_vec* xaug, const lfpsqp_ineq_data* id) {
    LF_ARG(ctx, ctx && xnewaug && xaug && ineq_ok(id));
    const int64_t hs = lfpsqp_half_stride(id->n);
    LF_ARG(ctx, xaug->n == hs + id->n && xnewaug->n == hs + id->n && xnewaug->p != xaug->p);
    return run_vec<YRetractF, 0, NoPost>(ctx, id->n, YRetractF{xnewaug->p, xaug->p, hs, view(id)}, 0u, nullptr, NoPost());
}

// a plain basis with its generator and the generator's sparse twin: U = A W applied on the nonzeros (sparse.hip)
static bool sp_factored_ok(const lfpsqp_basis* Q, int64_t n) {
    return Q->ncols > 0 && Q->SA && Q->A && plain_mat(Q->A) && Q->W && Q->SA->n == n && Q->A->n == n && Q->SA->m >= 1 && Q->A->m >= Q->SA->m && Q->A->m - Q->SA->m <= 4 &&
           Q->ncols <= 1024;
}

// the basis in factored form with a DENSE generator (lfpsqp_basis.Z == NULL, A and W given): U = [sx; sy] .* (A W) is never materialised,
// U't = W'(A'v) and U t = A (W t) stream A, the m x m factor is applied by a one-workgroup kernel (FINDINGS.md 5.3)
static bool sp_factored_ok(const lfpsqp_basis* Q, int64_t n);
// (a PLAIN basis with a sparse twin of its generator goes through the nonzeros instead, sp_factored_*; a bound-stacked one streams the dense twin)
static bool dense_factored(const lfpsqp_basis* Q) {
    return !Q->Z && Q->A && Q->W && Q->ncols > 0 && Q->ncols <= Q->A->m && Q->A->m <= kOnepassMaxCols && !(!Q->Dx && sp_factored_ok(Q, Q->A->n));
}
struct QPlainV {
    const double* v;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ double2 load(int64_t r, bool v0, bool v1) const {
        const double2 a = ld2(v + r);
        return make_double2(v0 ? a.x : 0.0, v1 ? a.y : 0.0);
    }
};
struct QAxpbyE {       // y = alpha * acc + beta * y
    double* y;
    double alpha, beta;
    __device__ __forceinline__ bool skip() const { return false; }
    __device__ __forceinline__ void apply(int64_t i, double2 acc, bool v0, bool v1, double*) const {
        double2 o;
        if (beta == 0.0) o = make_double2(alpha * acc.x, alpha * acc.y);
        else {
            const double2 yy = ld2(y + i);
            o = make_double2(fma(alpha, acc.x, beta * yy.x), fma(alpha, acc.y, beta * yy.y));
        }
        if (v1) st2(y + i, o);
        else if (v0) y[i] = o.x;
    }
};

int lfpsqp_q_gemv_t(lfpsqp_ctx* ctx, const lfpsqp_basis* Q, const lfpsqp_vec* v, lfpsqp_vec* w, lfpsqp_vec* t) {
    LF_RANGE("lfpsqp_q_gemv_t");
    LF_ARG(ctx, ctx && Q && v && t && Q->ncols >= 0 && (Q->ncols == 0 || dense_factored(Q) || (!Q->Dx && sp_factored_ok(Q, v->n)) || (Q->Z && Q->ncols <= Q->Z->m)) && t->n >= Q->ncols);
    if (dense_factored(Q)) {
        const int m = (int)Q->ncols, wm = (int)Q->A->m;
        const int64_t N = Q->A->n;
        double *dW, *tA, *uA;
        LF_TRY(factored_setup(ctx, Q->A, Q->W, m, &dW, &tA, &uA));
        if (!Q->Dx) {
            LF_ARG(ctx, v->n >= N);
            LF_TRY(run_gemv_t(ctx, Q->A, wm, N, QPlainV{v->p}, tA));
        } else {
            LF_ARG(ctx, Q->Dy && Q->sx && Q->sy && w);
            const int64_t hs = lfpsqp_half_stride(N);
            LF_ARG(ctx, Q->Dx->n == N && v->n == hs + N && w->n == N && Q->Dy->n == N && Q->sx->n == N && Q->sy->n == N);
            LF_TRY(run_gemv_t(ctx, Q->A, wm, N, QtV{v->p, hs, Q->Dx->p, Q->Dy->p, Q->sx->p, Q->sy->p, w->p}, tA));
        }
        return sp_basis_small(ctx, dW, wm, m, tA, t->p, nullptr);
    }
    if (!Q->Dx) {
        if (sp_factored_ok(Q, v->n)) return sp_factored_gemv_t(ctx, Q->SA, Q->A, Q->W, (int)Q->ncols, v->p, t->p);     // on the nonzeros
        LF_ARG(ctx, Q->Z);
        return lfpsqp_gemv_t(ctx, Q->Z, Q->ncols, v, t);
    }
    LF_ARG(ctx, Q->Dy && Q->sx && Q->sy && w);
    const int64_t N = Q->Dx->n, hs = lfpsqp_half_stride(N);
    LF_ARG(ctx, v->n == hs + N && w->n == N && Q->Dy->n == N && Q->sx->n == N && Q->sy->n == N && (!Q->Z || Q->Z->n == N));
    return run_gemv_t(ctx, Q->ncols > 0 ? Q->Z : nullptr, (int)Q->ncols, N, QtV{v->p, hs, Q->Dx->p, Q->Dy->p, Q->sx->p, Q->sy->p, w->p}, t->p);
}

int lfpsqp_q_gemv_n(lfpsqp_ctx* ctx, const lfpsqp_basis* Q, double alpha, const lfpsqp_vec* w, const lfpsqp_vec* t, double beta,
                    lfpsqp_vec* y) {
    LF_RANGE("lfpsqp_q_gemv_n");
    LF_ARG(ctx, ctx && Q && y && Q->ncols >= 0 && (Q->ncols == 0 || ((dense_factored(Q) || (!Q->Dx && sp_factored_ok(Q, y->n)) || (Q->Z && Q->ncols <= Q->Z->m)) && t && t->n >= Q->ncols)));
    if (dense_factored(Q)) {
        const int m = (int)Q->ncols, wm = (int)Q->A->m;
        const int64_t N = Q->A->n;
        double *dW, *tA, *uA;
        LF_TRY(factored_setup(ctx, Q->A, Q->W, m, &dW, &tA, &uA));
        LF_TRY(factored_w_times_t(ctx, dW, wm, m, t->p, uA));
        if (!Q->Dx) {
            LF_ARG(ctx, y->n >= N);
            return run_gemv_n<QAxpbyE, 0, NoPost>(ctx, Q->A, wm, N, uA, QAxpbyE{y->p, alpha, beta}, nullptr, NoPost());
        }
        LF_ARG(ctx, Q->Dy && Q->sx && Q->sy);
        const int64_t hs = lfpsqp_half_stride(N);
        LF_ARG(ctx, Q->Dx->n == N && y->n == hs + N && (!w || w->n == N));
        return run_gemv_n<QApplyE, 0, NoPost>(ctx, Q->A, wm, N, uA, QApplyE{y->p, hs, Q->Dx->p, Q->Dy->p, Q->sx->p, Q->sy->p, w ? w->p : nullptr, alpha, beta},
                                              nullptr, NoPost());
    }
    if (!Q->Dx) {
        if (sp_factored_ok(Q, y->n)) return sp_factored_gemv_n(ctx, Q->SA, Q->A, Q->W, (int)Q->ncols, alpha, t->p, beta, y->p);
        LF_ARG(ctx, Q->Z);
        return lfpsqp_gemv_n(ctx, Q->Z, Q->ncols, alpha, t, beta, y);
    }
    LF_ARG(ctx, Q->Dy && Q->sx && Q->sy);
    const int64_t N = Q->Dx->n, hs = lfpsqp_half_stride(N);
    LF_ARG(ctx, y->n == hs + N && (!w || w->n == N));
    return run_gemv_n<QApplyE, 0, NoPost>(ctx, Q->ncols > 0 ? Q->Z : nullptr, (int)Q->ncols, N, t ? t->p : nullptr,
                                          QApplyE{y->p, hs, Q->Dx->p, Q->Dy->p, Q->sx->p, Q->sy->p, w ? w->p : nullptr, alpha, beta}, nullptr,
                                          NoPost());
}

}  // extern "C"
