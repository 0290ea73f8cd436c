"""Problem front-ends of ``optimize``.

* :class:`QuadLinearBallBox` -- the device-resident problem class of BASELINE configs 2-5:
  f = ||x - xc||^2, dense linear equalities J x = b, optional ball x'x <= R2 (turned into an
  equality with a slack variable exactly as src/optimize.jl:23-51 does) and optional box bounds.
  f, grad!, c!, jac! and the (diagonal) Lagrangian Hessian all run on the device.
* :func:`optimize` -- the reference's method table (src/optimize.jl:13,83,88,107,112,119) for
  arbitrary HOST callables: the "host-callback fallback".  Iterates are downloaded for every user
  call, so it is for plumbing / small problems (config 1), not for the 1e7-variable configs.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from .device import Context, DeviceMatrix, DeviceVector
from .optimize import optimize_core
from .params import LFPSQPParams
from .retractions import DeviceConstraints


class QuadLinearBallBox:
    def __init__(self, ctx: Context, n: int, m: int, Jct: DeviceMatrix, b, R2: Optional[float] = None, xl=None, xu=None,
                 xc: float = 0.0, n_global: Optional[int] = None, owns_slack: bool = True, Jsp=None):
        """n = rows of this rank's shard.  Jct: device matrix with n + ploc rows and m + p columns
        (p = 1 iff R2 is given; ploc = 1 on the rank that owns the slack variable -- the last one --
        else 0) whose leading n x m block holds the constraint gradients; the slack row and the ball
        column are managed here."""
        self.ctx, self.n, self.m, self.xc = ctx, n, m, float(xc)
        self.p = 0 if R2 is None else 1
        self.ploc = self.p if owns_slack else 0
        self.N, self.M = n + self.ploc, m + self.p
        assert Jct.n == self.N and Jct.m == self.M
        self.Jct = Jct
        self.R2 = 0.0 if R2 is None else float(R2)
        # Jsp: optional SparseMatrix (N x m) with the entries of the linear block Jct[:, :m]: c! and ProjPenalty's inner solves
        # then stream its nonzeros (the tangent setup and the Newton retraction keep the dense block)
        self.cons = DeviceConstraints(Jct, m, b, has_ball=self.p == 1, R2=self.R2, n_x=n, slack_row=n if self.ploc else -1, Jsp=Jsp)
        self.xl = None if xl is None else np.asarray(xl, dtype=np.float64)
        self.xu = None if xu is None else np.asarray(xu, dtype=np.float64)
        self.n_global = n if n_global is None else n_global
        self.is_diagonal = True
        # grad^2 of the Lagrangian is 2 I when there is no ball: the truncated-Newton solve converges in ONE projected-CG iteration (config 3).
        # optimize then takes its first allocations instead of placing the work vectors by trial (tens of ms that a handful of iterations
        # cannot repay)
        self.scalar_hessian = self.p == 0 and type(self) is QuadLinearBallBox

    # -- callbacks in the contract of optimize_core -------------------------------------------
    def f(self, x: DeviceVector) -> float:
        out = C.c_double()
        self.ctx.check(self.ctx.L.lfpsqp_sumsq_shift(self.ctx.h, x.h, self.n, self.xc, C.byref(out)))
        return out.value

    def grad_(self, g: DeviceVector, x: DeviceVector):
        self.ctx.check(self.ctx.L.lfpsqp_affine_head(self.ctx.h, 2.0, x.h, -2.0 * self.xc, self.n, g.h))

    def jac_(self, Jct, cval, x):
        return self.cons.jac_(Jct, cval, x)

    def diag_objective_(self, hx: DeviceVector, x: DeviceVector):
        """The objective's part of the diagonal Lagrangian Hessian; the constraints' part is ``self.cons.hess_diag_`` -- the split
        ``optimize`` uses to fold the latter into its tangent-step pass (lfpsqp_tangent_step): diag_ == diag_objective_ + cons.hess_diag_."""
        L = self.ctx.L
        self.ctx.check(L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, 0, self.n, 2.0))
        if self.ploc:
            self.ctx.check(L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, self.n, 1, 0.0))

    def diag_(self, hx: DeviceVector, x: DeviceVector, lam: np.ndarray):
        """diag of grad^2 f + sum lam_i grad^2 c_i: 2 (+ 2 lam_ball) on the user's variables, 0 on the slack."""
        h = 2.0 + (2.0 * float(lam[self.m]) if self.p else 0.0)
        L = self.ctx.L
        self.ctx.check(L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, 0, self.n, h))
        if self.ploc:
            self.ctx.check(L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, self.n, 1, 0.0))

    # -- the slack transformation of src/optimize.jl:23-36 --------------------------------------
    def aux_start(self, x0):
        x0 = np.asarray(x0, dtype=np.float64)
        if not self.p:
            return x0, self.xl, self.xu
        tmp = self.ctx.vector(self.n, x0)                       # global x0'x0 (all-reduced over the shards)
        saved, self.xc = self.xc, 0.0
        xx = self.f(tmp) if self.n else self.f(self.ctx.vector(0))
        self.xc = saved
        tmp.free()
        xl = -np.inf * np.ones(self.n) if self.xl is None else self.xl
        xu = np.inf * np.ones(self.n) if self.xu is None else self.xu
        if not self.ploc:                                       # another rank owns the slack variable
            return x0, xl, xu
        return np.concatenate([x0, [xx - self.R2]]), np.concatenate([xl, [-np.inf]]), np.concatenate([xu, [0.0]])

    def optimize(self, x0, param: LFPSQPParams | None = None, trace=None):
        x0a, xl, xu = self.aux_start(x0)
        x, obj, lam, ti = optimize_core(self.f, self.grad_, self.cons, self.jac_, self, x0a, xl, xu, self.M, param, ctx=self.ctx,
                                        n_global=self.n_global + self.p, trace=trace)
        return x[:self.n], obj, lam, ti


class SeparableLinearBallBox(QuadLinearBallBox):
    """A second device-resident problem class: a SEPARABLE objective f(x) = sum_i phi(x_i - c_i; a_i) with per-variable
    parameters -- ``kind`` 0: a t^2, 1: a t^4 + t^2, 2: a (sqrt(1 + t^2) - 1) (pseudo-Huber) -- under the same constraint set as
    :class:`QuadLinearBallBox` (dense or sparse linear equalities, optional ball with slack, optional box).  f, grad! and the
    diagonal of the Lagrangian Hessian phi''(x_i) + 2 lam_ball are elementwise kernels (lfpsqp_separable), so the whole
    `optimize` run stays on the device and uses the fused projected-CG path; unlike the quadratic class the Hessian changes with x
    and the truncated-Newton solves take several iterations."""

    def __init__(self, ctx: Context, n: int, m: int, Jct: DeviceMatrix, b, kind: int, a, c=0.0, **kw):
        super().__init__(ctx, n, m, Jct, b, **kw)
        self.kind = int(kind)
        self.a_dev = None if np.isscalar(a) else ctx.vector(n, np.asarray(a, dtype=np.float64))
        self.c_dev = None if np.isscalar(c) else ctx.vector(n, np.asarray(c, dtype=np.float64))
        self.a0 = float(a) if np.isscalar(a) else 0.0
        self.c0 = float(c) if np.isscalar(c) else 0.0

    def _sep(self, mode, x, out_vec=None):
        out = C.c_double()
        L = self.ctx.L
        self.ctx.check(L.lfpsqp_separable(self.ctx.h, self.kind, mode, self.a_dev.h if self.a_dev is not None else None, self.a0,
                                          self.c_dev.h if self.c_dev is not None else None, self.c0, x.h, self.n,
                                          out_vec.h if out_vec is not None else None, C.byref(out)))
        return out.value

    def f(self, x: DeviceVector) -> float:
        return self._sep(0, x)

    def grad_(self, g: DeviceVector, x: DeviceVector):
        self._sep(1, x, g)
        if self.ploc:
            self.ctx.check(self.ctx.L.lfpsqp_vec_fill_range(self.ctx.h, g.h, self.n, 1, 0.0))      # the slack variable is not in f

    def diag_objective_(self, hx: DeviceVector, x: DeviceVector):
        self._sep(2, x, hx)
        if self.ploc:
            self.ctx.check(self.ctx.L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, self.n, 1, 0.0))

    def diag_(self, hx: DeviceVector, x: DeviceVector, lam: np.ndarray):
        self._sep(2, x, hx)
        if self.p:                                                                                  # + 2 lam_ball on the user's variables
            ones = getattr(self, "_lamvec", None)
            if ones is None:
                ones = self._lamvec = self.ctx.vector(hx.n)
            L = self.ctx.L
            self.ctx.check(L.lfpsqp_vec_fill_range(self.ctx.h, ones.h, 0, self.n, 2.0 * float(lam[self.m])))
            from .device import axpby
            axpby(1.0, ones, 1.0, hx)
        if self.ploc:
            self.ctx.check(self.ctx.L.lfpsqp_vec_fill_range(self.ctx.h, hx.h, self.n, 1, 0.0))

    def aux_start(self, x0):
        x0 = np.asarray(x0, dtype=np.float64)
        if not self.p:
            return x0, self.xl, self.xu
        tmp = self.ctx.vector(self.n, x0)                       # global x0'x0 through the quadratic kernel
        out = C.c_double()
        self.ctx.check(self.ctx.L.lfpsqp_sumsq_shift(self.ctx.h, tmp.h, self.n, 0.0, C.byref(out)))
        tmp.free()
        xl = -np.inf * np.ones(self.n) if self.xl is None else self.xl
        xu = np.inf * np.ones(self.n) if self.xu is None else self.xu
        if not self.ploc:
            return x0, xl, xu
        return np.concatenate([x0, [out.value - self.R2]]), np.concatenate([xl, [-np.inf]]), np.concatenate([xu, [0.0]])


class ChainSeparableLinear(SeparableLinearBallBox):
    """A separable objective plus a CHAIN term: f(x) = sum_i phi(x_i - c_i; a_i) + kappa/2 sum_{i<n-1} (x_{i+1} - x_i)^2 (smoothing / first
    differences -- the kind of objective whose Hessian the reference reaches only through hess_lag_vec!, src/autodiff_generators.jl:72-107)
    under dense linear equalities.  The Lagrangian Hessian is TRIDIAGONAL: diagonal phi''(x_i) + kappa deg_i (deg = 1 at the two ends, 2
    inside), couplings -kappa.  ``optimize`` hands it to projcg_ as a :class:`TridiagonalOperator` (``offdiag`` below): the truncated-Newton
    solves keep one pass over the basis per iteration (lfpsqp_projcg_tridiag).  One rank, no ball, no bounds (the one-pass form's limits)."""

    def __init__(self, ctx: Context, n: int, m: int, Jct: DeviceMatrix, b, kind: int, a, c=0.0, kappa: float = 1.0, **kw):
        assert kw.get("R2") is None and kw.get("xl") is None and kw.get("xu") is None, "chain objective: equalities only"
        assert kw.get("n_global", n) in (None, n), "chain objective: one rank (the couplings would cross the shard boundaries)"
        super().__init__(ctx, n, m, Jct, b, kind, a, c, **kw)
        from .projcg import TridiagonalOperator
        self.kappa = float(kappa)
        deg = np.full(n, 2.0 * self.kappa)
        deg[0] = deg[-1] = self.kappa if n > 1 else 0.0
        self._deg = ctx.vector(n, deg)
        self.offdiag = ctx.vector(n, np.full(n, -self.kappa))             # (entry n-1 is ignored)
        self._lap = TridiagonalOperator(0.0, self._deg, self.offdiag)     # kappa * L, L = the path graph's Laplacian
        self._tmp = ctx.vector(n)

    def f(self, x: DeviceVector) -> float:
        from .device import dot
        self._lap.mul_(self._tmp, x)
        return super().f(x) + 0.5 * dot(x, self._tmp)

    def grad_(self, g: DeviceVector, x: DeviceVector):
        from .device import axpby
        super().grad_(g, x)
        self._lap.mul_(self._tmp, x)
        axpby(1.0, self._tmp, 1.0, g)

    def diag_objective_(self, hx: DeviceVector, x: DeviceVector):
        from .device import axpby
        super().diag_objective_(hx, x)
        axpby(1.0, self._deg, 1.0, hx)

    def diag_(self, hx: DeviceVector, x: DeviceVector, lam: np.ndarray):
        from .device import axpby
        super().diag_(hx, x, lam)
        axpby(1.0, self._deg, 1.0, hx)


class SeparableElementwiseBox(SeparableLinearBallBox):
    """The device-resident problem class with NONLINEAR equality constraints: a separable objective
    (``kind`` / ``a`` / ``c`` as in :class:`SeparableLinearBallBox`) under ``cons``, an :class:`ElementwiseConstraints`
    (c(x) = A' phi(x) + qw x'x - b: the reference's sin and sphere test systems and their relatives), with optional box
    bounds.  f, grad!, c!, jac! and the diagonal Lagrangian Hessian phi_f''(x) + phi''(x) .* (A lam) + 2 qw'lam all run on the
    device, so `optimize` keeps the fused projected-CG path and nothing n-sized crosses PCIe in the loop."""

    def __init__(self, ctx: Context, cons, kind: int, a, c=0.0, xl=None, xu=None, n_global: Optional[int] = None):
        n, m = cons.Jct.n, cons.m_lin
        assert not cons.has_ball and cons.Jct.m == m
        self.ctx, self.n, self.m, self.xc = ctx, n, m, 0.0
        self.p = self.ploc = 0
        self.N, self.M = n, m
        self.Jct, self.R2, self.cons = cons.Jct, 0.0, cons
        self.xl = None if xl is None else np.asarray(xl, dtype=np.float64)
        self.xu = None if xu is None else np.asarray(xu, dtype=np.float64)
        self.n_global = n if n_global is None else n_global
        self.is_diagonal = True
        self.kind = int(kind)
        self.a_dev = None if np.isscalar(a) else ctx.vector(n, np.asarray(a, dtype=np.float64))
        self.c_dev = None if np.isscalar(c) else ctx.vector(n, np.asarray(c, dtype=np.float64))
        self.a0 = float(a) if np.isscalar(a) else 0.0
        self.c0 = float(c) if np.isscalar(c) else 0.0

    def diag_(self, hx: DeviceVector, x: DeviceVector, lam: np.ndarray):
        self._sep(2, x, hx)
        self.cons.hess_diag_(hx, x, lam)


# ------------------------------------------------------------------------------------------------
@dataclass
class Derivatives:
    """Analytic derivatives in the user's variables (the reference gets them by AD,
    src/autodiff_generators.jl -- out of scope).  All callables take HOST arrays."""
    grad_: Callable
    hess_lag_vec_: Callable
    jac_c_: Optional[Callable] = None
    jac_d_: Optional[Callable] = None


def optimize(*args, derivatives: Optional[Derivatives] = None, ctx: Optional[Context] = None, trace=None):
    """optimize(f, x0) / (f, c!, x0, m) / (f, c!, x0, xl, xu, m) / (f, c!, d!, x0, xl, xu, m, p) /
    (f, c!, d!, dl, du, x0, xl, xu, m, p) / (f, grad!, c!, jac!, hess_lag_vec!, x0, xl, xu, m) with an
    optional trailing LFPSQPParams -- host callables, device hot path (projcg!, retractions, tangent
    setup run on the GPU; every user call costs one n-vector PCIe round trip)."""
    args = list(args)
    param = LFPSQPParams()
    if args and isinstance(args[-1], LFPSQPParams):
        param = args.pop()
    if ctx is None:
        ctx = Context(0)
    k = len(args)
    if k == 9:
        f, grad_, c_, jac_, hlv_, x0, xl, xu, m = args
        return _host_core(ctx, f, grad_, c_, jac_, hlv_, x0, xl, xu, m, param, trace)
    if derivatives is None:
        raise NotImplementedError("the AD generators (src/autodiff_generators.jl) are out of scope; pass derivatives=Derivatives(...)")
    dv = derivatives
    if k == 2:
        f, x0 = args
        return _host_core(ctx, f, dv.grad_, None, None, dv.hess_lag_vec_, x0, None, None, 0, param, trace)
    if k == 4:
        f, c_, x0, m = args
        return _host_core(ctx, f, dv.grad_, c_, dv.jac_c_, dv.hess_lag_vec_, x0, None, None, m, param, trace)
    if k == 6:
        f, c_, x0, xl, xu, m = args
        return _host_core(ctx, f, dv.grad_, c_, dv.jac_c_ if m > 0 else None, dv.hess_lag_vec_, x0, xl, xu, m, param, trace)
    if k == 8:      # (f, c!, d!, x0, xl, xu, m, p)   d <= 0                         src/optimize.jl:83
        f, c_, d_, x0, xl, xu, m, p = args
        return _host_slack(ctx, f, c_, d_, -np.inf * np.ones(p), np.zeros(p), x0, xl, xu, m, p, param, dv, trace)
    if k == 10:     # (f, c!, d!, dl, du, x0, xl, xu, m, p)                          src/optimize.jl:13
        f, c_, d_, dl, du, x0, xl, xu, m, p = args
        return _host_slack(ctx, f, c_, d_, dl, du, x0, xl, xu, m, p, param, dv, trace)
    raise TypeError(f"no optimize method with {k} positional arguments")


def _host_slack(ctx, f, c_, d_, dl, du, x0, xl, xu, m, p, param, dv, trace):
    """src/optimize.jl:13-71: slack variables turn dl <= d(x) <= du into equalities d(x) - s = 0 with bounds on
    s; n -> n+p, m -> m+p; the result is truncated to the user's n (:68)."""
    if d_ is None or p == 0:
        return optimize(f, c_, x0, xl, xu, m, param, derivatives=dv, ctx=ctx, trace=trace)
    if not (len(dl) == len(du) == p):
        raise ValueError("Bound vectors dl and du must be of size p")
    x0 = np.asarray(x0, dtype=np.float64)
    n = len(x0)
    xl = -np.inf * np.ones(n) if xl is None else np.asarray(xl, dtype=np.float64)
    xu = np.inf * np.ones(n) if xu is None else np.asarray(xu, dtype=np.float64)
    x0_aux = np.empty(n + p)
    x0_aux[:n] = x0
    d_(x0_aux[n:], x0)
    xl_aux, xu_aux = np.concatenate([xl, dl]), np.concatenate([xu, du])

    def f_aux(x):
        return f(x[:n])

    def c_aux_(cval, x):
        if m > 0:
            c_(cval[:m], x[:n])
        d_(cval[m:m + p], x[:n])
        cval[m:m + p] -= x[n:n + p]
        return cval

    def grad_aux_(g, x):
        dv.grad_(g[:n], x[:n])
        g[n:] = 0.0

    def jac_aux_(J, cval, x):
        J[:, :] = 0.0
        if m > 0:
            dv.jac_c_(J[:m, :n], cval[:m], x[:n])
        dv.jac_d_(J[m:m + p, :n], cval[m:m + p], x[:n])
        cval[m:m + p] -= x[n:n + p]
        J[m:m + p, n:n + p] = -np.eye(p)

    def hlv_aux_(dest, src, x, lam):
        dv.hess_lag_vec_(dest[:n], src[:n], x[:n], lam)
        dest[n:] = 0.0

    x, obj, lam, ti = _host_core(ctx, f_aux, grad_aux_, c_aux_, jac_aux_, hlv_aux_, x0_aux, xl_aux, xu_aux, m + p, param, trace)
    return x[:n], obj, lam, ti


def _host_core(ctx, f, grad_, c_, jac_, hlv_, x0, xl, xu, m, param, trace):
    """Adapters: host callables -> the device-vector contract of optimize_core (no bounds + general
    Hessian: generic projcg path)."""
    x0 = np.asarray(x0, dtype=np.float64)
    n = len(x0)

    def f_dev(x):
        return float(f(x.download(n, 0)))

    def grad_dev(g, x):
        gh = np.zeros(n)
        grad_(gh, x.download(n, 0))
        g.upload(gh, 0)

    def jac_dev(Jct, cval, x):
        J = np.zeros((m, n), order='F')
        jac_(J, cval, x.download(n, 0))
        Jct.upload(np.asfortranarray(J.T))

    def hlv_dev(dest, src, x, lam):
        out = np.zeros(n)
        hlv_(out, src.download(n, 0), x.download(n, 0), lam.download(max(m, 1))[:m])
        dest.upload(out, 0)

    return optimize_core(f_dev, grad_dev, c_, jac_dev if m > 0 else None, hlv_dev, x0, xl, xu, m, param, ctx=ctx, trace=trace)
