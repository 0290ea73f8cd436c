"""ProjPenalty retraction and its inner pcg! (reference src/retractions.jl:179-246, 265-441),
the reference's DEFAULT retraction (do_project_retract = true).

Round-1 form: the reference's statement order on device vectors with the BLAS-1/2 primitives
(two passes over Jct per pcg! iteration, vector updates unfused).  With bounds the full Jacobian
operator  [[diag(Dx.*S), Jct]; [diag(Dy.*S), 0]]  (InequalityDecomp, src/inequality_helper.jl:215-271)
is the stacked-operator form of lfpsqp_q_gemv_t/_n with row scalings (1, 0), so it shares the
kernels of the projection operator.  Reference quirks are reproduced (BUG-COMPAT notes)."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _capi
from .device import Context, DeviceMatrix, DeviceVector, axpby, dot, gemv_n, gemv_t, nrm2, vmul, waxpby
from .inequality import InequalityData, InequalityDecomp, StackedVector, calculate_h_, inequality_gradient_


class ProjPenaltyWork:  # src/retractions.jl:21-33 (J itself is the shared device Jct)
    def __init__(self, ctx: Context, m: int, n: int, ineq: bool, against=None):
        """``against``: the constraint-gradient matrix the pcg! iteration streams -- its five n-vectors then come from ONE allocation
        chosen by the library's placement policy against that matrix (lfpsqp_vecs_alloc_placed, FINDINGS.md 6), as ProjCGWork's do."""
        mk = (lambda: StackedVector(ctx, n)) if ineq else (lambda: DeviceVector(ctx, n))
        if against is not None and ctx.options.placement_tries > 1:
            self.r, self.p, self.z, self.dx, self.g = ctx.vectors_placed(against, n, 5, stacked_N=n if ineq else None)
        else:
            self.r, self.p, self.z, self.dx, self.g = mk(), mk(), mk(), mk(), mk()
        self.tmp_m = DeviceVector(ctx, max(m, 1))
        self.cval_dev = DeviceVector(ctx, max(m, 1))
        self.ineq = ineq
        if ineq:
            self.tmp_w = DeviceVector(ctx, n)
            self.h = DeviceVector(ctx, n)
            self.DxS, self.DyS = DeviceVector(ctx, n), DeviceVector(ctx, n)
            self.ones = DeviceVector(ctx, n).fill(1.0)
            self.zeros = DeviceVector(ctx, n)
        # DeviceOptions.pp_precondition: the inner solves run lfpsqp_pcg_pre (one more n-vector; with bounds the rows of D0^-1)
        self.precondition = bool(getattr(ctx.options, "pp_precondition", False))
        if self.precondition:
            self.q = mk()
            if ineq:
                self.i11, self.i12, self.i22 = DeviceVector(ctx, n), DeviceVector(ctx, n), DeviceVector(ctx, n)

    def _c(self):
        g = lambda name: getattr(self, name).h if hasattr(self, name) else None
        return _capi.PPWork(self.r.h, self.p.h, self.z.h, self.dx.h, self.g.h, self.tmp_m.h, g("tmp_w"), g("h"), g("DxS"), g("DyS"),
                            g("ones"), g("zeros"), g("q"), g("i11"), g("i12"), g("i22"), 1 if self.precondition else 0)


class LazyProjPenaltyWork:
    """ProjPenaltyWork allocated on FIRST USE.  The driver builds its ProjPenalty method up front like the reference (src/optimize.jl:196-212) but
    most runs retract with Newton-Raphson and never touch it -- and its five n-vectors are placed by trial against Jct (Context.vectors_placed):
    72 ms of a 150 ms one-iteration run at n = 1e7, m = 128 (tools/profile_setup.py)."""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs, self._w = args, kwargs, None

    def _get(self):
        if self._w is None:
            self._w = ProjPenaltyWork(*self._args, **self._kwargs)
        return self._w

    def __getattr__(self, name):                       # (only reached for attributes this wrapper does not have itself)
        return getattr(self._get(), name)


@dataclass
class ProjPenalty:  # src/retractions.jl:35-49
    jac_: object
    U: object
    Sigma: np.ndarray
    Vt: np.ndarray
    rank: int
    mu0: float
    tol: float
    maxiter: int
    maxiter_pcg: int
    work: ProjPenaltyWork
    ineq: bool
    idecomp: InequalityDecomp
    idata: InequalityData | None


class _JacPlain:
    """J = Jct' (m x n): tmp = J p ; z = J' tmp + mu z."""

    def __init__(self, Jct: DeviceMatrix, work: ProjPenaltyWork, Jsp=None):
        self.Jct, self.w, self.Jsp = Jct, work, Jsp      # Jsp: optional SparseMatrix with the same entries

    def _basis(self):
        return _capi.Basis(self.Jct.h, self.Jct.m, None, None, None, None, None, None, self.Jsp.h if self.Jsp is not None else None)

    def apply(self, p):                       # mul!(tmp_m, J, p)       :221
        gemv_t(self.Jct, p, self.w.tmp_m)

    def apply_t(self, z, a, b, from_cval=False):   # mul!(z, J', tmp_m, a, b)   :222 / :369
        gemv_n(self.Jct, self.w.cval_dev if from_cval else self.w.tmp_m, z, a, b)


class _JacStacked:
    """fulljac = idecomp' (InequalityDecompAdjoint): tmp = [S.*(Dx.*px + Dy.*py); Jct' px], in the
    lfpsqp_basis form with row scalings (1, 0) on the work struct's vectors."""

    def __init__(self, idecomp: InequalityDecomp, work: ProjPenaltyWork, Jsp=None):
        self.idc, self.w, self.Jsp = idecomp, work, Jsp

    def _basis(self):
        w = self.w
        return _capi.Basis(self.idc.Jct.h, self.idc.Jct.m, w.DxS.h, w.DyS.h, w.ones.h, w.zeros.h, None, None,
                           self.Jsp.h if self.Jsp is not None else None)

    def refresh(self):
        vmul(self.idc.Dx, self.idc.S, self.w.DxS)
        vmul(self.idc.Dy, self.idc.S, self.w.DyS)

    def apply(self, p):
        c = p.ctx
        b = self._basis()
        c.check(c.L.lfpsqp_q_gemv_t(c.h, C.byref(b), p.h, self.w.tmp_w.h, self.w.tmp_m.h))

    def apply_t(self, z, a, b_, from_cval=False):
        c = z.ctx
        b = self._basis()
        wv = self.w.h if from_cval else self.w.tmp_w
        tv = self.w.cval_dev if from_cval else self.w.tmp_m
        c.check(c.L.lfpsqp_q_gemv_n(c.h, C.byref(b), float(a), wv.h, tv.h, float(b_), z.h))


def no_precondition(z, r):  # :259-263
    z.copy_from(r)
    return z


def proj_precondition_(z, r, mu, U, Sigma, rank, tmp_m):
    """proj_precondition!(z, r, mu, U, Sigma, rank, tmp_m) (src/retractions.jl:248-257): the exact inverse of
    U Sigma^2 U' + mu I applied to r.  Dead code on the reference's live path (its call is commented out at
    :374) and untested there; provided for completeness on the kgemv! primitives.  U: DeviceMatrix."""
    z.copy_from(r)
    gemv_t(U, r, tmp_m, ncols=rank)                                   # kgemv!('T', rank, 1, U, r, 0, tmp_m)
    t = tmp_m.download(rank)
    s2 = np.asarray(Sigma[:rank]) ** 2
    t *= s2 / (mu + s2)
    tmp_m.upload(t)
    gemv_n(U, tmp_m, z, -1.0 / mu, 1.0 / mu, ncols=rank)             # kgemv!('N', rank, -1/mu, U, tmp_m, 1/mu, z)
    return z


class ProjPrecondition:
    """M! = (z, r) -> proj_precondition!(z, r, mu, U, Sigma, rank, tmp_m) (the line commented out at src/retractions.jl:374) as an
    object, for a basis known through its generator: U = Jct W (ksvd_'s W).  pcg_ recognises it and runs the whole solve fused on the
    device (lfpsqp_pcg_pre with K = mu W diag(s^2 / (mu + s^2)) W'); called as a function it is the statement-by-statement form."""

    def __init__(self, Jct: DeviceMatrix, W: np.ndarray, Sigma, rank: int, q: DeviceVector, Z: DeviceMatrix | None = None):
        self.Jct, self.W, self.Sigma, self.rank, self.q, self.Z = Jct, np.asarray(W), np.asarray(Sigma, dtype=np.float64), int(rank), q, Z
        self.mu = None
        self._tmp = None

    def K(self, mu: float) -> np.ndarray:
        s2 = self.Sigma[:self.rank] ** 2
        Wr = self.W[:, :self.rank]
        return np.asfortranarray(mu * (Wr * (s2 / (mu + s2))) @ Wr.T)

    def __call__(self, z, r):
        if self.Z is None:
            raise ValueError("the statement-by-statement form needs the materialised basis Z")
        if self._tmp is None:
            self._tmp = DeviceVector(z.ctx, max(self.rank, 1))
        return proj_precondition_(z, r, self.mu, self.Z, self.Sigma, self.rank, self._tmp)


def pcg_(mu, J, M_, x, r, p, z, tmp_m, tol, maxiter):
    """pcg!(mu, J, M!, x, r, p, z, tmp_m, tol, maxiter) (src/retractions.jl:179-246) -> (flag, i).
    J is one of the operator adapters above (tmp_m lives inside it).  With the reference's live
    preconditioner (no_precondition, :375) the whole solve is ONE C call (lfpsqp_pcg, fused
    kernels); any other M! runs the statement-by-statement loop below on the device primitives."""
    if M_ is no_precondition:
        ctx = x.ctx
        flag = C.c_int()
        iters = _capi.c_i64()
        b = J._basis()
        w = J.w
        ctx.check(ctx.L.lfpsqp_pcg(ctx.h, float(mu), C.byref(b), x.h, r.h, p.h, z.h, w.tmp_w.h if hasattr(w, "tmp_w") else None,
                                   w.tmp_m.h, float(tol), int(maxiter), C.byref(flag), C.byref(iters)))
        return flag.value, iters.value
    if isinstance(M_, ProjPrecondition) and isinstance(J, _JacPlain) and J.Jsp is None:
        ctx = x.ctx
        flag = C.c_int()
        iters = _capi.c_i64()
        b = J._basis()
        K = M_.K(float(mu))
        pc = _capi.PcgPrecond(K.ctypes.data, None, None, None, M_.q.h)
        rc = ctx.L.lfpsqp_pcg_pre(ctx.h, float(mu), C.byref(b), C.byref(pc), x.h, r.h, p.h, z.h, float(tol), int(maxiter), C.byref(flag),
                                  C.byref(iters))
        if rc != -5:                               # (LFPSQP_ERR_UNSUPPORTED: no one-pass kernel for this shape -> the loop below)
            ctx.check(rc)
            return flag.value, iters.value
    if isinstance(M_, ProjPrecondition):
        M_.mu = float(mu)
    norm_res = math.inf
    rho = 1.0
    p.fill(0.0)
    i = 0
    while norm_res > tol and i < maxiter:
        M_(z, r)                                   # :209
        rho_prev = rho
        rho = dot(z, r)                            # :213
        beta = rho / rho_prev                      # :216
        waxpby(1.0, z, beta, p, p)                 # :217  p = z + beta p
        z.copy_from(p)                             # :220
        J.apply(p)                                 # :221
        J.apply_t(z, 1.0, mu)                      # :222
        alpha = rho / dot(p, z)                    # :227
        axpby(alpha, p, 1.0, x)                    # :232
        axpby(-alpha, z, 1.0, r)                   # :233
        norm_res = nrm2(r)                         # :235
        i += 1
    flag = 1 if i == maxiter else 0                # BUG-COMPAT :240-243
    return flag, i


def _call_c(c_, cval, xnew, n):
    if hasattr(c_, "c_"):
        c_.c_(cval, xnew)
    else:
        c_(cval, xnew.download(n, 0))


def retract_pp(cval, xnew, c_, xtilde, x, method: ProjPenalty):
    """retract!(cval, xnew, c!, xtilde, x, method::ProjPenalty) (src/retractions.jl:265-441) -- ONE C call
    (lfpsqp_retract_pp); c! / jac! are device-resident constraints or host callables behind trampolines."""
    ctx = x.ctx
    idecomp, idata = method.idecomp, method.idata
    m = len(method.Sigma)
    n = idecomp.N
    flag, iters, pcg_iters = C.c_int(), _capi.c_i64(), _capi.c_i64()
    wc = method.work._c()
    idc = idata._c() if method.ineq else None
    keep = []
    from .retractions import DeviceConstraints
    if isinstance(c_, DeviceConstraints):      # device-resident c!/jac! (the matching jac! is c_.jac_)
        cons = c_._c()
        keep.append(cons)
        cons_p, cfun, jacfun = C.byref(cons), _capi.CFUN(), _capi.JACFUN()
    else:
        from .device import DeviceMatrix as DM

        def borrow_vec(handle, length):
            v = DeviceVector.__new__(DeviceVector)
            v.ctx, v.n, v.h = ctx, length, C.c_void_p(handle)
            return v

        # The vector handed to a callback belongs to the library: the wrapper must lose the handle on EVERY path (a
        # DeviceVector that still holds it would free it when collected).
        def c_tramp(user, xh, cval_ptr):
            v = borrow_vec(xh, x.n)
            try:
                out = np.ctypeslib.as_array(cval_ptr, shape=(m,))
                if hasattr(c_, "c_"):
                    c_.c_(out, v)
                else:
                    c_(out, v.download(n, 0))
                return 0
            except Exception as e:   # never unwind through C
                print("c! callback failed:", repr(e))
                return 1
            finally:
                v.h = None

        def j_tramp(user, xh, jh, cval_ptr):
            v = borrow_vec(xh, x.n)
            try:
                out = np.ctypeslib.as_array(cval_ptr, shape=(m,))
                method.jac_(idecomp.Jct, out, v)        # the library passes back the same Jct it was given
                return 0
            except Exception as e:
                print("jac! callback failed:", repr(e))
                return 1
            finally:
                v.h = None
        cfun, jacfun = _capi.CFUN(c_tramp), _capi.JACFUN(j_tramp)
        keep += [cfun, jacfun, c_tramp, j_tramp]
        cons_p = None
    ctx.check(ctx.L.lfpsqp_retract_pp(ctx.h, cons_p, cfun, jacfun, None, idecomp.Jct.h, m, C.byref(idc) if idc is not None else None,
                                      idecomp.Dx.h if method.ineq else None, idecomp.Dy.h if method.ineq else None,
                                      idecomp.S.h if method.ineq else None, xtilde.h, x.h, xnew.h, float(method.mu0), float(method.tol),
                                      int(method.maxiter), int(method.maxiter_pcg), C.byref(wc), cval.ctypes.data_as(_capi.PD),
                                      C.byref(flag), C.byref(iters), C.byref(pcg_iters)))
    del keep
    return flag.value, iters.value, pcg_iters.value
