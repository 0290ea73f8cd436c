"""projcg! (reference src/projcg.jl) on the device.

``projcg_(x, lam, A, U, b, c, tol=, maxit=, work=)`` keeps the reference's signature and
return value ``(i, nr)`` and, like Julia's method dispatch, picks an implementation
from the operator types:

  * ``A`` a :class:`DiagOperator` and ``U`` a :class:`DeviceBasis`  -> one C call
    (``lfpsqp_projcg``): the fused three-kernel iteration, scalars on the device;
  * anything else supporting the ``mul_`` protocol -> the generic loop below, the
    reference's statement order on device vectors with the BLAS-1/2 primitives
    (what a Julia host gets by defining ``mul!`` on device arrays).
"""
from __future__ import annotations

import ctypes as C
import math

from . import _capi
from .device import (Context, DeviceMatrix, DeviceVector, axpby, dot, gemv_n, gemv_t, nrm2, vmul, waxpby)

WANT_LAMBDA = 1
RESUME = 2
START_GIVEN = 4
START_PROJECTED = 8


class DiagOperator:
    """A = a0*I + diag(dg): device-resident Lagrangian-Hessian action (lfpsqp_diag_op)."""

    def __init__(self, a0: float = 0.0, dg: DeviceVector | None = None):
        self.a0, self.dg = float(a0), dg

    def _c(self):
        return _capi.DiagOp(self.a0, self.dg.h if self.dg is not None else None)

    # mul! protocol (generic path / tests)
    def mul_(self, dest: DeviceVector, v: DeviceVector, a=None, b=None):
        if a is None:
            a, b = 1.0, 0.0
        if self.dg is None:
            return waxpby(a * self.a0, v, b, dest, dest)
        tmp = DeviceVector(dest.ctx, dest.n)
        vmul(self.dg, v, tmp)
        axpby(self.a0, v, 1.0, tmp)
        waxpby(a, tmp, b, dest, dest)
        tmp.free()
        return dest

    def adjoint(self):
        return self


class LowRankOperator:
    """A = a0*I + diag(dg) + V diag(sigma) V' (lfpsqp_lowrank_op): a diagonal Hessian with k <= 8 coupling directions.  On a
    :class:`DeviceBasis` projcg_ runs it on the fused ONE-pass iteration (lfpsqp_projcg_lowrank); the ``mul_`` protocol below is the
    statement-by-statement form (the generic loop / lfpsqp_projcg_op make two passes per iteration with it)."""

    def __init__(self, a0: float, dg: DeviceVector | None, V: DeviceMatrix, k: int | None = None, sigma=None):
        import numpy as np
        self.a0, self.dg, self.V = float(a0), dg, V
        self.k = V.m if k is None else int(k)
        assert 0 <= self.k <= min(V.m, 8)
        self.sigma = None if sigma is None else np.ascontiguousarray(sigma, dtype=np.float64)
        assert self.sigma is None or self.sigma.size >= self.k
        self._t = None

    def _c(self):
        return _capi.LowRankOp(self.a0, self.dg.h if self.dg is not None else None, self.V.h, self.k,
                               self.sigma.ctypes.data if self.sigma is not None else None)

    def mul_(self, dest: DeviceVector, v: DeviceVector, a=None, b=None):
        import numpy as np
        if a is None:
            a, b = 1.0, 0.0
        ctx = dest.ctx
        tmp = DeviceVector(ctx, dest.n)
        if self.dg is not None:
            vmul(self.dg, v, tmp)
            axpby(self.a0, v, 1.0, tmp)
        else:
            waxpby(self.a0, v, 0.0, v, tmp)
        if self.k > 0:
            if self._t is None:
                self._t = DeviceVector(ctx, self.k)
            gemv_t(self.V, v, self._t, self.k)
            if self.sigma is not None:
                self._t.upload(self._t.download(self.k) * self.sigma[:self.k])
            gemv_n(self.V, self._t, tmp, 1.0, 1.0, self.k)
        waxpby(a, tmp, b, dest, dest)
        tmp.free()
        return dest

    def adjoint(self):
        return self


class TridiagonalOperator:
    """(A v)_i = (a0 + dg_i) v_i + off_{i-1} v_{i-1} + off_i v_{i+1} (lfpsqp_tridiag_op): a diagonal Hessian plus nearest-neighbour
    couplings.  ``off``: DeviceVector of length n (entry i couples rows i and i+1; the last entry is ignored).  On a :class:`DeviceBasis`
    projcg_ runs it on the fused ONE-pass iteration (lfpsqp_projcg_tridiag); ``mul_`` is the operator on its own (lfpsqp_tridiag_mul), which
    the generic loop / lfpsqp_projcg_op use -- two passes over the basis per iteration."""

    def __init__(self, a0: float, dg: DeviceVector | None, off: DeviceVector):
        self.a0, self.dg, self.off = float(a0), dg, off
        self._tmp = None

    def _c(self):
        return _capi.TridiagOp(self.a0, self.dg.h if self.dg is not None else None, self.off.h)

    def mul_(self, dest: DeviceVector, v: DeviceVector, a=None, b=None):
        ctx = dest.ctx
        a_c = self._c()
        if a is None:
            ctx.check(ctx.L.lfpsqp_tridiag_mul(ctx.h, C.byref(a_c), v.h, dest.h))
            return dest
        if self._tmp is None or self._tmp.n != dest.n:
            self._tmp = DeviceVector(ctx, dest.n)
        ctx.check(ctx.L.lfpsqp_tridiag_mul(ctx.h, C.byref(a_c), v.h, self._tmp.h))
        waxpby(a, self._tmp, b, dest, dest)
        return dest

    def adjoint(self):
        return self


class DeviceBasis:
    """Orthonormal U = Z[:, :ncols] (``view(U, :, 1:rank)``, src/optimize.jl:370).
    ``Z = None`` with ``generator = (A, W)``: the basis in FACTORED form U = A W -- never materialised; projcg_, the Newton retraction and
    the projections stream A and apply the small factor W on the side (lfpsqp_basis.Z == NULL, FINDINGS.md 5.3)."""

    def __init__(self, Z: DeviceMatrix | None, ncols: int | None = None, generator=None, sparse=None):
        self.Z = Z
        assert Z is not None or generator is not None
        self.ncols = (Z.m if Z is not None else generator[1].shape[1]) if ncols is None else int(ncols)
        self.generator = generator          # optional (A, W) with Z == A @ W (ksvd_'s W): lfpsqp_basis.A / .W
        self.sparse = sparse                # optional SparseMatrix twin of A's leading columns: projcg_ runs on the nonzeros (lfpsqp_basis.SA)

    @property
    def ctx(self):
        return self.Z.ctx if self.Z is not None else self.generator[0].ctx

    def _c(self):
        zh = self.Z.h if self.Z is not None else None
        if self.generator is not None:
            A, W = self.generator
            return _capi.Basis(zh, self.ncols, None, None, None, None, A.h, W.ctypes.data, None,
                               self.sparse.h if self.sparse is not None else None)
        return _capi.Basis(zh, self.ncols, None, None, None, None)

    def _factored(self):
        return self.generator is not None and (self.sparse is not None or self.Z is None)

    def mul_(self, dest, v, a=None, b=None):
        if a is None:
            a, b = 1.0, 0.0
        if self._factored():                # U t = A (W t): on the nonzeros (lfpsqp_basis.SA) or over the dense generator (Z == NULL)
            c = self.ctx
            bs = self._c()
            c.check(c.L.lfpsqp_q_gemv_n(c.h, C.byref(bs), float(a), None, v.h, float(b), dest.h))
            return dest
        return gemv_n(self.Z, v, dest, a, b, self.ncols)

    def adjoint(self):
        return _BasisAdjoint(self)


class _BasisAdjoint:
    def __init__(self, basis):
        self.basis = basis

    def mul_(self, dest, v, a=None, b=None):
        assert a is None
        if self.basis._factored():          # U'v = W'(A'v)
            c = self.basis.ctx
            bs = self.basis._c()
            c.check(c.L.lfpsqp_q_gemv_t(c.h, C.byref(bs), v.h, None, dest.h))
            return dest
        return gemv_t(self.basis.Z, v, dest, self.basis.ncols)

    def adjoint(self):
        return self.basis


class ProjCGWork:
    """ProjCGWork(n, m) (src/projcg.jl:1-11); only g, d, rp, Utr exist on the device."""

    def __init__(self, ctx: Context, n: int, m: int, stacked_N: int | None = None, against=None, extra: int = 0):
        """n = length of the n-vectors on this rank; with bounds pass stacked_N = N and the
        vectors get the stacked [x | gap | y] layout (length hs + N).
        ``against`` (DeviceMatrix): the basis these vectors will be streamed with -- they then come from one placement-tuned
        allocation (lfpsqp_vecs_alloc_placed, FINDINGS.md 6), together with ``extra`` more vectors of the same kind left in
        ``self.placed_extra`` (in the trial, the first of them plays the operator diagonal: use it for DiagOperator).
        ``against = ("new", rows, m)``: the basis matrix is allocated here as well, jointly with the vectors (every pair of candidate
        allocations tried, lfpsqp_basis_work_alloc_placed), and left in ``self.basis``."""
        self.placed_extra = []
        self.basis = None
        if isinstance(against, tuple):              # ("new", rows, m): allocate the basis too, jointly (lfpsqp_basis_work_alloc_placed) -> self.basis
            self.basis, vs = ctx.basis_and_vectors_placed(against[1], against[2], 3 + extra, stacked_N=stacked_N)
            against = None
            self.g, self.d = vs[0], vs[1]
            self.placed_extra = vs[2:2 + extra]
            self.rp = vs[2 + extra]
        elif against is not None:
            vs = ctx.vectors_placed(against, n, 3 + extra, stacked_N=stacked_N)
            self.g, self.d = vs[0], vs[1]
            self.placed_extra = vs[2:2 + extra]
            self.rp = vs[2 + extra]
        elif stacked_N is not None:
            from .inequality import StackedVector
            self.g, self.d, self.rp = (StackedVector(ctx, stacked_N) for _ in range(3))
        else:
            self.g, self.d, self.rp = (DeviceVector(ctx, n) for _ in range(3))
        # (3 m + 8: room behind the m coefficients for the sums lfpsqp_tangent_step parks for LFPSQP_PROJCG_START_PROJECTED)
        self.Utr = DeviceVector(ctx, 3 * max(m, 1) + 8)
        # extra scratch for a general operator A (lfpsqp_projcg_op's Av) and for the all-Python loop, allocated on demand
        self.Av = None
        self._extra = None

    def _c(self):
        return _capi.ProjCGWorkC(self.g.h, self.d.h, self.rp.h, self.Utr.h)


def projcg_(x: DeviceVector, lam: DeviceVector | None, A, U, b: DeviceVector, c: DeviceVector | None,
            tol: float = 1e-6, maxit: int | None = None, work: ProjCGWork | None = None, n_global: int | None = None,
            want_lambda: bool = True, resume: bool = False, start_given: bool = False, start_projected: bool = False):
    """``resume=True`` (device path only): ``maxit`` MORE iterations of the solve the previous call left at its iteration
    limit (LFPSQP_PROJCG_RESUME); the returned count runs from the start of the solve.
    ``start_given=True`` (device path, factored plain basis, ``c`` None): ``work.rp`` holds r0 = -b and ``work.Utr`` holds U'r0, left there
    by lfpsqp_tangent_step -- the solve skips its own residual pass (LFPSQP_PROJCG_START_GIVEN).
    ``start_projected=True``: the initial projection itself was made by lfpsqp_tangent_step (LFPSQP_TANGENT_INIT_PROJCG): ``work.g``, ``work.d``
    and the sums parked in ``work.Utr`` are the state the first iteration starts from (LFPSQP_PROJCG_START_PROJECTED)."""
    ctx = x.ctx
    n = b.n
    m = U.ncols if hasattr(U, "ncols") else (c.n if c is not None else 0)
    if n_global is None:
        n_global = n
    if maxit is None:
        maxit = n_global + m
    from .inequality import InequalityDecompProject
    stacked = isinstance(U, InequalityDecompProject)
    if stacked and n_global == n:
        n_global = 2 * U.idecomp.N          # the reference's length(b)
    if work is None:
        work = ProjCGWork(ctx, n, m, U.idecomp.N if stacked else None)
    if isinstance(A, DiagOperator) and (isinstance(U, DeviceBasis) or stacked):
        iters = _capi.c_i64()
        nr = C.c_double()
        a_c, u_c, w_c = A._c(), U._c(), work._c()
        flags = ((WANT_LAMBDA if (want_lambda and lam is not None) else 0) | (RESUME if resume else 0) | (START_GIVEN if start_given else 0)
                 | (START_PROJECTED if start_projected else 0))
        ctx.check(ctx.L.lfpsqp_projcg(ctx.h, x.h, lam.h if lam is not None else None, C.byref(a_c), C.byref(u_c), b.h,
                                      c.h if c is not None else None, float(tol), int(maxit), int(n_global), flags,
                                      C.byref(w_c), C.byref(iters), C.byref(nr)))
        return iters.value, nr.value
    if isinstance(A, LowRankOperator) and isinstance(U, DeviceBasis) and not stacked and not (resume or start_given or start_projected) and getattr(A, "fused", True):
        iters = _capi.c_i64()
        nr = C.c_double()
        a_c, u_c, w_c = A._c(), U._c(), work._c()
        flags = WANT_LAMBDA if (want_lambda and lam is not None) else 0
        rc = ctx.L.lfpsqp_projcg_lowrank(ctx.h, x.h, lam.h if lam is not None else None, C.byref(a_c), C.byref(u_c), b.h,
                                         c.h if c is not None else None, float(tol), int(maxit), int(n_global), flags,
                                         C.byref(w_c), C.byref(iters), C.byref(nr))
        if rc != -5:                 # LFPSQP_ERR_UNSUPPORTED (a shape without the one-pass iteration): the callback path below
            ctx.check(rc)
            return iters.value, nr.value
    if isinstance(A, TridiagonalOperator) and isinstance(U, DeviceBasis) and not stacked and not (resume or start_projected) and getattr(A, "fused", True):
        if getattr(work, "Av", None) is None:
            work.Av = DeviceVector(ctx, n)
        iters = _capi.c_i64()
        nr = C.c_double()
        a_c, u_c, w_c = A._c(), U._c(), work._c()
        flags = (WANT_LAMBDA if (want_lambda and lam is not None) else 0) | (START_GIVEN if start_given else 0)
        rc = ctx.L.lfpsqp_projcg_tridiag(ctx.h, x.h, lam.h if lam is not None else None, C.byref(a_c), work.Av.h, C.byref(u_c), b.h,
                                         c.h if c is not None else None, float(tol), int(maxit), int(n_global), flags,
                                         C.byref(w_c), C.byref(iters), C.byref(nr))
        if rc != -5 or start_given:  # LFPSQP_ERR_UNSUPPORTED (no one-pass iteration for this shape / more than one rank): the callback path below
            ctx.check(rc)
            return iters.value, nr.value
    if resume or start_given or start_projected:
        raise ValueError("resume / start_given / start_projected are features of the device-resident projcg loop")
    if (isinstance(U, DeviceBasis) or stacked) and hasattr(A, "mul_"):
        # general A (the LinearMap case, src/optimize.jl:228-230) on the C loop: lfpsqp_projcg_op calls back once per iteration
        # for A*d; the products stay on the device and nothing in the loop waits for it (no per-dot host round trips)
        if getattr(work, "Av", None) is None:
            work.Av = work.g.__class__(ctx, work.g.N) if hasattr(work.g, "N") else DeviceVector(ctx, n)
        Av = work.Av
        known = {x.h.value: x, work.d.h.value: work.d}

        def tramp(user, src_h, dst_h):
            try:
                A.mul_(Av, known[src_h])
                return 0
            except Exception as e:       # never unwind through C
                print("operator callback failed:", repr(e))
                return 1
        cb = _capi.OPFUN(tramp)
        iters = _capi.c_i64()
        nr = C.c_double()
        u_c, w_c = U._c(), work._c()
        flags = WANT_LAMBDA if (want_lambda and lam is not None) else 0
        ctx.check(ctx.L.lfpsqp_projcg_op(ctx.h, x.h, lam.h if lam is not None else None, cb, None, Av.h, C.byref(u_c), b.h,
                                         c.h if c is not None else None, float(tol), int(maxit), int(n_global), flags,
                                         C.byref(w_c), C.byref(iters), C.byref(nr)))
        del cb
        return iters.value, nr.value
    return _projcg_generic(x, lam, A, U, b, c, tol, maxit, work, n_global, m)


def _projcg_generic(x, lam, A, U, b, c, tol, maxit, work, n_global, m):
    """src/projcg.jl:55-120 statement by statement on device vectors."""
    ctx = x.ctx
    n = b.n
    if work._extra is None:
        mk = (lambda: g0.__class__(ctx, g0.N)) if hasattr(work.g, "N") else (lambda: DeviceVector(ctx, n))
        g0 = work.g
        work._extra = (mk(), mk())
    Ad, r = work._extra
    g, d, rp = work.g, work.d, work.rp
    # coefficient vectors are opaque to the loop: a device m-vector, or a [w; t] pair for the bound operator Q
    Utr = U.new_coeff() if hasattr(U, "new_coeff") else work.Utr
    Ut = U.adjoint()
    if c is not None:
        U.mul_(x, c)
    else:
        x.fill(0.0)
    r.copy_from(b)
    A.mul_(r, x, 1.0, -1.0)
    g.copy_from(r)
    Ut.mul_(Utr, r)
    U.mul_(g, Utr, -1.0, 1.0)
    r.copy_from(g)
    waxpby(-1.0, g, 0.0, g, d)
    i = 0
    nr = math.inf
    while i < min(maxit, n_global + m):
        i += 1
        A.mul_(Ad, d)
        dAd = dot(d, Ad)
        if dAd <= 0:
            waxpby(1.0 / nrm2(d), d, 0.0, d, x)
            if lam is not None:
                lam.fill(math.nan)
            return i, math.inf
        rg = dot(r, g)
        if rg <= 0:
            break
        alpha = rg / dAd
        axpby(alpha, d, 1.0, x)
        waxpby(1.0, r, alpha, Ad, rp)
        g.copy_from(rp)                      # gp lives in g
        Ut.mul_(Utr, rp)
        U.mul_(g, Utr, -1.0, 1.0)
        beta = dot(rp, g) / rg
        waxpby(beta, d, -1.0, g, d)
        r.copy_from(g)
        nr = nrm2(g)
        if nr < tol:
            break
    r.copy_from(b)
    A.mul_(r, x, -1.0, 1.0)
    if lam is not None:
        Ut.mul_(lam, r)
    return i, nr
