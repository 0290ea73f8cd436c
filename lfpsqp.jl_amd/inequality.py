"""Bound manifolds on the device (reference src/inequality_helper.jl and
src/retractions.jl:451-500).  Names and argument order mirror the reference; arrays are
device buffers.  ``xaug = [x; y]`` vectors use the stacked layout of include/lfpsqp_hip.h:
x-half at [0, N), y-half at [hs, hs+N)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .device import Context, DeviceMatrix, DeviceVector


def half_stride(ctx: Context, N: int) -> int:
    return int(ctx.L.lfpsqp_half_stride(N))


class StackedVector(DeviceVector):
    """A length-2N ``[x; y]`` vector of the reference, both halves tile-aligned on the device."""

    def __init__(self, ctx: Context, N: int):
        self.N = int(N)
        self.hs = half_stride(ctx, N)
        super().__init__(ctx, self.hs + self.N)

    def upload2(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        assert host.size == 2 * self.N
        self.upload(host[:self.N], 0)
        self.upload(host[self.N:], self.hs)
        return self

    def download2(self) -> np.ndarray:
        return np.concatenate([self.download(self.N, 0), self.download(self.N, self.hs)])


class InequalityData:
    """InequalityData(xl, xu) (src/inequality_helper.jl:39-89), device-resident q, r, s, t."""

    def __init__(self, ctx: Context, xl, xu):
        xl = np.asarray(xl, dtype=np.float64)
        xu = np.asarray(xu, dtype=np.float64)
        if len(xl) != len(xu):
            raise ValueError("xl and xu are of different lengths")
        self.ctx, self.n = ctx, len(xl)
        self.q, self.r, self.s, self.t = (DeviceVector(ctx, self.n) for _ in range(4))
        dxl, dxu = ctx.vector(self.n, xl), ctx.vector(self.n, xu)
        ctx.check(ctx.L.lfpsqp_ineq_data_build(ctx.h, dxl.h, dxu.h, self.q.h, self.r.h, self.s.h, self.t.h))
        dxl.free()
        dxu.free()

    def _c(self):
        return _capi.IneqData(self.q.h, self.r.h, self.s.h, self.t.h, self.n)


def generate_initial_y_(xaug: StackedVector, idata: InequalityData):
    c = xaug.ctx
    d = idata._c()
    c.check(c.L.lfpsqp_generate_initial_y(c.h, xaug.h, C.byref(d)))
    return xaug


def calculate_h_(h: DeviceVector, xaug: StackedVector, idata: InequalityData, want_max: bool = True):
    """h[:N] = bound-constraint values; returns norm(h, Inf) (None if not wanted)."""
    c = xaug.ctx
    d = idata._c()
    hm = C.c_double()
    c.check(c.L.lfpsqp_calculate_h(c.h, h.h, xaug.h, C.byref(d), C.byref(hm) if want_max else None))
    return hm.value if want_max else None


class InequalityDecomp:
    """InequalityDecomp (src/inequality_helper.jl:10-19).  ``U`` of the reference (2N x M) is held
    as the N x M matrix ``Z`` plus the row scalings sx = Dy^2, sy = -Dx*Dy."""

    def __init__(self, ctx: Context, N: int, M: int, Jct: DeviceMatrix | None = None, Z: DeviceMatrix | None = None, factored: bool = False):
        self.ctx, self.N, self.M = ctx, N, M
        # (the driver hands in a basis allocated jointly with ProjCGWork, FINDINGS.md 6 -- or none at all: factored form, 5.3)
        self.Z = Z if (Z is not None or factored) else DeviceMatrix(ctx, N, M)
        self.Sigma = np.zeros(M)
        self.Vt = np.zeros((M, M), order='F')
        self.Dx, self.Dy, self.S, self.sx, self.sy = (DeviceVector(ctx, N) for _ in range(5))
        self.Jct = Jct if Jct is not None else DeviceMatrix(ctx, N, M)
        self.rank = M
        self.W = None                       # ksvd_'s small factor (Z == Jct @ W) when the driver keeps it
        self.Jsp = None                     # sparse twin of Jct's leading columns: projcg_ then runs on the nonzeros (lfpsqp_basis.SA)

    def basis_c(self):
        zh = self.Z.h if self.Z is not None else None        # None: the basis stays in factored form U = [sx; sy] .* (Jct W)
        if self.W is not None:
            return _capi.Basis(zh, self.rank, self.Dx.h, self.Dy.h, self.sx.h, self.sy.h, self.Jct.h, self.W.ctypes.data, None,
                               self.Jsp.h if self.Jsp is not None else None)
        return _capi.Basis(zh, self.rank, self.Dx.h, self.Dy.h, self.sx.h, self.sy.h)


def inequality_gradient_(idecomp: InequalityDecomp, xaug: StackedVector, idata: InequalityData):
    c = xaug.ctx
    d = idata._c()
    c.check(c.L.lfpsqp_inequality_gradient(c.h, xaug.h, C.byref(d), idecomp.Dx.h, idecomp.Dy.h, idecomp.S.h, idecomp.sx.h,
                                           idecomp.sy.h))


def y_retract_(xnewaug: StackedVector, xaug: StackedVector, idata: InequalityData):
    c = xaug.ctx
    d = idata._c()
    c.check(c.L.lfpsqp_y_retract(c.h, xnewaug.h, xaug.h, C.byref(d)))
    return xnewaug


class QCoeff:
    """A coefficient vector [w; t] of Q (length N + rank in the reference) as two device vectors."""

    def __init__(self, ctx: Context, N: int, m: int):
        self.w, self.t = DeviceVector(ctx, N), DeviceVector(ctx, max(m, 1))

    def fill(self, value: float):
        self.w.fill(value)
        self.t.fill(value)
        return self


class InequalityDecompProject:
    """Q = [[diag Dx; diag Dy], U[:, :rank]] (src/inequality_helper.jl:25-27, 161-212).
    Coefficient vectors [w; t] are kept as two device vectors: w (N) and t (rank)."""

    def __init__(self, idecomp: InequalityDecomp):
        self.idecomp = idecomp

    # -- the mul! protocol of the generic projcg path (coefficients are QCoeff objects)
    def new_coeff(self) -> "QCoeff":
        return QCoeff(self.idecomp.ctx, self.idecomp.N, self.idecomp.M)

    def mul_(self, dest, coeff: "QCoeff", a=None, b=None):
        if a is None:
            a, b = 1.0, 0.0
        self.mul_n(dest, coeff.w, coeff.t, a, b)
        return dest

    def adjoint(self):
        return _QAdjoint(self)

    @property
    def ncols(self):
        return self.idecomp.rank

    def _c(self):
        return self.idecomp.basis_c()

    def mul_t(self, w: DeviceVector, t: DeviceVector, v: StackedVector):
        """[w; t] = Q' v."""
        c = v.ctx
        b = self._c()
        c.check(c.L.lfpsqp_q_gemv_t(c.h, C.byref(b), v.h, w.h, t.h))

    def mul_n(self, y: StackedVector, w: DeviceVector | None, t: DeviceVector, alpha=1.0, beta=0.0):
        """y = alpha Q [w; t] + beta y."""
        c = y.ctx
        b = self._c()
        c.check(c.L.lfpsqp_q_gemv_n(c.h, C.byref(b), float(alpha), w.h if w is not None else None, t.h, float(beta), y.h))


class _QAdjoint:
    def __init__(self, q):
        self.q = q

    def mul_(self, coeff: "QCoeff", v, a=None, b=None):
        assert a is None
        self.q.mul_t(coeff.w, coeff.t, v)
        return coeff

    def adjoint(self):
        return self.q


class InequalityDecompOp:
    """The full constraint-Jacobian-transpose operator  [[diag(Dx.*S), Jct]; [diag(Dy.*S), 0]]  and its
    adjoint: the three mul! methods of InequalityDecomp / InequalityDecompAdjoint
    (src/inequality_helper.jl:215-271) -- what pcg!/ProjPenalty use as `fulljac` when bounds exist
    (src/retractions.jl:324).  It is the stacked-operator form with row scalings (1, 0), so it runs on the
    same kernels as the projection operator.  Coefficient vectors [v_h; v_c] are two device vectors."""

    def __init__(self, idecomp: InequalityDecomp):
        from .device import vmul
        self.idecomp = idecomp
        ctx, N = idecomp.ctx, idecomp.N
        self.DxS, self.DyS = DeviceVector(ctx, N), DeviceVector(ctx, N)
        self.ones = DeviceVector(ctx, N).fill(1.0)
        self.zeros = DeviceVector(ctx, N)
        self.refresh()

    def refresh(self):
        """Recompute Dx.*S, Dy.*S after inequality_gradient_ changed the decomposition."""
        from .device import vmul
        vmul(self.idecomp.Dx, self.idecomp.S, self.DxS)
        vmul(self.idecomp.Dy, self.idecomp.S, self.DyS)

    def _c(self):
        return _capi.Basis(self.idecomp.Jct.h, self.idecomp.Jct.m, self.DxS.h, self.DyS.h, self.ones.h, self.zeros.h)

    def mul_n(self, dest: StackedVector, v_h: DeviceVector, v_c: DeviceVector, a=1.0, b=0.0):
        """dest = a * idecomp * [v_h; v_c] + b * dest   (mul!(dest, idecomp, v[, a, b]), :215-251)."""
        c = dest.ctx
        bb = self._c()
        c.check(c.L.lfpsqp_q_gemv_n(c.h, C.byref(bb), float(a), v_h.h, v_c.h, float(b), dest.h))

    def mul_t(self, dest_h: DeviceVector, dest_c: DeviceVector, w: StackedVector):
        """[dest_h; dest_c] = idecomp' * w   (mul!(dest, idecomp', w), :254-271)."""
        c = w.ctx
        bb = self._c()
        c.check(c.L.lfpsqp_q_gemv_t(c.h, C.byref(bb), w.h, dest_h.h, dest_c.h))


def calculate_lambda_kkt_(lam_kkt: np.ndarray, lamy_kkt: DeviceVector, Qt_w: DeviceVector, Qt_t: DeviceVector,
                          idecomp: InequalityDecomp):
    """calculate_lambda_kkt!(lam_kkt, lamy_kkt, Qt_grad_f, idecomp) (src/inequality_helper.jl:286-308) with
    Qt_grad_f = [Qt_w (N); Qt_t (rank)] as produced by InequalityDecompProject.mul_t:
    lam = Vt' Sigma^-1 Qt_t (replicated m x m host algebra), lamy = (-Dx .* (Jct lam) + Qt_w) ./ S."""
    ctx = lamy_kkt.ctx
    m, rank = idecomp.M, idecomp.rank
    th = Qt_t.download(m)
    th[:rank] /= idecomp.Sigma[:rank]
    th[rank:m] = 0.0
    lam_kkt[:] = idecomp.Vt.T @ th
    lam_dev = ctx.vector(max(m, 1), lam_kkt if m else None)
    ctx.check(ctx.L.lfpsqp_calculate_lambda_y(ctx.h, idecomp.Jct.h, m, lam_dev.h, idecomp.Dx.h, idecomp.S.h, Qt_w.h, lamy_kkt.h))
    lam_dev.free()
    return lam_kkt, lamy_kkt


def augmented_hess_diag_(a: StackedVector, hx: DeviceVector, lamy_kkt: DeviceVector, idata: InequalityData):
    """The diagonal of augmented_hess_lag_vec! (src/inequality_helper.jl:144-158) for a diagonal Lagrangian
    Hessian hx: a = [hx + 2 lamy.*q ; 2 lamy.*s]."""
    ctx = a.ctx
    d = idata._c()
    ctx.check(ctx.L.lfpsqp_augmented_diag(ctx.h, hx.h, lamy_kkt.h, C.byref(d), a.h))
    return a
