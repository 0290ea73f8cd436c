"""Retractions (reference src/retractions.jl) on the device.

``retract_(cval, xnew, c_, xtilde, x, method)`` dispatches on the method type like the
reference's ``retract!``: Euclidean / YRetract / NR / ProjPenalty.  ``c_`` is either a
:class:`DeviceConstraints` (device-resident c!, no PCIe traffic) or a Python callable
``c_(cval, x_host)`` (arbitrary user code: x is downloaded for every evaluation)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi
from .device import DeviceMatrix, DeviceVector
from .inequality import InequalityData, y_retract_


class DeviceConstraints:
    """c(x) = [J x - b ; sum_{i<n_x} x_i^2 - R2 - x[slack_row]] with Jct resident on the device
    (lfpsqp_constraints).  Callable as c!(cval, x) and jac!(Jct, cval, x)."""

    def __init__(self, Jct: DeviceMatrix, m_lin: int, b, has_ball: bool = False, R2: float = 0.0, n_x: int | None = None,
                 slack_row: int = -1, Jsp=None):
        """``Jsp`` (optional SparseMatrix with the entries of Jct[:, :m_lin]): c! and the inner solves of the ProjPenalty
        retraction stream its nonzeros instead of the dense block."""
        self.Jct, self.m_lin = Jct, int(m_lin)
        self.Jsp = Jsp
        self.b = np.ascontiguousarray(b, dtype=np.float64) if m_lin else np.zeros(1)
        self.has_ball, self.R2 = bool(has_ball), float(R2)
        self.n_x = Jct.n if n_x is None else int(n_x)
        self.slack_row = int(slack_row)
        self.m = self.m_lin + (1 if has_ball else 0)

    def _c(self):
        return _capi.Constraints(self.Jct.h, self.m_lin, self.b.ctypes.data, 1 if self.has_ball else 0, self.R2, self.n_x,
                                 self.slack_row, self.Jsp.h if self.Jsp is not None else None, self._ew_ptr())

    def _ew_ptr(self):
        return None

    def c_(self, cval: np.ndarray, x: DeviceVector):
        ctx = x.ctx
        cc = self._c()
        ctx.check(ctx.L.lfpsqp_constraints_eval(ctx.h, C.byref(cc), x.h, cval.ctypes.data_as(_capi.PD)))
        return cval

    __call__ = c_

    def jac_(self, Jct: DeviceMatrix, cval: np.ndarray, x: DeviceVector, evaluate: bool = True):
        """``evaluate=False``: the gradients only (cval untouched) -- for a caller that holds c(x) already."""
        ctx = x.ctx
        cc = self._c()
        ctx.check(ctx.L.lfpsqp_constraints_jac(ctx.h, C.byref(cc), x.h, Jct.h, cval.ctypes.data_as(_capi.PD) if evaluate else None))
        return cval


    def hess_diag_(self, hx: DeviceVector, x: DeviceVector, lam: np.ndarray):
        """hx += the diagonal of sum_j lam_j grad^2 c_j(x) (lfpsqp_constraints_hess_diag): the ball term for this class,
        phi'' .* (A lam) and the quadratic term for ElementwiseConstraints."""
        ctx = x.ctx
        cc = self._c()
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert lam.size >= self.m
        ctx.check(ctx.L.lfpsqp_constraints_hess_diag(ctx.h, C.byref(cc), x.h, lam.ctypes.data_as(_capi.PD), hx.h))
        return hx


KIND_ID, KIND_SIN, KIND_SQUARE = 0, 1, 2


class ElementwiseConstraints(DeviceConstraints):
    """The device-resident NONLINEAR constraint class (lfpsqp_elementwise):

        c(x) = A' phi(x) + qw * sum_{i < n_x} x_i^2 - b,      phi_i in {t, sin t, t^2} per variable (``kind``),

    with constraint gradients Jct(x) = diag(phi'(x)) A + 2 x qw' refreshed in place by ``jac_`` (the reference's jac!,
    src/autodiff_generators.jl:60-66) and a DIAGONAL Lagrangian-Hessian term (``hess_diag_``; hess_lag_vec!, :80-104).
    Covers the reference's own nonlinear test systems (test/test_retractions.jl:1-54, see :func:`sin_system_constraints` /
    :func:`sphere_system_constraints`).  ``A``: DeviceMatrix (n x m), or a SparseMatrix -- then c!, jac!, the tangent setup, the
    Newton steps and the inner solves of ProjPenalty all stream the nonzeros.  ``Jct`` (n x (m + ball)) is the working matrix.

    ``stream`` (dense ``A`` with a ``kind`` and / or ``qw``, no ball; default: whenever the library's one-pass kernels cover the shape): the
    gradients are STREAMED -- ``Jct`` is a view diag(rs) A + u qw' of ``A`` itself (:meth:`DeviceMatrix.view`), jac! refreshes the n-vectors
    rs = phi'(x) and u = 2 x instead of rewriting an n x m matrix, and there is no second matrix in memory."""

    def __init__(self, ctx, A, b, kind=None, qw=None, Jct: DeviceMatrix | None = None, has_ball: bool = False, R2: float = 0.0,
                 n_x: int | None = None, slack_row: int = -1, stream: bool | None = None):
        from .device import SparseMatrix
        sparse = isinstance(A, SparseMatrix)
        n, m = A.n, A.m
        can_stream = not sparse and (kind is not None or qw is not None) and not has_ball and Jct is None and m >= 1
        if stream is None:
            stream = can_stream and ctx.factored_basis_supported(A, None)
        elif stream and not can_stream:
            raise ValueError("streamed gradients need a dense A with a kind or a quadratic term, no ball and no caller-owned Jct")
        self.streamed = bool(stream)
        if self.streamed:
            self.rs = ctx.vector(n, np.ones(n)) if kind is not None else None
            self.ru = ctx.vector(n) if qw is not None else None
            self.rw = ctx.vector(m, np.asarray(qw, dtype=np.float64)) if qw is not None else None
            Jct = A.view(self.rs, self.ru, self.rw)
        elif Jct is None:
            Jct = DeviceMatrix(ctx, n, m + (1 if has_ball else 0), placed=True)
        Jsp = A.clone() if sparse else None
        super().__init__(Jct, m, b, has_ball=has_ball, R2=R2, n_x=(n if n_x is None else n_x), slack_row=slack_row, Jsp=Jsp)
        self.A = None if sparse else A
        self.Asp = A if sparse else None
        if kind is None or isinstance(kind, DeviceVector):
            self.kind = kind
        else:
            self.kind = ctx.vector(n, np.asarray(kind, dtype=np.float64))
        self.qw = None if qw is None else np.ascontiguousarray(qw, dtype=np.float64)
        assert self.qw is None or self.qw.size == m
        self.work = ctx.vector(n) if sparse else None
        self._ew = _capi.Elementwise(self.A.h if self.A is not None else None, self.Asp.h if self.Asp is not None else None,
                                     self.kind.h if self.kind is not None else None,
                                     self.qw.ctypes.data if self.qw is not None else None, self.work.h if self.work is not None else None)

    def _ew_ptr(self):
        return C.pointer(self._ew)


def sin_system_constraints(ctx, n: int, m: int):
    """generate_sin_system(n, m) of the reference's tests (test/test_retractions.jl:34-54) as a device-resident class:
    c_i = x[2i+1] - sin(x[2i]) (0-based): two nonzeros per constraint, kind = sin on the even variables below 2m."""
    from .device import SparseMatrix
    i = np.arange(m)
    A = SparseMatrix(ctx, n, m, np.concatenate([2 * i + 1, 2 * i]), np.concatenate([i, i]), np.concatenate([np.ones(m), -np.ones(m)]))
    kind = np.zeros(n)
    kind[0:2 * m:2] = KIND_SIN
    return ElementwiseConstraints(ctx, A, np.zeros(m), kind=kind)


def sphere_system_constraints(ctx, centers: np.ndarray, Rs: np.ndarray):
    """generate_sphere_system of the reference's tests (test/test_retractions.jl:1-31): c_i = |x - center_i|^2 - R_i^2
    = x'x - 2 center_i'x + |center_i|^2 - R_i^2  ->  A = -2 centers (n x m, dense), phi = identity, qw = 1."""
    n, m = centers.shape
    A = ctx.matrix(n, m, np.asfortranarray(-2.0 * centers))
    b = Rs ** 2 - np.einsum("ij,ij->j", centers, centers)
    return ElementwiseConstraints(ctx, A, b, qw=np.ones(m))


class NRWork:
    """NRWork(m) (src/retractions.jl:1-8): the m x m state lives inside lfpsqp_retract_nr."""

    def __init__(self, m: int):
        self.m = m


@dataclass
class NR:  # src/retractions.jl:10-19
    U: object                 # DeviceBasis or InequalityDecompProject
    Sigma: np.ndarray
    Vt: np.ndarray
    tol: float
    maxiter: int
    work: NRWork
    ineq: bool
    idata: InequalityData | None


class Euclidean:  # :51-52
    pass


@dataclass
class YRetract:  # :54-56
    idata: InequalityData


def retract_(cval: np.ndarray, xnew: DeviceVector, c_, xtilde: DeviceVector, x: DeviceVector, method):
    """retract!(cval, xnew, c!, xtilde, x, method) -> (flag, iter1, iter2)."""
    ctx = x.ctx
    if isinstance(method, Euclidean):                     # :61-65
        xnew.copy_from(xtilde)
        return 0, 0, 0
    if isinstance(method, YRetract):                      # :67-72
        xnew.copy_from(xtilde)
        y_retract_(xnew, x, method.idata)
        return 0, 0, 0
    if isinstance(method, NR):
        return _retract_nr(cval, xnew, c_, xtilde, x, method)
    from .projpenalty import ProjPenalty, retract_pp
    if isinstance(method, ProjPenalty):
        return retract_pp(cval, xnew, c_, xtilde, x, method)
    raise TypeError(f"no retract_ method for {type(method)}")


def retract_nr_batch_width_(c_, method) -> int:
    """How many trial points retract_nr_batch_ takes per pass for this retraction method and these constraints
    (lfpsqp_retract_nr_batch_width) in the context's batch mode: 4 (the default exact batch), 16 / 8 (the matrix-core opt-in, ctx.set_nr_batch_mode(True)),
    or 0 (cannot batch)."""
    if not isinstance(c_, DeviceConstraints) or not isinstance(method, NR):
        return 0
    bc = method.U._c()
    if not bc.A or not bc.W:
        return 0
    ctx = c_.Jct.ctx
    cons = c_._c()
    w = C.c_int(0)
    ctx.check(ctx.L.lfpsqp_retract_nr_batch_width(ctx.h, C.byref(bc), C.byref(cons), C.byref(w)))
    return int(w.value)


def retract_nr_batch_(cvals: np.ndarray, xnews, c_, xtildes, x, method: NR):
    """Several Newton retractions of one linesearch at once (lfpsqp_retract_nr_batch): cvals is (nb, m), xnews / xtildes lists
    of nb device vectors.  Returns a list of (flag, iter1, 0) per trial, or None when the device cannot batch this
    configuration (the caller then retracts one by one)."""
    if not isinstance(c_, DeviceConstraints) or not isinstance(method, NR):
        return None
    ctx = x.ctx
    nb = len(xnews)
    m = len(method.Sigma)
    bc = method.U._c()
    if not bc.A or not bc.W:
        return None
    Sig = np.ascontiguousarray(method.Sigma, dtype=np.float64)
    Vt = np.asfortranarray(method.Vt, dtype=np.float64)
    idc = method.idata._c() if method.ineq else None
    cons = c_._c()
    xt = (_capi.P * nb)(*[v.h for v in xtildes])
    xn = (_capi.P * nb)(*[v.h for v in xnews])
    flags = (C.c_int * nb)()
    iters = (_capi.c_i64 * nb)()
    out = np.zeros((nb, m))
    rc = ctx.L.lfpsqp_retract_nr_batch(ctx.h, C.byref(bc), Sig.ctypes.data, Vt.ctypes.data, m, C.byref(cons),
                                       C.byref(idc) if idc is not None else None, nb, xt, x.h, xn, float(method.tol),
                                       int(method.maxiter), out.ctypes.data_as(_capi.PD), flags, iters)
    if rc == -5:                                           # LFPSQP_ERR_UNSUPPORTED
        return None
    ctx.check(rc)
    cvals[:nb, :] = out
    return [(int(flags[b]), int(iters[b]), 0) for b in range(nb)]


def _download_borrowed(ctx, handle, count):
    """The first `count` entries of a device vector the LIBRARY owns (the x handed to a user callback), through the raw
    handle: wrapping it in a DeviceVector would free it on any exception path."""
    out = np.empty(count)
    ctx.check(ctx.L.lfpsqp_vec_download(ctx.h, C.c_void_p(handle), 0, out.ctypes.data_as(_capi.PD), count))
    return out


def _retract_nr(cval, xnew, c_, xtilde, x, method: NR):
    ctx = x.ctx
    m = len(method.Sigma)
    U = method.U
    bc = U._c()
    flag = C.c_int()
    iters = _capi.c_i64()
    Sig = np.ascontiguousarray(method.Sigma, dtype=np.float64)
    Vt = np.asfortranarray(method.Vt, dtype=np.float64)
    idc = method.idata._c() if method.ineq else None
    keep = None
    if isinstance(c_, DeviceConstraints):
        cons = c_._c()
        cons_p, cfun, keep = C.byref(cons), _capi.CFUN(), (cons, c_)
    else:
        nrows = method.U.idecomp.N if method.ineq else x.n

        def tramp(user, xvec_handle, cval_ptr):
            try:
                xh = _download_borrowed(ctx, xvec_handle, nrows)      # the library owns this vector: no owning wrapper
                out = np.ctypeslib.as_array(cval_ptr, shape=(m,))
                c_(out, xh)
                return 0
            except Exception as e:  # never unwind through C
                print("c! callback failed:", repr(e))
                return 1
        cfun = _capi.CFUN(tramp)
        cons_p, keep = None, (cfun, tramp)
    ctx.check(ctx.L.lfpsqp_retract_nr(ctx.h, C.byref(bc), Sig.ctypes.data, Vt.ctypes.data, m, cons_p, cfun, None,
                                      C.byref(idc) if idc is not None else None, xtilde.h, x.h, xnew.h, float(method.tol),
                                      int(method.maxiter), cval.ctypes.data_as(_capi.PD), C.byref(flag), C.byref(iters)))
    del keep
    return flag.value, iters.value, 0
