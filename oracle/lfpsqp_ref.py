"""numpy restatement of the LFPSQP.jl hot path (TEST INFRASTRUCTURE -- see
oracle/__init__.py).  Every function cites the reference file:line it follows
(paths relative to the reference checkout, e.g. src/projcg.jl:40-121).

Conventions
  * Julia ``f!`` -> Python ``f_`` (mutating, same argument order).
  * Arrays are float64 numpy; matrices are column-major (Fortran order) where
    the reference hands them to BLAS/LAPACK.
  * Operators follow the reference's duck-typed ``mul!`` protocol
    (src/projcg.jl:55-60): ``mul_(dest, Op, v)`` / ``mul_(dest, Op, v, a, b)``
    and ``adj(Op)``; plain ndarrays and the four inequality structs are
    supported, plus ``LinearMap`` (the LinearMaps.jl closure wrapper used at
    src/optimize.jl:228-230).
  * Quirks of the reference are reproduced on purpose (stale ``cval`` in the
    ProjPenalty backtracking, flag handling, pcg flag at exactly maxiter ...);
    each is marked BUG-COMPAT.
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field
from typing import Callable, Optional

import os

import numpy as np
from scipy.linalg import lapack as _lapack

_POISON = os.environ.get("ORACLE_POISON_UNINIT", "") not in ("", "0")

# --------------------------------------------------------------------------
# enums / params / termination info               (src/LFPSQP.jl:27-81)
# --------------------------------------------------------------------------


class DisplayOption(enum.Enum):
    off = 0
    iter = 1


class LinesearchOption(enum.Enum):
    armijo = 0
    exact = 1


class TerminationCondition(enum.Enum):
    f_tol = 0
    x_tol = 1
    kkt_tol = 2
    max_iter = 3
    armijo_error = 4


@dataclass
class TerminationInfo:  # src/LFPSQP.jl:45-54
    condition: TerminationCondition
    f_diff: float
    step_diff: float
    kkt_diff: float
    iter: int

    def __str__(self):
        return (f"TerminationInfo:\ncondition = {self.condition.name}\n"
                f"       Δf = {self.f_diff!r}\n   ||Δx|| = {self.step_diff!r}\n"
                f"||P(∇f)|| = {self.kkt_diff!r}\n    iters = {self.iter}")


@dataclass
class LFPSQPParams:  # src/LFPSQP.jl:57-81 (names transliterated: ϵ->eps, α->alpha ...)
    alpha: float = 1.0
    beta: float = 0.0
    t_beta: int = 0
    s: float = 0.5
    sigma: float = 1e-4
    eps_c: float = 1e-6
    eps_f: float = 1e-6
    eps_x: float = 0.0
    eps_kkt: float = 1e-6
    eps_rank: float = 1e-10
    maxiter: int = 10000
    maxiter_retract: int = 100
    maxiter_pcg: int = 100
    mu0: float = 1e-2
    disable_linesearch: bool = False
    do_project_retract: bool = True
    disp: DisplayOption = DisplayOption.iter
    callback: Optional[Callable] = None
    callback_period: int = 100
    linesearch: LinesearchOption = LinesearchOption.armijo
    do_newton: bool = True
    tn_maxiter: int = 10000
    tn_kappa: float = 0.5


# --------------------------------------------------------------------------
# operator protocol
# --------------------------------------------------------------------------


class LinearMap:
    """LinearMaps.LinearMap{Float64}(f!, n; ismutating=true, issymmetric=true)
    as used at src/optimize.jl:228,230: 3-arg mul! calls f!(dest, src); the
    5-arg form is dest = a*A*src + b*dest."""

    def __init__(self, f_, n):
        self.f_ = f_
        self.n = n
        self._tmp = _uninit(n)

    def mul_(self, dest, v, a=None, b=None):
        if a is None:
            self.f_(dest, v)
        else:
            self.f_(self._tmp, v)
            dest[:] = a * self._tmp + _scaled(b, dest)
        return dest

    def adjoint(self):
        return self


class _DenseAdj:
    def __init__(self, M):
        self.M = M


def adj(Op):
    """Op' for every operator family the reference uses."""
    if isinstance(Op, np.ndarray):
        return Op.T
    return Op.adjoint()


def _uninit(shape, order='C'):
    """Work arrays the reference allocates uninitialised.  ORACLE_POISON_UNINIT=1 fills them with NaN, so that a read before the first
    write shows up in any test (a long-lived test process recycles memory: what np.empty returns there is garbage, not zeros)."""
    a = np.empty(shape, order=order)
    if _POISON:
        a.fill(np.nan)
    return a


def _scaled(beta, y):
    """beta * y as BLAS gemv and Julia's mul!(y, A, x, alpha, beta) understand it: with beta == 0 the destination is NOT read (its contents
    may be uninitialised memory -- the reference's work arrays are `similar(...)` / `Vector{Float64}(undef, n)`, here np.empty -- and
    0 * NaN would turn garbage into NaN)."""
    return 0.0 if beta == 0 else beta * y


def mul_(dest, Op, v, a=None, b=None):
    """LinearAlgebra.mul!(dest, Op, v[, a, b])."""
    if isinstance(Op, np.ndarray):
        if Op.shape[1] == 0:  # n x 0 operator: A*v is the zero vector
            if a is None:
                dest[:] = 0.0
            else:
                dest[:] = _scaled(b, dest)
            return dest
        if a is None:
            dest[:] = Op @ v
        else:
            dest[:] = a * (Op @ v) + _scaled(b, dest)
        return dest
    return Op.mul_(dest, v, a, b)


# --------------------------------------------------------------------------
# dense LA shim                                       (src/la_helper.jl:8-44)
# --------------------------------------------------------------------------


def ksvd_(A, U, S, VT):
    """src/la_helper.jl:8-34: LAPACK dgesvd('S','S'), thin SVD, A destroyed.
    (The lwork=-1 workspace query at :19-21,:31-33 has no numerical effect.)"""
    if A.shape[1] == 0:
        return
    if A.size >= _LP64_LIMIT:
        _ksvd_by_row_blocks(A, U, S, VT)
    else:
        u, s, vt, info = _lapack.dgesvd(np.asfortranarray(A), compute_uv=1, full_matrices=0)
        U[:, :] = u
        S[:] = s
        VT[:, :] = vt
    A[:, :] = np.nan  # "destroyed" -- poison it so accidental reuse shows up


# scipy's LAPACK is LP64: dgesvd indexes the array and its workspace with 32-bit integers, and a 2e7 x 129 matrix (config 4 at the full
# size: the doubled variables of the bound formulation) has 2.58e9 entries -- the call dies with a segmentation fault.  Julia's OpenBLAS is
# ILP64 (src/la_helper.jl:22 passes BlasInt = Int64) and has no such limit.  For such a matrix the thin SVD is taken by the standard
# backward-stable tall-skinny route instead: Householder QR of row blocks that LAPACK can index (dgeqrf / dorgqr through numpy), one
# dgesvd of the stacked triangular factors, U = blockdiag(Q_k) (Qs Us).  Same singular values and right singular vectors as dgesvd's to
# rounding, the same subspace for U (all the driver uses); independent of the device's Gram-matrix route.
_LP64_LIMIT = 2 ** 31 - 2 ** 24


def _ksvd_by_row_blocks(A, U, S, VT, limit=None):
    n, m = A.shape
    limit = _LP64_LIMIT if limit is None else limit
    nblk = int(np.ceil(A.size / float(limit))) + 1
    edges = np.linspace(0, n, nblk + 1).astype(np.int64)
    Rs = []
    for k in range(nblk):
        q, r = np.linalg.qr(A[edges[k]:edges[k + 1], :])          # reduced: (rows x m), (m x m)
        U[edges[k]:edges[k + 1], :] = q                           # (parked in U until the small SVD is known)
        Rs.append(r)
    us, s, vt, info = _lapack.dgesvd(np.asfortranarray(np.vstack(Rs)), compute_uv=1, full_matrices=0)
    for k in range(nblk):
        U[edges[k]:edges[k + 1], :] = U[edges[k]:edges[k + 1], :] @ us[k * m:(k + 1) * m, :]
    S[:] = s
    VT[:, :] = vt


def kgemv_(tA, rank, alpha, A, x, beta, y):
    """src/la_helper.jl:36-44: dgemv on the leading ``rank`` columns of A."""
    Ar = A[:, :rank]
    if tA == 'N':
        y[:] = alpha * (Ar @ x[:rank]) + _scaled(beta, y)
    else:
        y[:rank] = alpha * (Ar.T @ x) + _scaled(beta, y[:rank])
    return y


# --------------------------------------------------------------------------
# projected CG                                        (src/projcg.jl)
# --------------------------------------------------------------------------


class ProjCGWork:  # src/projcg.jl:1-11
    def __init__(self, n, m):
        self.r, self.g, self.d, self.rp, self.gp, self.Ad = (_uninit(n) for _ in range(6))
        self.Utr = _uninit(m)


def projcg_(x, lam, A, U, b, c, tol=1e-6, maxit=None, work=None):
    """src/projcg.jl:40-121.  Returns (i, nr)."""
    n = len(b)
    m = len(c)
    if maxit is None:
        maxit = n + m
    if work is None:
        work = ProjCGWork(n, m)
    r, g, d, rp, gp, Ad = (w[:n] for w in (work.r, work.g, work.d, work.rp, work.gp, work.Ad))
    Utr = work.Utr[:m]
    Ut = adj(U)

    mul_(x, U, c)                      # :55
    r[:] = b                           # :56
    mul_(r, A, x, 1.0, -1.0)           # :57  r = A x - b
    g[:] = r                           # :58
    mul_(Utr, Ut, r)                   # :59
    mul_(g, U, Utr, -1.0, 1.0)         # :60
    r[:] = g                           # :61
    d[:] = -1.0 * g                    # :62

    i = 0
    nr = math.inf
    while i < min(maxit, n + m):       # :71
        i += 1
        mul_(Ad, A, d)                 # :74
        dAd = float(np.dot(d, Ad))     # :75
        if dAd <= 0:                   # :77-82
            x[:] = d / np.linalg.norm(d)
            lam[:] = np.nan
            return i, math.inf
        rg = float(np.dot(r, g))       # :84
        if rg <= 0:                    # :87
            break
        alpha = rg / dAd               # :91
        x += alpha * d                 # :92
        rp[:] = r + alpha * Ad         # :93
        gp[:] = rp                     # :95
        mul_(Utr, Ut, rp)              # :96
        mul_(gp, U, Utr, -1.0, 1.0)    # :97
        beta = float(np.dot(rp, gp)) / rg  # :98
        d[:] = beta * d - gp           # :99
        g[:] = gp                      # :100
        r[:] = gp                      # :101
        nr = float(np.linalg.norm(g))  # :103
        if nr < tol:                   # :107
            break
    r[:] = b                           # :115
    mul_(r, A, x, -1.0, 1.0)           # :116  r = b - A x
    mul_(lam, Ut, r)                   # :118
    return i, nr


# --------------------------------------------------------------------------
# bound / inequality helpers                     (src/inequality_helper.jl)
# --------------------------------------------------------------------------


class InequalityData:  # src/inequality_helper.jl:1-8, 39-89
    def __init__(self, xl=None, xu=None):
        if xl is None:
            xl = np.zeros(0)
            xu = np.zeros(0)
        xl = np.asarray(xl, dtype=float)
        xu = np.asarray(xu, dtype=float)
        n = len(xl)
        if len(xu) != n:
            raise ValueError("xl and xu are of different lengths")
        linf = np.isinf(xl)
        uinf = np.isinf(xu)
        q = np.zeros(n)
        r = np.zeros(n)
        s = np.zeros(n)
        t = np.zeros(n)
        isline = linf & uinf
        lower = ~linf & uinf
        upper = linf & ~uinf
        both = ~linf & ~uinf
        r[lower] = xl[lower]; s[lower] = -1.0; t[lower] = xl[lower]
        r[upper] = xu[upper]; s[upper] = 1.0; t[upper] = xu[upper]
        q[both] = 1.0
        r[both] = (xu[both] + xl[both]) / 2
        s[both] = 1.0
        t[both] = (xu[both] - xl[both]) ** 2 / 4
        self.q, self.r, self.s, self.t = q, r, s, t
        self.isline = isline
        self.isparabola = lower | upper


class InequalityDecomp:  # src/inequality_helper.jl:10-19
    def __init__(self, U, S, Vt, Dx, Dy, Sc, Jct, rank):
        self.U, self.Sigma, self.Vt = U, S, Vt
        self.Dx, self.Dy, self.S = Dx, Dy, Sc
        self.Jct = Jct
        self.rank = rank

    # full constraint-Jacobian-transpose operator, :215-251
    def mul_(self, dest, v, a=None, b=None):
        n = len(self.Dx)
        m = self.Jct.shape[1]
        Dx, Dy, S = self.Dx, self.Dy, self.S
        if a is None:
            mul_(dest[:n], self.Jct, v[n:n + m])
            dest[:n] += Dx * S * v[:n]
            dest[n:2 * n] = Dy * S * v[:n]
        else:
            mul_(dest[:n], self.Jct, v[n:n + m], a, b)
            dest[:n] += a * Dx * S * v[:n]
            dest[n:2 * n] *= b
            dest[n:2 * n] += a * Dy * S * v[:n]
        return dest

    def adjoint(self):
        return InequalityDecompAdjoint(self)


class InequalityDecompAdjoint:  # :21-23, :254-271
    def __init__(self, idecomp):
        self.idecomp = idecomp

    def mul_(self, dest, v, a=None, b=None):
        idc = self.idecomp
        n = len(idc.Dx)
        m = idc.Jct.shape[1]
        if a is None:
            dest[:n] = idc.S * idc.Dx * v[:n]
            dest[:n] += idc.S * idc.Dy * v[n:2 * n]
            mul_(dest[n:n + m], idc.Jct.T, v[:n])
        else:
            # generic 5-arg fallback (the reference reaches this only through
            # mul!(z, J', tmp, 1.0, mu) in pcg!, where J' is the non-adjoint
            # InequalityDecomp; kept for completeness)
            tmp = _uninit(n + m)
            self.mul_(tmp, v)
            dest[:n + m] = a * tmp + _scaled(b, dest[:n + m])
        return dest

    def adjoint(self):
        return self.idecomp


class InequalityDecompProject:  # :25-27, :161-194
    def __init__(self, idecomp):
        self.idecomp = idecomp

    def mul_(self, dest, v, a=None, b=None):
        idc = self.idecomp
        n = len(idc.Dx)
        rank = idc.rank
        Ur = idc.U[:, :rank]
        if a is None:
            mul_(dest, Ur, v[n:n + rank])
            dest[:n] += idc.Dx * v[:n]
            dest[n:2 * n] += idc.Dy * v[:n]
        else:
            mul_(dest, Ur, v[n:n + rank], a, b)
            dest[:n] += a * idc.Dx * v[:n]
            dest[n:2 * n] += a * idc.Dy * v[:n]
        return dest

    def adjoint(self):
        return InequalityDecompProjectAdjoint(self.idecomp)


class InequalityDecompProjectAdjoint:  # :29-31, :197-212
    def __init__(self, idecomp):
        self.idecomp = idecomp

    def mul_(self, dest, v, a=None, b=None):
        assert a is None
        idc = self.idecomp
        n = len(idc.Dx)
        rank = idc.rank
        dest[:n] = idc.Dx * v[:n]
        dest[:n] += idc.Dy * v[n:2 * n]
        mul_(dest[n:n + rank], idc.U[:, :rank].T, v)
        return dest

    def adjoint(self):
        return InequalityDecompProject(self.idecomp)


def generate_initial_y_(xaug, idata):  # :92-109
    n = len(idata.q)
    x = xaug[:n]
    y = xaug[n:2 * n]
    line, par = idata.isline, idata.isparabola
    circ = ~(line | par)
    y[line] = x[line]
    with np.errstate(divide='ignore', invalid='ignore'):
        y[par] = np.sqrt(np.maximum(-(x[par] - idata.t[par]) / idata.s[par], 0.0)) + idata.r[par]
    y[circ] = np.sqrt(np.maximum(idata.t[circ] - (x[circ] - idata.r[circ]) ** 2, 0.0)) + idata.r[circ]
    return xaug


def calculate_h_(cvalaug, x, idata):  # :112-122
    n = len(x) // 2
    xv = x[:n]
    yv = x[n:2 * n]
    cvalaug[:n] = (idata.q * (xv - idata.r) ** 2 + (1.0 - idata.q ** 2) * xv +
                   idata.s * (yv - idata.r) ** 2 - (1.0 - idata.s ** 2) * yv - idata.t)
    return cvalaug


def inequality_gradient_(idecomp, x, idata):  # :125-141
    n = len(x) // 2
    Dx, Dy, S = idecomp.Dx, idecomp.Dy, idecomp.S
    Dx[:] = 2.0 * idata.q * (x[:n] - idata.r) + (idata.q == 0.0)
    Dy[:] = 2.0 * idata.s * (x[n:2 * n] - idata.r) - (idata.s == 0.0)
    S[:] = np.sqrt(Dx * Dx + Dy * Dy)
    Dx /= S
    Dy /= S


def augmented_hess_lag_vec_(dest, src, hess_lag_vec_, x, lam_kkt, lamy_kkt, idata):  # :144-158
    n = len(x) // 2
    hess_lag_vec_(dest[:n], src[:n], x[:n], lam_kkt)
    dest[:n] += 2 * lamy_kkt * idata.q * src[:n]
    dest[n:2 * n] = 2 * lamy_kkt * idata.s * src[n:2 * n]
    return dest


def calculate_lambda_kkt_(lam_kkt, lamy_kkt, Qtgf, idecomp):  # :286-308
    n = len(idecomp.Dx)
    rank = idecomp.rank
    Sig = idecomp.Sigma
    m = len(Sig)
    Qtgf[n:n + rank] /= Sig[:rank]
    Qtgf[n + rank:n + m] = 0.0
    lam_kkt[:] = idecomp.Vt.T @ Qtgf[n:n + m]
    mul_(lamy_kkt, idecomp.Jct, lam_kkt)
    lamy_kkt *= -1.0 * idecomp.Dx / idecomp.S
    lamy_kkt += Qtgf[:n] / idecomp.S
    return lam_kkt, lamy_kkt


# --------------------------------------------------------------------------
# retractions                                        (src/retractions.jl)
# --------------------------------------------------------------------------


def y_retract_(xnewaug, xaug, idata):  # src/retractions.jl:451-500
    n = len(xaug) // 2
    xnew = xnewaug[:n]
    ynew = xnewaug[n:2 * n]
    x = xaug[:n]
    y = xaug[n:2 * n]
    line, par = idata.isline, idata.isparabola
    circ = ~(line | par)

    xnew[line] = ynew[line]                               # :463

    if par.any():                                         # :464-486
        s = idata.s[par]; r = idata.r[par]
        g1 = -s
        g2 = -2 * (y[par] - r)
        ng = np.sqrt(g1 * g1 + g2 * g2)
        ux = x[par] - xnew[par] + g1 / ng
        uy = y[par] - ynew[par] + g2 / ng
        with np.errstate(divide='ignore', invalid='ignore'):
            a = s * uy ** 2
            b = ux + 2 * s * (ynew[par] - r) * uy
            c = xnew[par] + s * (ynew[par] - r) ** 2 - r
            a1 = -b / (2 * a)
            a2 = np.sqrt(b ** 2 - 4 * a * c) / (2 * a)
            gam = np.minimum(a1 + a2, a1 - a2)
        xnew[par] += gam * ux
        ynew[par] += gam * uy

    if circ.any():                                        # :487-496
        c = idata.r[circ]
        rho = np.sqrt(idata.t[circ])
        dist = np.sqrt((xnew[circ] - c) ** 2 + (ynew[circ] - c) ** 2)
        yn = c + rho * (ynew[circ] - c) / dist
        xn = c + rho * (xnew[circ] - c) / dist
        ynew[circ] = yn
        xnew[circ] = xn
    return xnew


class NRWork:  # src/retractions.jl:1-8
    def __init__(self, m):
        self.D = _uninit((m, m), order='F')
        self.tmp_m = _uninit(m)
        self.tmp_m2 = _uninit(m)
        self.dc = _uninit(m)


@dataclass
class NR:  # :10-19
    U: np.ndarray
    Sigma: np.ndarray
    Vt: np.ndarray
    tol: float
    maxiter: int
    work: NRWork
    ineq: bool
    idata: InequalityData


class ProjPenaltyWork:  # :21-33
    def __init__(self, m, n, m_ineq, n_ineq):
        self.J = _uninit((m, n), order='F')
        self.tmp_m = _uninit(m_ineq)
        self.r, self.p, self.z, self.dx, self.g = (_uninit(n_ineq) for _ in range(5))
        # The reference allocates cvalaug undef as well (src/retractions.jl:31-32) and READS its last m entries before their first write:
        # norm(cvalaug, Inf) at :352 covers the whole vector, the tail is filled at :356 -- on the first call that is uninitialised memory
        # (later calls see the previous call's values).  Zeros here, as the device's zero-filled allocation has it: the one place where
        # the reference's behaviour is undefined and the restatement has to choose.
        self.cvalaug = np.zeros(m_ineq)


@dataclass
class ProjPenalty:  # :35-49
    jac_: Callable
    U: np.ndarray
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
    idata: InequalityData


class Euclidean:  # :51-52
    pass


@dataclass
class YRetract:  # :54-56
    idata: InequalityData


def _retract_nr(cval, xnew, c_, xtilde, x, method):
    """src/retractions.jl:75-177."""
    U, Sig, Vt = method.U, method.Sigma, method.Vt
    tol, maxiter, work = method.tol, method.maxiter, method.work
    D, tmp_m, tmp_m2, dc = work.D, work.tmp_m, work.tmp_m2, work.dc
    m = len(Sig)
    xnew[:] = xtilde
    if method.ineq:
        y_retract_(xnew, x, method.idata)
        c_(cval, xnew[:len(xnew) // 2])
    else:
        c_(cval, xnew)
    D[:, :] = Vt / Sig[:, None]                        # :126-130  D[k,j] = Vt[k,j]/Σ[k]
    i = 0
    while i < maxiter:
        if np.max(np.abs(cval), initial=0.0) < tol:    # :135
            break
        tmp_m[:] = -1.0 * (D @ cval)                   # :140
        xnew += U @ tmp_m                              # :141
        if method.ineq:                                # :144-149
            y_retract_(xnew, x, method.idata)
            c_(tmp_m2, xnew[:len(xnew) // 2])
        else:
            c_(tmp_m2, xnew)
        dc[:] = tmp_m2 - cval                          # :152
        cval[:] = tmp_m2                               # :153
        tmp_m2[:] = D.T @ tmp_m                        # :156
        tmp_m[:] = -1.0 * (D @ dc) + tmp_m             # :157
        alpha = 1 / float(np.dot(tmp_m2, dc))          # :159
        D += alpha * np.outer(tmp_m, tmp_m2)           # :160
        i += 1
    flag = 1 if i == maxiter else 0
    return flag, i, 0


def no_precondition(z, r):  # :259-263
    z[:] = r
    return z


def proj_precondition_(z, r, mu, U, Sig, rank, tmp_m):  # :248-257 (dead on the live path)
    z[:] = r
    kgemv_('T', rank, 1.0, U, r, 0.0, tmp_m)
    tmp_m[:rank] *= Sig[:rank] * Sig[:rank] / (mu + Sig[:rank] * Sig[:rank])
    kgemv_('N', rank, -1 / mu, U, tmp_m, 1 / mu, z)
    return z


# Switches for restatements of DEVICE options that have no counterpart on the reference's live path (all off = the reference).
EXTENSIONS = {"pp_precondition": False}


def exact_precondition(mu, J, idecomp=None):
    """EXTENSION (no counterpart on the reference's live path): M!(z, r) = (D0 + E E')^-1 r for the operator of ProjPenalty's inner solve,
    E = [J'; 0], D0 = mu I (plain) or mu I + [a; b][a b] per variable with a = Dx.*S, b = Dy.*S (bounds; src/inequality_helper.jl:215-271),
    by the Woodbury identity: D0^-1 - D0^-1 E (I + E' D0^-1 E)^-1 E' D0^-1.  Plain operator: the same map as proj_precondition!
    (src/retractions.jl:248-257) with U, Sigma from J itself."""
    Jct = np.asarray(J).T if idecomp is None else idecomp.Jct
    N, m = Jct.shape
    if idecomp is None:
        i11 = np.full(N, 1.0 / mu)
        i12 = i22 = None
    else:
        a, b = idecomp.Dx * idecomp.S, idecomp.Dy * idecomp.S
        det = mu * (a * a + b * b + mu)
        i11, i12, i22 = (b * b + mu) / det, -(a * b) / det, (a * a + mu) / det
    K = np.linalg.inv(np.eye(m) + Jct.T @ (i11[:, None] * Jct))

    def M_(z, r):
        if idecomp is None:
            ux = i11 * r
            z[:] = ux - i11 * (Jct @ (K @ (Jct.T @ ux)))
        else:
            rx, ry = r[:N], r[N:]
            ux, uy = i11 * rx + i12 * ry, i12 * rx + i22 * ry
            v = Jct @ (K @ (Jct.T @ ux))
            z[:N] = ux - i11 * v
            z[N:] = uy - i12 * v
        return z
    return M_


def pcg_(mu, J, M_, x, r, p, z, tmp_m, tol, maxiter):
    """src/retractions.jl:179-246.  Returns (flag, i)."""
    norm_res = math.inf
    rho = 1.0
    p[:] = 0.0
    Jt = adj(J)
    i = 0
    while norm_res > tol and i < maxiter:
        M_(z, r)                                   # :209
        rho_prev = rho
        rho = float(np.dot(z, r))                  # :213
        beta = rho / rho_prev                      # :216
        p[:] = z + beta * p                        # :217
        z[:] = p                                   # :220
        mul_(tmp_m, J, p)                          # :221
        mul_(z, Jt, tmp_m, 1.0, mu)                # :222
        alpha = rho / float(np.dot(p, z))          # :227
        x += alpha * p                             # :232
        r += -alpha * z                            # :233
        norm_res = float(np.linalg.norm(r))        # :235
        i += 1
    flag = 1 if i == maxiter else 0                # BUG-COMPAT :240-243 (set even if converged at maxiter)
    return flag, i


def _retract_pp(cval, xnew, c_, xtilde, x, method):
    """src/retractions.jl:265-441."""
    jac_ = method.jac_
    mu0, tol, maxiter, maxiter_pcg = method.mu0, method.tol, method.maxiter, method.maxiter_pcg
    w = method.work
    idecomp, idata = method.idecomp, method.idata
    J, tmp_m, r, p, z, dx, g, cvalaug = w.J, w.tmp_m, w.r, w.p, w.z, w.dx, w.g, w.cvalaug
    fulljac = adj(idecomp) if method.ineq else J          # :324
    fulljac_t = adj(fulljac)
    flag = 0
    xnew[:] = xtilde
    mu = mu0
    n = len(xnew) // 2 if method.ineq else len(xnew)
    m = len(method.Sigma)
    i = 0
    pcg_iter_count = 0
    while i < maxiter:
        jac_(J, cval, xnew[:n])                              # :340
        curtol = np.max(np.abs(cval), initial=0.0)
        if method.ineq:                                      # :343-353
            inequality_gradient_(idecomp, xnew, idata)
            idecomp.Jct[:, :] = J.T
            calculate_h_(cvalaug, xnew, idata)
            curtol = max(curtol, np.max(np.abs(cvalaug), initial=0.0))
        cvalaug[len(cvalaug) - m:] = cval                    # :356
        if curtol < tol:                                     # :359
            break
        g[:] = xnew - xtilde                                 # :364
        prev_obj_val = float(np.dot(cvalaug, cvalaug)) + mu * float(np.dot(g, g))  # :366
        mul_(g, fulljac_t, cvalaug, 1.0, mu)                 # :369
        dx[:] = 0.0
        r[:] = g
        M_ = no_precondition
        if getattr(method, "precondition", False):
            # NOT on the reference's live path (its M! is no_precondition, :375): the exact preconditioner the device offers as an option
            # (DeviceOptions.pp_precondition, lfpsqp_pcg_pre), restated here so that the option has a checker.  For the plain operator it is
            # proj_precondition! (:248-257, the call commented out at :374) with the factors of the CURRENT J.
            M_ = exact_precondition(mu, J, idecomp if method.ineq else None)
        pcg_flag, pcg_i = pcg_(mu, fulljac, M_, dx, r, p, z, tmp_m, tol, maxiter_pcg)  # :375
        pcg_iter_count += pcg_i
        if pcg_flag > 0:                                     # :377-381
            flag = 2
            break
        p[:] = xnew                                          # :384
        ar_dot = -float(np.dot(g, dx))                       # :385
        alpha = 1.0
        xnew -= alpha * dx                                   # :389
        g[:] = xnew - xtilde
        dist2 = float(np.dot(g, g))
        c_(cval, xnew[:n])                                   # :392
        if method.ineq:
            calculate_h_(cvalaug, xnew, idata)
        cvalaug[len(cvalaug) - m:] = cval                    # :399
        armijo_count = 0
        while float(np.dot(cvalaug, cvalaug)) + mu * dist2 > prev_obj_val + 1e-4 * alpha * ar_dot:  # :403
            alpha /= 2
            xnew[:] = p - alpha * dx
            g[:] = xnew - xtilde
            dist2 = float(np.dot(g, g))
            c_(cvalaug, xnew[:n])                            # :410 writes cvalaug[0:m]
            if method.ineq:
                calculate_h_(cvalaug, xnew, idata)
            cvalaug[len(cvalaug) - m:] = cval                # BUG-COMPAT :417 stale full-step cval
            armijo_count += 1
            if armijo_count == 100:                          # :422-425 (only leaves the inner loop)
                flag = 3
                break
        i += 1
        mu = min(mu * 0.1, float(np.linalg.norm(cvalaug)))   # :431
    if i == maxiter:                                         # :435-437
        flag = 1
    return flag, i, pcg_iter_count


def retract_(cval, xnew, c_, xtilde, x, method):
    """retract!(cval, xnew, c!, xtilde, x, method) -- dispatch on the method
    type as Julia does (src/retractions.jl:61,67,75,265)."""
    if isinstance(method, Euclidean):
        xnew[:] = xtilde
        return 0, 0, 0
    if isinstance(method, YRetract):
        xnew[:] = xtilde
        y_retract_(xnew, x, method.idata)
        return 0, 0, 0
    if isinstance(method, NR):
        return _retract_nr(cval, xnew, c_, xtilde, x, method)
    if isinstance(method, ProjPenalty):
        return _retract_pp(cval, xnew, c_, xtilde, x, method)
    raise TypeError(f"no retract_ method for {type(method)}")


# --------------------------------------------------------------------------
# linesearches                                        (src/linesearch.jl)
# --------------------------------------------------------------------------


class ArmijoWork:  # :1-5
    def __init__(self, n):
        self.xtilde = _uninit(n)


class ExactLinesearchWork:  # :7-14
    def __init__(self, n):
        self.tmp_n1, self.tmp_n2, self.tmp_n3, self.tmp_n4 = (_uninit(n) for _ in range(4))


def armijo_(xnew, x, n, d, g, f, fval, retract_method, cval, c_, param, work):
    """src/linesearch.jl:32-89."""
    f_diff = math.inf
    step_diff = math.inf
    alpha = param.alpha
    flag = 0
    tot_iter1 = 0
    tot_iter2 = 0
    newf = 0.0
    ar_dot = float(np.dot(d, g))
    xtilde = work.xtilde
    step = xtilde
    while step_diff > param.eps_x:
        xtilde[:] = x + alpha * d
        flag, iter1, iter2 = retract_(cval, xnew, c_, xtilde, x, retract_method)
        tot_iter1 += iter1
        tot_iter2 += iter2
        if flag > 0:                                   # :57-60
            alpha *= param.s
            continue
        step[:] = xnew - x
        newf = f(xnew)
        step_diff = float(np.linalg.norm(step[:n]))
        f_diff = abs(newf - fval)
        if param.disable_linesearch:
            break
        if (newf - fval) <= param.sigma * alpha * ar_dot:   # :75
            break
        alpha *= param.s
        if alpha < 1e-100:                             # :82
            flag = 99
            break
    return flag, tot_iter1, tot_iter2, newf, f_diff, step_diff, alpha


def exact_linesearch_(xnew, x, n, d, f, fval, retract_method, cval, c_, param, work):
    """src/linesearch.jl:107-339 (golden-section search with bracket growing /
    shrinking; the four work vectors rotate roles exactly as in the reference)."""
    phi1 = (3 - math.sqrt(5)) / 2
    phi2 = (math.sqrt(5) - 1) / 2
    phi3 = (math.sqrt(5) + 1) / 2
    Delta = param.alpha
    flag = 0
    tot_iter1 = tot_iter2 = 0
    newf = 0.0
    f_a = f_b = f_c = f_d = 0.0
    a_a = a_b = a_c = a_d = 0.0
    x_a, x_b, x_c, x_d = work.tmp_n1, work.tmp_n2, work.tmp_n3, work.tmp_n4
    step = work.tmp_n1
    do_shrinking = True

    def _retract(pt):
        nonlocal tot_iter1, tot_iter2
        fl, i1, i2 = retract_(cval, xnew, c_, pt, x, retract_method)
        tot_iter1 += i1
        tot_iter2 += i2
        pt[:] = xnew
        return fl

    x_d[:] = x
    f_d = fval
    while True:                                         # growing :145-183
        x_b, x_c, x_d = x_c, x_d, x_b
        f_b, f_c = f_c, f_d
        a_b, a_c = a_c, a_d
        x_d[:] = x + (a_d + Delta) * d
        flag = _retract(x_d)
        a_d += Delta
        if flag > 0 or a_d > 1.0:
            f_d = math.inf
            break
        f_d = f(x_d)
        if f_d > f_c:
            break
        do_shrinking = False
        Delta *= phi3

    if do_shrinking:                                    # :186-233
        f_b = fval
        a_b = 0.0
        x_b[:] = x
        f_c = math.inf
        a_c = Delta
        x_d, x_c = x_c, x_d
        while True:
            x_d, x_c = x_c, x_d
            f_d = f_c
            a_d = a_c
            x_c[:] = x + (phi1 * a_c) * d
            flag = _retract(x_c)
            a_c *= phi1
            if flag > 0 or a_c > 1.0:
                f_c = math.inf
            else:
                f_c = f(x_c)
            if f_c <= fval or a_c < 1e-100:
                break

    f_a, f_b = f_b, f_c                                  # :236-245
    a_a, a_b = a_b, a_c
    x_a, x_b, x_c = x_b, x_c, x_a
    a_c = a_a + phi2 * (a_d - a_a)
    x_c[:] = x + a_c * d
    flag = _retract(x_c)
    if flag > 0 or a_c > 1.0:
        f_c = math.inf
    else:
        f_c = f(x_c)

    nd = float(np.linalg.norm(d))
    while (a_c - a_b) > 1e-6 * nd:                       # :267-321
        if f_b < f_c or math.isinf(f_c):
            x_d, x_c, x_b = x_c, x_b, x_d
            f_d, f_c = f_c, f_b
            a_d, a_c = a_c, a_b
            a_b = a_a + phi1 * (a_d - a_a)
            x_b[:] = x + a_b * d
            flag = _retract(x_b)
            f_b = f(x_b)
        else:
            x_a, x_b, x_c = x_b, x_c, x_a
            f_a, f_b = f_b, f_c
            a_a, a_b = a_b, a_c
            a_c = a_a + phi2 * (a_d - a_a)
            x_c[:] = x + a_c * d
            flag = _retract(x_c)
            if flag > 0 or a_c > 1.0:
                f_c = math.inf
            else:
                f_c = f(x_c)

    if f_b < f_c:                                        # :324-332
        xnew[:] = x_b
        newf = f_b
        alpha = a_b
    else:
        xnew[:] = x_c
        newf = f_c
        alpha = a_c
    step[:] = xnew - x
    step_diff = float(np.linalg.norm(step[:n]))
    f_diff = abs(newf - fval)
    return flag, tot_iter1, tot_iter2, newf, f_diff, step_diff, alpha


# --------------------------------------------------------------------------
# outer driver                                        (src/optimize.jl)
# --------------------------------------------------------------------------


def _print_iter_header():  # src/optimize.jl:445-448
    print("   step |          f     ||c||      |Δf|    ||Δx||  |   S iter      res  |   M   iter  (pcg)  |        α  flag")
    print("-" * 110)


def _print_first_line(fval, normc):  # :450-452
    print("      0 | %10.3e  %8.1e                      |                    |                    |               " % (fval, normc))


def _print_iter(i, fval, normc, fstep, normx, steptype, tn_iter, tn_res, methodtype, iter1, iter2, alpha, flag):  # :454-472
    method = "NR" if methodtype == 0 else "PP"
    stepname = "GD" if steptype == 0 else "TN"
    print("%7d | %10.3e  %8.1e  %8.1e  %8.1e  |  %s %4d %8.1e  |  %s %6d %6d  | %8.1e  %4d" %
          (i, fval, normc, fstep, normx, stepname, tn_iter, tn_res, method, iter1, iter2, alpha, flag))


def _amax(v):
    return float(np.max(np.abs(v), initial=0.0))


def optimize_core(f, grad_, c_, jac_, hess_lag_vec_, x0, xl, xu, m, param, trace=None):
    """src/optimize.jl:119-443, the explicit-derivative method all others funnel
    into.  ``trace`` (optional list) receives a dict per outer iteration with
    copies of the iterate and the scalars the parity protocol compares
     -- not part of the reference signature."""
    x0 = np.asarray(x0, dtype=float)
    n = len(x0)
    if xl is not None and xu is not None:
        if not (len(xl) == len(xu) == len(x0)):
            raise ValueError("xl, xu, and x0 must all be the same length")
    if (xl is None and xu is None) or (np.all(np.asarray(xl) == -np.inf) and np.all(np.asarray(xu) == np.inf)):
        ineq = False
        f_aug = f
        ineqdata = InequalityData()
    else:
        ineq = True
        xl = np.asarray(xl, dtype=float)
        xu = np.asarray(xu, dtype=float)
        if np.any(xl > xu):
            raise ValueError("Infeasible: lower bounds cannot be greater than upper bounds")
        f_aug = lambda xx: f(xx[:n])
        PJct = _uninit((2 * n, m), order='F')
        lamy_kkt = np.zeros(n)
        ineqdata = InequalityData(xl, xu)

    n_ineq = 2 * n if ineq else n
    m_ineq = m + n if ineq else m

    x = _uninit(n_ineq)
    x[:n] = x0
    if ineq:
        generate_initial_y_(x, ineqdata)
    obj_values = []

    xnew = _uninit(n_ineq)
    Jc = _uninit((m, n), order='F')
    Jct = _uninit((n, m), order='F')
    g = np.zeros(n_ineq)
    d = _uninit(n_ineq)
    tmp_n = _uninit(n_ineq)
    tmp_m = _uninit(m_ineq)
    cval = np.zeros(m)
    lam_kkt = np.zeros(m)
    term_cond = TerminationCondition.f_tol

    U = _uninit((n_ineq, m), order='F')
    Sig = _uninit(m)
    Vt = _uninit((m, m), order='F')

    newton_d = _uninit(n_ineq)
    newton_dlam = _uninit(m_ineq)
    newton_b2 = np.zeros(m_ineq)
    projcgwork = ProjCGWork(n_ineq, m_ineq)
    prev_grad_norm = 0.0
    grad_norm = math.inf

    if ineq:
        ineqdecomp = InequalityDecomp(U, Sig, Vt, _uninit(n), _uninit(n), _uninit(n), Jct, m)
    else:
        ineqdecomp = InequalityDecomp(U, Sig, Vt, np.zeros(0), np.zeros(0), np.zeros(0), Jct, m)
    ineqproject = InequalityDecompProject(ineqdecomp)

    if ineq:                                               # :227-231
        newton_map = LinearMap(lambda dest, src: augmented_hess_lag_vec_(
            dest, src, hess_lag_vec_, x, lam_kkt, lamy_kkt, ineqdata), 2 * n)
    else:
        newton_map = LinearMap(lambda dest, src: hess_lag_vec_(dest, src, x, lam_kkt), n)

    nr = NR(U, Sig, Vt, param.eps_c, param.maxiter_retract, NRWork(m), ineq, ineqdata)
    pp = ProjPenalty(jac_, U, Sig, Vt, m, param.mu0, param.eps_c, param.maxiter_retract, param.maxiter_pcg,
                     ProjPenaltyWork(m, n, m_ineq, n_ineq), ineq, ineqdecomp, ineqdata)
    pp.precondition = bool(EXTENSIONS["pp_precondition"])      # (False: the reference's live path)
    euc = Euclidean()
    yr = YRetract(ineqdata)
    armijo_work = ArmijoWork(n_ineq)
    exact_work = ExactLinesearchWork(n_ineq)

    i = 0
    f_diff = math.inf
    step_diff = math.inf
    kkt_diff = math.inf

    fval = f_aug(x)
    obj_values.append(fval)
    if m > 0:
        c_(cval, x[:n])
    disp = param.disp == DisplayOption.iter
    if disp:
        _print_iter_header()
        _print_first_line(fval, _amax(cval))

    while True:
        grad_(g[:n], x[:n])                                # :259
        d[:] = -1.0 * g                                    # :262
        if param.beta > 0:                                 # :264-273
            tmp_n[:] = np.random.standard_normal(n_ineq)
            if param.t_beta > 0:
                d += param.beta * max(1 - i / param.t_beta, 0.0) * tmp_n
            else:
                d += param.beta * tmp_n
        if ineq:
            inequality_gradient_(ineqdecomp, x, ineqdata)  # :277
        rank = m
        if m > 0:
            jac_(Jc, cval, x[:n])                          # :283
            Jct[:, :] = Jc.T                               # :284
            if ineq:
                PJct[:n, :] = (1.0 - ineqdecomp.Dx * ineqdecomp.Dx)[:, None] * Jct   # :288
                PJct[n:, :] = (-1.0 * ineqdecomp.Dy * ineqdecomp.Dx)[:, None] * Jct  # :289
                ksvd_(PJct, U, Sig, Vt)
            else:
                # the reference destroys Jct here (:293); keep a copy semantics-free:
                ksvd_(Jct.copy(order='F'), U, Sig, Vt)
            for j, a in enumerate(Sig):                    # :297-302
                if a < param.eps_rank:
                    rank = j
                    break
            if not ineq:                                   # :305-308
                kgemv_('T', rank, 1.0, U, d, 0.0, tmp_m)
                kgemv_('N', rank, -1.0, U, tmp_m, 1.0, d)
        if ineq:                                           # :312-318
            ineqdecomp.rank = rank
            mul_(tmp_m, adj(ineqproject), d)
            mul_(d, ineqproject, tmp_m, -1.0, 1.0)
        kkt_diff = _amax(d)                                # :320
        pp.rank = rank
        steptype = 0
        tn_iter = 0
        tn_res = 0.0
        if ineq:                                           # :331-343
            calculate_lambda_kkt_(lam_kkt, lamy_kkt, tmp_m, ineqdecomp)
        elif m > 0:
            tmp_m[:rank] /= Sig[:rank]
            tmp_m[rank:m] = 0.0
            lam_kkt[:] = Vt.T @ tmp_m

        if trace is not None:
            trace.append(dict(iter=i, x=x.copy(), fval=fval, kkt_diff=kkt_diff, rank=rank,
                              lam_kkt=lam_kkt.copy(), cval=cval.copy()))

        if f_diff <= param.eps_f:                          # :347-359
            term_cond = TerminationCondition.f_tol
            break
        elif step_diff <= param.eps_x:
            term_cond = TerminationCondition.x_tol
            break
        elif i >= param.maxiter:
            term_cond = TerminationCondition.max_iter
            break
        elif kkt_diff <= param.eps_kkt:
            term_cond = TerminationCondition.kkt_tol
            break

        if param.do_newton:                                # :364-390
            if ineq:
                Qview = ineqproject
                b2 = newton_b2[:n + rank]
            else:
                Qview = U[:, :rank]
                b2 = newton_b2[:rank]
            grad_norm = float(np.linalg.norm(d))
            with np.errstate(divide='ignore', invalid='ignore'):
                ratio = np.float64(grad_norm) / np.float64(prev_grad_norm)
            tol = param.tn_kappa * min(1.0, float(ratio)) * grad_norm   # julia min(1, NaN) = NaN
            if math.isnan(float(ratio)):
                tol = math.nan
            prev_grad_norm = grad_norm
            tn_iter, tn_res = projcg_(newton_d, newton_dlam, newton_map, Qview, d, b2,
                                      tol=tol, maxit=param.tn_maxiter, work=projcgwork)
            if float(np.dot(newton_d, d)) > 0.0:
                d[:] = newton_d
                steptype = 1
            if trace is not None:
                trace[-1].update(tn_iter=tn_iter, tn_res=tn_res, steptype=steptype, tn_tol=tol)

        if m > 0:                                          # :396-412
            if rank == m and not param.do_project_retract:
                retract_method, mtype = nr, 0
            else:
                retract_method, mtype = pp, 1
        else:
            retract_method, mtype = (yr, 0) if ineq else (euc, 0)

        if param.linesearch == LinesearchOption.armijo or param.disable_linesearch:   # :415-420
            flag, iter1, iter2, newf, f_diff, step_diff, alpha = armijo_(
                xnew, x, n, d, g, f_aug, fval, retract_method, cval, c_, param, armijo_work)
        else:
            flag, iter1, iter2, newf, f_diff, step_diff, alpha = exact_linesearch_(
                xnew, x, n, d, f_aug, fval, retract_method, cval, c_, param, exact_work)

        x[:] = xnew                                        # :424-427
        fval = newf
        obj_values.append(fval)
        if disp:
            _print_iter(i + 1, fval, _amax(cval), f_diff, step_diff, steptype, tn_iter, tn_res,
                        mtype, iter1, iter2, alpha, flag)
        if trace is not None:
            trace[-1].update(mtype=mtype, retract_iter1=iter1, retract_iter2=iter2, alpha=alpha, ls_flag=flag)
        i += 1
        if param.callback is not None and i % param.callback_period == 0:   # :432-434
            param.callback(i, x)

    if i == param.maxiter and disp:
        print("Warning: Maximum # of outer iterations reached")
    return x[:n].copy(), np.array(obj_values), lam_kkt, TerminationInfo(term_cond, f_diff, step_diff, kkt_diff, i)


@dataclass
class Derivatives:
    """Analytic derivatives in the USER's variables.  The reference obtains
    these by AD (src/autodiff_generators.jl), which is out of scope;
    callers of the convenience methods supply them instead.

      grad_(g, x)                    gradient of f
      jac_c_(J, cval, x)             m x n Jacobian of c and c(x)      (None if m == 0)
      jac_d_(J, dval, x)             p x n Jacobian of d and d(x)      (None if p == 0)
      hess_lag_vec_(dest, src, x, lam)   (∇²f + Σ lam_i ∇²[c;d]_i) src, lam of length m+p,
                                     contract of autodiff_generators.jl:80-104
    """
    grad_: Callable
    hess_lag_vec_: Callable
    jac_c_: Optional[Callable] = None
    jac_d_: Optional[Callable] = None


def optimize(*args, derivatives: Optional[Derivatives] = None, trace=None):
    """The reference's six ``optimize`` methods (src/optimize.jl:13,83,88,107,112,119),
    selected by positional-argument count exactly as Julia's dispatch resolves
    them (a trailing LFPSQPParams is optional everywhere except the 10-argument
    explicit-derivative form, as in the reference)."""
    args = list(args)
    param = LFPSQPParams()
    if args and isinstance(args[-1], LFPSQPParams):
        param = args.pop()
    k = len(args)
    if k == 9:      # (f, grad!, c!, jac!, hess_lag_vec!, x0, xl, xu, m)            :119
        f, grad_, c_, jac_, hlv_, x0, xl, xu, m = args
        return optimize_core(f, grad_, c_, jac_, hlv_, x0, xl, xu, m, param, trace)
    if derivatives is None:
        raise NotImplementedError("the AD generators (src/autodiff_generators.jl) are out of scope; "
                                  "pass derivatives=Derivatives(...)")
    dv = derivatives
    if k == 2:      # (f, x0)                                                        :112
        f, x0 = args
        return optimize_core(f, dv.grad_, None, None, dv.hess_lag_vec_, x0, None, None, 0, param, trace)
    if k == 4:      # (f, c!, x0, m)                                                 :107
        f, c_, x0, m = args
        return optimize_core(f, dv.grad_, c_, dv.jac_c_, dv.hess_lag_vec_, x0, None, None, m, param, trace)
    if k == 6:      # (f, c!, x0, xl, xu, m)                                         :88
        f, c_, x0, xl, xu, m = args
        return optimize_core(f, dv.grad_, c_, dv.jac_c_ if m > 0 else None, dv.hess_lag_vec_, x0, xl, xu, m, param, trace)
    if k == 8:      # (f, c!, d!, x0, xl, xu, m, p)   d <= 0                         :83
        f, c_, d_, x0, xl, xu, m, p = args
        return _optimize_slack(f, c_, d_, -np.inf * np.ones(p), np.zeros(p), x0, xl, xu, m, p, param, dv, trace)
    if k == 10:     # (f, c!, d!, dl, du, x0, xl, xu, m, p)                          :13
        f, c_, d_, dl, du, x0, xl, xu, m, p = args
        return _optimize_slack(f, c_, d_, dl, du, x0, xl, xu, m, p, param, dv, trace)
    raise TypeError(f"no optimize method with {k} positional arguments")


def _optimize_slack(f, c_, d_, dl, du, x0, xl, xu, m, p, param, dv, trace):
    """src/optimize.jl:13-71: slack variables turn dl <= d(x) <= du into
    equalities d(x) - s = 0 with bounds on s; n -> n+p, m -> m+p."""
    if d_ is None or p == 0:
        return optimize(f, c_, x0, xl, xu, m, param, derivatives=dv, trace=trace)
    if not (len(dl) == len(du) == p):
        raise ValueError("Bound vectors dl and du must be of size p")
    x0 = np.asarray(x0, dtype=float)
    n = len(x0)
    if xl is None:
        xl = -np.inf * np.ones(n)
    if xu is None:
        xu = np.inf * np.ones(n)
    x0_aux = _uninit(n + p)
    x0_aux[:n] = x0
    d_(x0_aux[n:], x0)
    xl_aux = np.concatenate([xl, dl])
    xu_aux = np.concatenate([xu, du])

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

    x, obj_values, lam_kkt, ti = optimize_core(f_aux, grad_aux_, c_aux_, jac_aux_, hlv_aux_,
                                               x0_aux, xl_aux, xu_aux, m + p, param, trace)
    return x[:n], obj_values, lam_kkt, ti
