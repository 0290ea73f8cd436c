"""Tangent setup on the device -- the replacement of ``ksvd!`` (reference src/la_helper.jl:8-34,
called every outer iteration at src/optimize.jl:291/293)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._capi import c_i64
from .device import DeviceMatrix, DeviceVector


def gram(M: DeviceMatrix, ncols: int | None = None, w2: DeviceVector | None = None) -> np.ndarray:
    """M[:, :ncols]' diag(w2) M[:, :ncols] (replicated, all-reduced)."""
    ncols = M.m if ncols is None else ncols
    G = np.empty((ncols, ncols), order='F')
    M.ctx.check(M.ctx.L.lfpsqp_gram(M.ctx.h, M.h, ncols, w2.h if w2 is not None else None, G.ctypes.data))
    return G


def gram_rhs(M: DeviceMatrix, es, ncols: int | None = None, w2: DeviceVector | None = None):
    """(G, X): G as :func:`gram`, X[:, k] = M[:, :ncols]' (sqrt(w2) .* es[k]) for up to two device n-vectors, summed by the same pass
    (lfpsqp_gram_rhs)."""
    ncols = M.m if ncols is None else ncols
    G = np.empty((ncols, ncols), order='F')
    X = np.zeros((ncols, max(len(es), 1)), order='F')
    arr = (C.c_void_p * max(len(es), 1))(*[e.h for e in es])
    M.ctx.check(M.ctx.L.lfpsqp_gram_rhs(M.ctx.h, M.h, ncols, w2.h if w2 is not None else None, len(es), arr, G.ctypes.data, X.ctypes.data))
    return G, X[:, :len(es)]


def rmul(In: DeviceMatrix, W: np.ndarray, Out: DeviceMatrix) -> DeviceMatrix:
    """Out[:, :W.shape[1]] = In[:, :W.shape[0]] @ W."""
    W = np.asfortranarray(W, dtype=np.float64)
    In.ctx.check(In.ctx.L.lfpsqp_rmul(In.ctx.h, In.h, W.shape[0], W.ctypes.data, W.shape[1], Out.h))
    return Out


def ksvd_(Jct: DeviceMatrix | None, Z: DeviceMatrix | None, w2: DeviceVector | None = None, eps_rank: float = 1e-10,
          W: np.ndarray | None = None, Jsp=None, Vt_prev: np.ndarray | None = None, rhs: DeviceVector | None = None,
          G_out: np.ndarray | None = None):
    """Thin factorisation diag(sqrt(w2)) Jct = U S Vt with U = diag(sqrt(w2)) Z.
    Returns (Sigma, Vt, rank); Z is overwritten (Jct is NOT destroyed, unlike dgesvd).  ``Z = None`` (dense Jct, ``W`` required): the
    basis Z = Jct @ W is not formed -- the caller keeps it in factored form (DeviceBasis(None, rank, generator=(Jct, W))).
    ``W`` (optional, m x m Fortran-ordered float64) receives the small factor with Z = Jct @ W.
    ``Jsp`` (optional SparseMatrix with the entries of the leading ``Jsp.m`` columns of Jct; Jct may then be None when there are no
    further columns): the basis-forming products stream the nonzeros (lfpsqp_factorize_sp).
    ``Vt_prev`` (optional, the Vt of a previous call on a nearby matrix): warm start of the small eigenproblem (lfpsqp_factorize_hint).
    ``rhs`` (optional device n-vector e; dense Jct only): returns a fourth value Jct' (sqrt(w2) .* e), summed during the Gram pass
    (lfpsqp_factorize_rhs) -- the outer iteration's Jct'd without a GEMV-T pass of its own; ``G_out`` (m x m, Fortran order) then receives the
    Gram matrix Jct' diag(w2) Jct the factors were computed from."""
    m = Jct.m if Jct is not None else Jsp.m
    ctx = Jct.ctx if Jct is not None else Jsp.ctx
    if W is not None:
        assert W.shape == (m, m) and W.flags.f_contiguous and W.dtype == np.float64
    S = np.zeros(m)
    Vt = np.zeros((m, m), order='F')
    rank = c_i64()
    if Vt_prev is not None and Vt_prev.shape == (m, m):
        vp = np.asfortranarray(Vt_prev, dtype=np.float64)
        ctx.check(ctx.L.lfpsqp_factorize_hint(ctx.h, vp.ctypes.data, m))
    if Jsp is not None:
        ctx.check(ctx.L.lfpsqp_factorize_sp(ctx.h, Jsp.h, Jct.h if Jct is not None else None, w2.h if w2 is not None else None,
                                            Z.h if Z is not None else None,
                                            S.ctypes.data, Vt.ctypes.data, W.ctypes.data if W is not None else None, C.byref(rank),
                                            float(eps_rank)))
    elif rhs is not None:
        Jte = np.zeros(m)
        ctx.check(ctx.L.lfpsqp_factorize_rhs(ctx.h, Jct.h, w2.h if w2 is not None else None, Z.h if Z is not None else None, S.ctypes.data,
                                             Vt.ctypes.data, W.ctypes.data if W is not None else None, C.byref(rank),
                                             float(eps_rank), rhs.h, Jte.ctypes.data, G_out.ctypes.data if G_out is not None else None))
        return S, Vt, rank.value, Jte
    else:
        ctx.check(ctx.L.lfpsqp_factorize(ctx.h, Jct.h, w2.h if w2 is not None else None, Z.h if Z is not None else None, S.ctypes.data,
                                         Vt.ctypes.data, W.ctypes.data if W is not None else None, C.byref(rank),
                                         float(eps_rank)))
    assert rhs is None, "rhs: dense Jct only"
    return S, Vt, rank.value


def small_svd_(ctx, A: np.ndarray, want_v: bool = True):
    """Thin SVD of a small replicated host matrix by one-sided Jacobi (lfpsqp_small_svd: the m x m step of the tangent
    setup; on the device from 64 columns on).  Returns (U, S, V) with A = U diag(S) V'."""
    A = np.asfortranarray(A, dtype=np.float64)
    rows, cols = A.shape
    U = np.zeros((rows, cols), order='F')
    S = np.zeros(cols)
    V = np.zeros((cols, cols), order='F') if want_v else None
    ctx.check(ctx.L.lfpsqp_small_svd(ctx.h, rows, cols, A.ctypes.data, U.ctypes.data, S.ctypes.data, V.ctypes.data if want_v else None))
    return U, S, V


def orthonormalize_(Z: DeviceMatrix, n_global: int | None = None) -> DeviceMatrix:
    """Replace the columns of Z by an orthonormal basis of their span (in place from the caller's
    point of view; uses one scratch matrix of the same size)."""
    tmp = DeviceMatrix(Z.ctx, Z.n, Z.m)
    tmp.copy_from(Z)
    ksvd_(tmp, Z)
    tmp.free()
    return Z
