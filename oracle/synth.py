"""Synthetic inputs of SURVEY §8(d) -- numpy side (TEST INFRASTRUCTURE, see
oracle/__init__.py).  The device library has its own generator kernel
(lfpsqp.jl_amd/csrc/kernels_vec.hip: hash_fill_*) that must produce bit-identical
values; tests/test_gpu_primitives.py checks that.

u(seed, k): splitmix64 finaliser used as a hash of the 0-based flat index k,
mapped to [-1, 1) with one exact scaling (no transcendental functions, so CPU
and GPU agree bit for bit).
"""
from __future__ import annotations

import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash_u(seed: int, k) -> np.ndarray:
    """u(seed, k) in [-1, 1) for an integer array (or scalar) of flat indices k."""
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + (k + np.uint64(1)) * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0


def hash_vector(seed: int, n: int, offset: int = 0) -> np.ndarray:
    return hash_u(seed, np.arange(offset, offset + n, dtype=np.uint64))


def hash_matrix(seed: int, n: int, m: int, row0: int = 0, nrows: int | None = None, n_total: int | None = None) -> np.ndarray:
    """Column-major n x m matrix M[i, j] = u(seed, j*n_total + i) (rows row0..row0+nrows
    of it when sharded)."""
    if n_total is None:
        n_total = n
    if nrows is None:
        nrows = n
    out = np.empty((nrows, m), order='F')
    idx = np.arange(row0, row0 + nrows, dtype=np.uint64)
    for j in range(m):
        out[:, j] = hash_u(seed, np.uint64(j) * np.uint64(n_total) + idx)
    return out


# ---------------------------------------------------------------------------
# BASELINE.json configs as plain-python problems for the oracle
# ---------------------------------------------------------------------------

class QuadLinearProblem:
    """f = ||x - xc||^2 (xc = 0 for the BASELINE configs), c(x) = J x - b with dense J
    (C2: J = e1', b = 0.75; C3: hash matrix)."""

    def __init__(self, Jct: np.ndarray, b: np.ndarray, xc: float = 0.0):
        self.Jct = np.asfortranarray(Jct)
        self.b = np.asarray(b, dtype=float)
        self.n, self.m = self.Jct.shape
        self.xc = xc

    def f(self, x):
        dx = x - self.xc
        return float(np.dot(dx, dx))

    def grad_(self, g, x):
        g[:] = 2.0 * (x - self.xc)

    def c_(self, cval, x):
        cval[:self.m] = self.Jct.T @ x - self.b
        return cval

    def jac_(self, J, cval, x):
        J[:, :] = self.Jct.T
        cval[:self.m] = self.Jct.T @ x - self.b

    def hess_lag_vec_(self, dest, src, x, lam):
        dest[:] = 2.0 * src


def config2(n: int):
    """C2: f = x'x, c(x) = x_1 - 0.75, x0 = ones (reference README.md:42-53)."""
    Jct = np.zeros((n, 1), order='F')
    Jct[0, 0] = 1.0
    return QuadLinearProblem(Jct, np.array([0.75])), np.ones(n)


def config3(n: int, m: int):
    """C3: Jct[i,j] = u(1, j*n+i), x*_i = u(2, i), b = J x*, f = x'x, x0 = ones."""
    Jct = hash_matrix(1, n, m)
    xstar = hash_vector(2, n)
    return QuadLinearProblem(Jct, Jct.T @ xstar), np.ones(n)


class BallBoxProblem:
    """C4: C3's equalities + ball d(x) = x'x - R^2 <= 0 + the four-way bound pattern of
    reference test/test_inequalities.jl:6-9 (i mod 4: none / lower -1 / upper +1 / both)."""

    def __init__(self, n: int, m: int, xc: float = 0.0, Jct=None, b=None):
        # (Jct / b: the same hash matrix and right-hand side built elsewhere -- oracle/port.py makes them in seconds at n = 1e7)
        self.eq = QuadLinearProblem(hash_matrix(1, n, m) if Jct is None else Jct, None, xc)
        self.eq.b = self.eq.Jct.T @ hash_vector(2, n) if b is None else np.asarray(b, dtype=float)
        self.n, self.m, self.p = n, m, 1
        self.R2 = n / 2.0
        i = np.arange(n)
        self.xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
        self.xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
        self.x0 = 0.5 * np.ones(n)

    f = property(lambda self: self.eq.f)
    c_ = property(lambda self: self.eq.c_)

    def d_(self, dval, x):
        dval[0] = float(np.dot(x, x)) - self.R2

    def derivatives(self):
        from .lfpsqp_ref import Derivatives
        eq = self

        def jac_d_(J, dval, x):
            J[0, :] = 2.0 * x
            dval[0] = float(np.dot(x, x)) - eq.R2

        def hlv_(dest, src, x, lam):
            # ∇²f = 2I, linear equalities contribute 0, ball contributes 2*lam_ball*I
            dest[:] = (2.0 + 2.0 * lam[eq.m]) * src

        return Derivatives(grad_=self.eq.grad_, hess_lag_vec_=hlv_, jac_c_=self.eq.jac_, jac_d_=jac_d_)
