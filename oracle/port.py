"""ctypes wrapper of oracle/projcg_port.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libprojcg_port.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def usable_cpus() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a box can
    report 256 hardware threads while the container is limited to a few)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        i64, dbl, P = C.c_int64, C.c_double, C.c_void_p
        L.port_num_threads.restype = C.c_int
        L.port_set_num_threads.argtypes = [C.c_int]
        L.port_hash_matrix.argtypes = [P, i64, i64, i64, C.c_uint64, i64, i64, dbl]
        L.port_hash_vector.argtypes = [P, i64, C.c_uint64, i64, dbl, dbl]
        L.port_gemv_t.argtypes = [i64, i64, P, i64, P, P]
        L.port_gemv_n.argtypes = [i64, i64, dbl, P, i64, P, dbl, P]
        L.port_dot.argtypes = [i64, P, P]
        L.port_triad.argtypes = [i64, dbl, P, P, P]
        L.port_dot.restype = dbl
        L.port_projcg.argtypes = [i64, i64, P, P, i64, P, P, dbl, i64, P, P, P, C.POINTER(dbl)]
        L.port_projcg.restype = i64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data


def hash_matrix(seed, n, m, scale=1.0, row0=0, n_global=None):
    M = np.empty((n, m), order='F')
    lib().port_hash_matrix(_p(M), n, m, n, seed, row0, n if n_global is None else n_global, scale)
    return M


def hash_vector(seed, n, offset=0, scale=1.0, shift=0.0):
    v = np.empty(n)
    lib().port_hash_vector(_p(v), n, seed, offset, scale, shift)
    return v


def gemv_t(M, v):
    n, m = M.shape
    t = np.empty(m)
    lib().port_gemv_t(n, m, _p(M), n, _p(v), _p(t))
    return t


def gemv_n(M, t, y, alpha=1.0, beta=0.0):
    n, m = M.shape
    lib().port_gemv_n(n, m, alpha, _p(M), n, _p(t), beta, _p(y))
    return y


def triad(a, x, y, z):
    lib().port_triad(x.size, a, _p(x), _p(y), _p(z))
    return z


def projcg(adiag, U, b, c, tol, maxit):
    """Returns (x, lam, iters, nr)."""
    n, m = U.shape
    x = np.zeros(n)
    lam = np.zeros(m)
    work = np.empty(6 * n + max(m, 1))
    nr = C.c_double()
    c = np.zeros(m) if c is None else np.ascontiguousarray(c)
    it = lib().port_projcg(n, m, _p(adiag), _p(U), n, _p(b), _p(c), tol, maxit, _p(x), _p(lam), _p(work), C.byref(nr))
    return x, lam, int(it), nr.value
