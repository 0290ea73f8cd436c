"""The one-pass tangent step (lfpsqp_tangent_step with LFPSQP_TANGENT_INIT_PROJCG) at full size: kernel ms (profiling slot 6) and ms per call for
  plain     linear equalities (projection + projcg!'s initial projection: 2 first products, 3 staged vectors)
  stream    the nonlinear class with streamed gradients (+ phi'' term: 3 first products, 4 staged vectors; a view with row scales)
  dense     ... with the common quadratic term (a view with a rank-one term as well)
    python tools/time_tangent.py [n] [m] [--modes plain,stream,dense] [--reps 20] [--lib variant.so]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L


def arg(name, default):
    if name in sys.argv:
        k = sys.argv.index(name); v = sys.argv[k + 1]; del sys.argv[k:k + 2]; return v
    return default


modes = arg("--modes", "plain,stream,dense").split(","); reps = int(arg("--reps", "20")); libpath = arg("--lib", None)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = L.Context(0, L.load_library(libpath) if libpath else None)
for mode in modes:
    A = ctx.matrix(n, m, placed=True).hash_fill(21, 0, n, 2.0 ** -11)
    cons = xv = None
    Jct = A
    if mode != "plain":
        cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=(np.arange(n) % 3).astype(np.float64),
                                        qw=(1e-7 * np.cos(np.arange(m)) if mode == "dense" else None), stream=True)
        xv = ctx.vector(n).hash_fill(31, 0, 0.5, 0.0)
        cons.jac_(cons.Jct, np.zeros(m), xv)
        Jct = cons.Jct
    d = ctx.vector(n).hash_fill(8, 0)
    W = np.zeros((m, m), order='F'); G = np.zeros((m, m), order='F')
    S, Vt, rank, Jtd = L.ksvd_(Jct, None, W=W, rhs=d, G_out=G)
    U = L.DeviceBasis(None, rank, generator=(Jct, W))
    work = L.ProjCGWork(ctx, n, m, against=A)
    hd = ctx.vector(n).hash_fill(9, 0, 4.0, 5.0)
    Utd, lam = np.zeros(m), np.zeros(m); ss = C.c_double()
    bs, wc = U._c(), work._c(); cc = cons._c() if cons is not None else None

    def call():
        ctx.check(ctx.L.lfpsqp_tangent_step(ctx.h, C.byref(bs), S.ctypes.data, Vt.ctypes.data, m, Jtd.ctypes.data, G.ctypes.data, d.h,
                                            C.byref(cc) if cc is not None else None, xv.h if xv is not None else None, hd.h, None, None, None, None,
                                            C.byref(wc), 1, Utd.ctypes.data, lam.ctypes.data, C.byref(ss)))
    for _ in range(3):
        call()
    ctx.set_profiling(True); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        call()
    ctx.sync(); wall = (time.perf_counter() - t0) * 1e3 / reps
    pms, pcnt = ctx.profile_read(); ctx.set_profiling(False)
    kern = pms[6] / pcnt[6] if pcnt[6] else float("nan")
    nvec = {"plain": 3 + 3, "stream": 6 + 4, "dense": 7 + 4}[mode]           # n-vectors read + written
    gb = (8.0 * n * m + 8.0 * n * nvec) / 1e9
    print(f"n={n} m={m} {mode:6s}: tangent step kernel {kern:.3f} ms ({gb / kern * 1e3:.0f} GB/s algorithmic = {gb / kern / 8.0:.3f} of 8 TB/s), {wall:.3f} ms per call")
    for v in (A, d, hd):
        v.free()
