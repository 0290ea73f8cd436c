"""Batched Newton-retraction step at full size (lfpsqp_retract_nr_batch): ms per Newton step for nb trial points sharing the pass over Jct,
with and without bounds, on the VALU form (nb <= 4) and on the matrix cores (5 .. 16, or every nb with LFPSQP_NRB_MFMA=1 in a second process).
    python tools/time_nrbatch.py [n] [m] [--bounds 0|1] [--nbs 4,8,16] [--iters 60]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd.inequality import InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_, inequality_gradient_


def arg(name, default):
    if name in sys.argv:
        k = sys.argv.index(name); v = sys.argv[k + 1]; del sys.argv[k:k + 2]; return v
    return default


bounds = int(arg("--bounds", "0")); nbs = [int(v) for v in arg("--nbs", "4,8,16").split(",")]; iters = int(arg("--iters", "60"))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = L.Context(0)
J = ctx.matrix(n, m, placed=True).hash_fill(1, 0, n)
xs_h = None
if bounds:
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    idata = InequalityData(ctx, xl, xu)
    xs = StackedVector(ctx, n)
    xs.upload(0.6 * np.sin(0.001 * i), 0)
    generate_initial_y_(xs, idata)
    dec = InequalityDecomp(ctx, n, m, J, factored=True)
    inequality_gradient_(dec, xs, idata)
    W = np.zeros((m, m), order="F")
    S, Vt, rank = L.ksvd_(J, None, w2=dec.sx, W=W)
    dec.rank = rank; dec.W = W
    basis = InequalityDecompProject(dec)
    mk = lambda: StackedVector(ctx, n)
else:
    idata = None
    xs = ctx.vector(n).hash_fill(2, 0)
    W = np.zeros((m, m), order="F")
    S, Vt, rank = L.ksvd_(J, None, W=W)
    basis = L.DeviceBasis(None, m, generator=(J, W))
    mk = lambda: ctx.vector(n)
bdev = ctx.vector(m)
if bounds:
    xh = ctx.vector(n, 0.6 * np.sin(0.001 * np.arange(n)))
    L.gemv_t(J, xh, bdev); xh.free()
else:
    L.gemv_t(J, xs, bdev)
cons = L.DeviceConstraints(J, m, bdev.download())
pert = mk()
if bounds:
    pert.upload(1e-3 * np.cos(0.002 * np.arange(n)), 0)
else:
    pert.hash_fill(7, 0, 1e-3, 0.0)
res = {}
for nb in nbs:
    xts, xns = [mk() for _ in range(nb)], [mk() for _ in range(nb)]
    for j, xt in enumerate(xts):
        L.waxpby(1.0, xs, 0.5 ** j, pert, xt)
    nr = L.NR(basis, S, Vt, 0.0, iters, L.NRWork(m), bool(bounds), idata)
    cvs = np.zeros((nb, m))
    best = None
    for rep in range(3):
        ctx.set_profiling(True)
        ctx.sync(); t0 = time.perf_counter(); got = L.retract_nr_batch_(cvs, xns, cons, xts, xs, nr); ctx.sync()
        wall = (time.perf_counter() - t0) * 1e3 / max(got[0][1], 1)
        pms, pcnt = ctx.profile_read(); ctx.set_profiling(False)
        kern = pms[7] / pcnt[7] if pcnt[7] else float("nan")
        best = (wall, kern) if best is None or wall < best[0] else best
    # one trial alone, same point: the batched result must be that of lfpsqp_retract_nr
    one, cv = mk(), np.zeros(m)
    nr1 = L.NR(basis, S, Vt, 0.0, 6, L.NRWork(m), bool(bounds), idata)
    L.retract_(cv, one, cons, xts[nb - 1], xs, nr1)
    nrb = L.NR(basis, S, Vt, 0.0, 6, L.NRWork(m), bool(bounds), idata)
    L.retract_nr_batch_(cvs, xns, cons, xts, xs, nrb)
    a = xns[nb - 1].download2() if bounds else xns[nb - 1].download()
    b_ = one.download2() if bounds else one.download()
    dev = float(np.abs(a - b_).max() / max(1.0, np.abs(b_).max()))
    vec_bytes = (80.0 if bounds else 16.0) * n * nb + (64.0 * n if bounds else 0.0)
    res[nb] = {"ms_per_step": best[0], "kernel_ms": best[1], "GBs": (8.0 * n * m + vec_bytes) / best[0] / 1e6, "max_dev_vs_single": dev}
    print(f"n={n} m={m} bounds={bounds} nb={nb:2d}: {best[0]:.3f} ms per Newton step (kernel {best[1]:.3f}), {res[nb]['GBs']:.0f} GB/s algorithmic, "
          f"last trial vs single retraction after 6 steps: {dev:.1e}", flush=True)
    for v in xts + xns + [one]:
        v.free()
