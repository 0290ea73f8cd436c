"""Seconds per call / per outer iteration of the device-resident nonlinear constraint class at full size (n = 1e7, m = 128, one MI355X):
c!, jac!, hess_diag!, tangent setup, and an `optimize` run (f = |x - target|^2) with the Newton and the ProjPenalty retraction.
    python tools/time_elementwise.py [dense|dense-materialised|sparse|stream|nostream] > gpurun_out/elementwise_<kind>.json
dense: mixed kinds + the common quadratic term, gradients STREAMED through a view diag(phi'(x)) A + 2 x qw' of A (the default for a dense class);
dense-materialised: the same with jac! writing an n x m Jct (the A/B); stream / nostream: the same pair without the quadratic term."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_gpu_fullsize import _ew_big

mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
sparse = mode == "sparse"
ctx = L.Context(0)
if mode in ("stream", "nostream"):
    from tests.test_gpu_fullsize import N as n, M as m
    A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
    cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=(np.arange(n) % 3).astype(np.float64), stream=(mode == "stream"))
elif mode == "dense-materialised":
    from tests.test_gpu_fullsize import N as n, M as m
    A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
    cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=(np.arange(n) % 3).astype(np.float64), qw=1e-7 * np.cos(np.arange(m)), stream=False)
else:
    cons, n, m = _ew_big(ctx, sparse)
x = ctx.vector(n).hash_fill(31, 0, 0.5, 0.0)
cv = np.zeros(m)


def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


out = {"n": n, "m": m, "A": "sparse, 4 nonzeros per row" if sparse else "dense", "mode": mode, "streamed_gradients": bool(getattr(cons, "streamed", False)),
       "device": ctx.device_name}
out["c_ms"] = timed(lambda: cons.c_(cv, x))
out["jac_ms"] = timed(lambda: cons.jac_(cons.Jct, cv, x))
hx = ctx.vector(n)
lam = np.cos(1.0 + np.arange(m))
out["hess_diag_ms"] = timed(lambda: cons.hess_diag_(hx, x, lam))
Z = ctx.matrix(n, m); W = np.zeros((m, m), order='F')
out["tangent_setup_ms"] = timed(lambda: L.ksvd_(cons.Jct, Z, W=W, Jsp=cons.Jsp), 3)
out["tangent_setup_factored_ms"] = timed(lambda: L.ksvd_(cons.Jct, None, W=W, Jsp=cons.Jsp), 3)
# Newton retraction, 12 iterations forced by tol = 0 (flag 1): ms per iteration with and without the generator hint (Z = Jct W)
Sg, Vtg, rank = L.ksvd_(cons.Jct, Z, W=W, Jsp=cons.Jsp)
xt, xnew = ctx.vector(n), ctx.vector(n)
pert = ctx.vector(n).hash_fill(51, 0, 1e-3, 0.0)
for tag, U in (("with_generator", L.DeviceBasis(Z, generator=(cons.Jct, W))), ("basis_only", L.DeviceBasis(Z))):
    nr = L.NR(U, Sg, Vtg, 0.0, 12, L.NRWork(m), False, None)
    best = 1e9
    for rep in range(3):
        xt.copy_from(x); L.axpby(1.0, pert, 1.0, xt)
        ctx.sync(); t0 = time.perf_counter()
        flag, it, _ = L.retract_(cv, xnew, cons, xt, x, nr)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3 / max(it, 1))
    out["nr_iteration_ms_" + tag] = best
Z.free()
cons.jac_(cons.Jct, cv, x)
cons.b = cons.b + cv                       # x feasible; the optimum of |x - target|^2 on the manifold lies nearby
target = ctx.vector(n).hash_fill(41, 0, 0.5, 0.0)
L.axpby(0.98, x, 0.02, target)
# outer iterations timed one by one: a callback after every iteration takes a timestamp (device idle), tolerances off so that `iters` iterations run;
# the first interval (allocation by trial, uploads) is reported apart from the steady ones
iters = 6
for name, dpr in (("newton", False), ("projpenalty", True)):
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target.download())
    x0h = x.download()
    tr = []
    par = L.LFPSQPParams(do_project_retract=dpr, maxiter=iters, disp=L.DisplayOption.off, eps_kkt=0.0, eps_f=-1.0, eps_x=-1.0)
    xs, obj, lamk, ti = prob.optimize(x0h, par, trace=tr)                      # counts (the trace downloads x every iteration: not timed)
    best = None
    for rep in range(2):
        stamps = []

        def cb(i, xx):
            ctx.sync()
            stamps.append(time.perf_counter())
        ctx.sync(); t0 = time.perf_counter()
        prob.optimize(x0h, L.LFPSQPParams(do_project_retract=dpr, maxiter=iters, disp=L.DisplayOption.off, eps_kkt=0.0, eps_f=-1.0, eps_x=-1.0,
                                          callback=cb, callback_period=1))
        d = np.diff(np.array([t0] + stamps)) * 1e3
        if best is None or np.median(d[1:]) < np.median(best[1:]):
            best = d
    out["optimize_" + name] = {"outer_iterations": ti.iter, "condition": ti.condition.name, "ms_first_iteration_with_setup": float(best[0]),
                               "ms_per_outer_iteration": [float(v) for v in best[1:]], "seconds_per_outer_iteration": float(np.median(best[1:])) * 1e-3,
                               "objective": [float(v) for v in obj],
                               "retraction_iterations": [d_.get("retract_iter1") for d_ in tr], "tn_iterations": [d_.get("tn_iter") for d_ in tr],
                               "cmax_final": float(np.abs(tr[-1]["cval"]).max())}
print(json.dumps(out, indent=1))
