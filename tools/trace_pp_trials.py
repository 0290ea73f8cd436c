"""Per-retraction log of config 4 with the reference's DEFAULT retraction (ProjPenalty): for every trial retraction of the first outer
iterations, (flag, Gauss-Newton steps, inner pcg! iterations, seconds).   python tools/trace_pp_trials.py [max_outer] [--precond]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import projpenalty as PPm, retractions as R

mo = int([a for a in sys.argv[1:] if not a.startswith('--')][0]) if any(not a.startswith('--') for a in sys.argv[1:]) else 3
ctx = L.Context(0)
n, m = 10_000_000, 128
Jct = ctx.matrix(n + 1, m + 1, placed=True).hash_fill(1, 0, n, 1.0, n, m)
xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0); ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
b = ctx.vector(m + 1); L.gemv_t(Jct, xs, b, ncols=m)
i = np.arange(n)
xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu); x0 = 0.5 * np.ones(n)
ctx.options.pp_precondition = '--precond' in sys.argv
log = []
orig = PPm.retract_pp
def traced(cval, xnew, c_, xtilde, x, method):
    ctx.sync(); t = time.perf_counter()
    out = orig(cval, xnew, c_, xtilde, x, method)
    ctx.sync(); log.append((out[0], out[1], out[2], time.perf_counter() - t))
    return out
PPm.retract_pp = traced
par = L.LFPSQPParams(do_project_retract=True); par.maxiter = mo
t0 = time.perf_counter()
x, obj, lam, ti = P.optimize(x0, par)
ctx.sync(); dt = time.perf_counter() - t0
print(f"{mo} outer iterations: {dt:.1f} s wall, {len(log)} retractions, {sum(l[3] for l in log):.1f} s inside them; maxiter {par.maxiter_retract if hasattr(par,'maxiter_retract') else '?'}")
print("flag  GN  pcg   seconds")
for l in log: print(f"{l[0]:4d} {l[1]:3d} {l[2]:5d} {l[3]:8.2f}")
ok = [l for l in log if l[0] == 0]; bad = [l for l in log if l[0] != 0]
print(f"successful: {len(ok)} retractions, {sum(l[2] for l in ok)} pcg iterations, {sum(l[3] for l in ok):.1f} s;  failed: {len(bad)}, {sum(l[2] for l in bad)} pcg iterations, {sum(l[3] for l in bad):.1f} s")
