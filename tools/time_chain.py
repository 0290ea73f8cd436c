"""`optimize` on a chain-objective problem (ChainSeparableLinear: separable objective + kappa/2 sum (x_{i+1} - x_i)^2 under dense equalities)
at n = 1e7, m = 128: wall time, outer iterations, truncated-Newton iterations -- with the tridiagonal Hessian on the one-pass solver
(lfpsqp_projcg_tridiag) and through the callback path (lfpsqp_projcg_op), same trajectory.
    python tools/time_chain.py [n] [m] [max_outer]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mo = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = L.Context(0)
Jct = ctx.matrix(n, m, placed=True).hash_fill(1)
xs = ctx.vector(n).hash_fill(2)
b = ctx.vector(m); L.gemv_t(Jct, xs, b)
a = 0.5 + np.linspace(0.0, 1.0, n) ** 2
P = L.ChainSeparableLinear(ctx, n, m, Jct, b.download(), 1, a, 0.0, kappa=1.5)
x0 = np.cos(np.arange(n) * 1e-3)
res = {}
for one_pass in (True, False, True):
    ctx.options.tridiagonal_one_pass = one_pass
    tr = []
    ctx.sync(); t0 = time.perf_counter()
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=mo, tn_kappa=1e-4), trace=tr)
    ctx.sync(); dt = time.perf_counter() - t0
    tn = [t.get('tn_iter') or 0 for t in tr]
    res[one_pass] = (dt, ti.iter, sum(tn), obj[-1], x)
    print(f"n={n} m={m} one_pass={one_pass}: {dt:.3f} s for {ti.iter} outer iterations ({ti.condition.name}), {sum(tn)} truncated-Newton iterations {tn}, f = {obj[-1]:.12e}")
d = np.linalg.norm(res[True][4] - res[False][4]) / np.linalg.norm(res[False][4])
print(f"|x_one_pass - x_callback| / |x| = {d:.2e}; truncated-Newton iterations cost {1e3 * (res[False][0] - res[True][0]) / max(res[True][2], 1):.3f} ms less each on the one-pass path "
      f"(set-up of U'AU included)")
