"""Milliseconds per steady outer iteration of `optimize` on config 4's problem (ball in slack form + four-way bounds, Newton retraction) at
n = 1e7, m = 128 from a start near the feasible set (no failing trial retraction): the tangent step in one pass (DeviceOptions.fused_tangent_step,
lfpsqp_tangent_step with bounds) against the statement-by-statement sequence.      python tools/time_outer_bounds.py [n] [m]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
libpath = None
if '--lib' in sys.argv:
    k = sys.argv.index('--lib'); libpath = sys.argv[k + 1]; del sys.argv[k:k + 2]
ctx = L.Context(0, L.load_library(libpath) if libpath else None)
Jct = ctx.matrix(n + 1, m + 1, placed=True).hash_fill(1, 0, n, 1.0, n, m)
xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0); ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
b = ctx.vector(m + 1); L.gemv_t(Jct, xs, b, ncols=m)
i = np.arange(n)
xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu)
x0 = 0.9 * xs.download()[:n] + 0.1 * 0.5
iters = 6
res = {}
for fused in (True, False, True, False):
    ctx.options.fused_tangent_step = fused
    stamps = []

    def cb(k, xx):
        ctx.sync(); stamps.append(time.perf_counter())
    ctx.sync(); t0 = time.perf_counter()
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, maxiter=iters, disp=L.DisplayOption.off, eps_kkt=0.0, eps_f=-1.0,
                                                    eps_x=-1.0, callback=cb, callback_period=1))
    d = np.diff(np.array([t0] + stamps)) * 1e3
    res.setdefault(fused, []).append((float(np.median(d[1:])), obj[-1]))
    print(f"fused_tangent_step={fused}: ms per outer iteration {[round(float(v), 2) for v in d]}  objective {obj[-1]:.12e}", flush=True)
f, s = min(v[0] for v in res[True]), min(v[0] for v in res[False])
print(f"n={n} m={m} with ball and bounds: steady outer iteration {f:.2f} ms with the one-pass tangent step, {s:.2f} ms statement by statement; "
      f"objectives {res[True][0][1]:.12e} / {res[False][0][1]:.12e}")
