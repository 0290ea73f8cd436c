"""How do the FAILING trial retractions of config 4's line searches fail?  Per batched call: flag, iterations and the final |c|max of every trial
(finite / NaN / Inf) -- decides whether a diverged trial could be retired early.   python tools/c4_trial_fates.py [n] [m] [outer iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
import lfpsqp_jl_amd.linesearch as LS

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
its = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = L.Context(0)
real = LS.retract_nr_batch_
log = []


def spy(cvs, xns, c_, xts, x, method):
    got = real(cvs, xns, c_, xts, x, method)
    if got is not None:
        log.append([(f, it, float(np.max(np.abs(cvs[b]))) if np.all(np.isfinite(cvs[b])) else float("nan")) for b, (f, it, _) in enumerate(got)])
    return got


LS.retract_nr_batch_ = spy
# config 4 as tools/run_config.py builds it
Jct = ctx.matrix(n + 1, m + 1, placed=True).hash_fill(1, 0, n, 1.0, n, m)
xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0); ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
b = ctx.vector(m + 1); L.gemv_t(Jct, xs, b, ncols=m)
i = np.arange(n)
xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu)
x, obj, lam, ti = P.optimize(0.5 * np.ones(n), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.iter, maxiter=its))
print("maxiter_retract", L.LFPSQPParams().maxiter_retract)
for k, call in enumerate(log):
    print(f"batched call {k}: " + "  ".join(f"[{f} {it} {c:.1e}]" for f, it, c in call))
