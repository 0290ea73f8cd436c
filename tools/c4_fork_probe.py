"""Config 4 at the test size (n = 4000, m = 16, start from P0.x0: linesearches with failed retractions) on the GPU against the oracle, under the
device options that reshape the arithmetic of an outer iteration: where does the accepted step of a chaotic linesearch leave the oracle's?
    python tools/c4_fork_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

n, m = 4000, 16
P0 = synth.BallBoxProblem(n, m)
tr0 = []
xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, P0.x0, P0.xl, P0.xu, P0.m, P0.p, R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=10000),
                                 derivatives=P0.derivatives(), trace=tr0)
print("oracle:", tir.iter, "outer iterations; (alpha, Newton iterations) per iteration:", [(t.get('alpha'), t.get('retract_iter1')) for t in tr0][:12])
ctx = L.Context(0)
for name, opts in (("defaults", {}), ("fused_tangent_step off", dict(fused_tangent_step=False)), ("ls_batch 1", dict(ls_batch=1)), ("ls_batch 4", dict(ls_batch=4)),
                   ("warm_factorize off", dict(warm_factorize=False)), ("factored_basis off", dict(factored_basis=False)),
                   ("fused_tangent_step off, ls_batch 1", dict(fused_tangent_step=False, ls_batch=1)),
                   ("all off", dict(fused_tangent_step=False, ls_batch=1, warm_factorize=False, factored_basis=False))):
    saved = {k: getattr(ctx.options, k) for k in opts}
    for k, v in opts.items():
        setattr(ctx.options, k, v)
    tr = []
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    x, obj, lam, ti = P.optimize(P0.x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=10000), trace=tr)
    for k, v in saved.items():
        setattr(ctx.options, k, v)
    fork = None
    for a, b in zip(tr, tr0):
        if a.get('alpha') != b.get('alpha'):
            fork = a['iter']; break
    dev = max(np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x']) for a, b in zip(tr[:(fork or len(tr))], tr0))
    print(f"{name:40s}: {ti.iter} outer iterations, first differing accepted alpha at iteration {fork}, max deviation before it {dev:.1e}; "
          f"(alpha, Newton its): {[(t.get('alpha'), t.get('retract_iter1')) for t in tr][:8]}")
