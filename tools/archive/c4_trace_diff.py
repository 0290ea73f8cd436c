"""Config-4 trajectory at test size on the device vs the oracle: per outer iteration, iterate deviation and any count that differs."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R, synth
ctx = L.Context(0, L.load_library(sys.argv[1]) if len(sys.argv) > 1 else None)
n, m = 4000, 16
P0 = synth.BallBoxProblem(n, m)
tr0, tr = [], []
xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, P0.x0, P0.xl, P0.xu, P0.m, P0.p,
                                 R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=10000),
                                 derivatives=P0.derivatives(), trace=tr0)
Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
x, obj, lam, ti = P.optimize(P0.x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=10000), trace=tr)
print(ti.iter, tir.iter, len(tr), len(tr0))
for a, b in zip(tr, tr0):
    dx = np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x'])
    diffs = {k: (a.get(k), b.get(k)) for k in ('tn_iter', 'steptype', 'mtype', 'retract_iter1', 'alpha', 'ls_flag', 'rank') if a.get(k) != b.get(k)}
    print(a['iter'], f"{dx:.2e}", diffs, a.get('retract_iter1'))
