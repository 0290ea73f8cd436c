"""streamed vs materialised vs oracle at a chosen size, per outer iteration deviations (emulator or GPU)"""
import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else os.getcwd())
import numpy as np
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from tests.test_streamed_gradients import _mixed
from tests.test_elementwise import ew_callables
emu = len(sys.argv) > 1 and sys.argv[1] == "emu"
n, m = int(sys.argv[2]), int(sys.argv[3])
bounds, project, quad = (int(v) for v in sys.argv[4:7])
lib = L.load_library(os.path.join("tests", "emu", "_build", "liblfpsqp_emu.so")) if emu else L.load_library()
ctx = L.Context(0, lib)
runs = {}
for stream in (True, False):
    cons, Ah, kind, qw, bh, rng = _mixed(ctx, n, m, 41, stream, quad)
    target = 0.5 * rng.standard_normal(n)
    x0 = 0.2 * rng.standard_normal(n)
    xl = xu = None
    if bounds:
        target = np.clip(target, -0.8, 0.8)
        xl = np.where(np.arange(n) % 4 == 1, -1.0, np.where(np.arange(n) % 4 == 3, -1.0, -np.inf))
        xu = np.where(np.arange(n) % 4 == 2, 1.0, np.where(np.arange(n) % 4 == 3, 1.0, np.inf))
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target, xl=xl, xu=xu)
    tr = []
    prob.optimize(x0, L.LFPSQPParams(do_project_retract=bool(project), maxiter=6, disp=L.DisplayOption.off), trace=tr)
    runs[stream] = tr
c_, jac_, hdiag = ew_callables(Ah, kind, qw, bh)
f = lambda xx: float(np.sum((xx[:n] - target) ** 2))
def grad_(g, xx): g[:n] = 2.0 * (xx[:n] - target)
def hlv_(dest, src, xx, lam_): dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]
trr = []
R.optimize_core(f, grad_, c_, jac_, hlv_, x0, xl, xu, m, R.LFPSQPParams(do_project_retract=bool(project), maxiter=6, disp=R.DisplayOption.off), trace=trr)
for k in range(min(len(trr), len(runs[True]), len(runs[False]))):
    xr = trr[k]["x"]; nb = max(np.linalg.norm(xr), 1)
    print(k, "streamed-oracle %.2e  materialised-oracle %.2e  streamed-materialised %.2e" % (np.linalg.norm(runs[True][k]["x"] - xr) / nb, np.linalg.norm(runs[False][k]["x"] - xr) / nb,
          np.linalg.norm(runs[True][k]["x"] - runs[False][k]["x"]) / nb), {kk: (runs[True][k].get(kk), runs[False][k].get(kk), trr[k].get(kk)) for kk in ("tn_iter", "retract_iter1", "alpha", "steptype")})
