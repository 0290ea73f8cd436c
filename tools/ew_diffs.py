"""Per-outer-iteration distance between the device run and the oracle for the end-to-end cases of tests/test_elementwise.py (EMU=<emulator lib> N=<rows>; ACTIVE_BOUNDS=1: the ill-conditioned variant with active bounds)."""
import sys; sys.path.insert(0, '.')
import numpy as np
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from tests.test_elementwise import _systems, ew_callables
import os
ctx = L.Context(0, L.load_library(os.environ["EMU"]) if "EMU" in os.environ else None)
N = int(os.environ.get("N", "10000"))
for system, bounds, project in [("sin-sparse", True, False), ("sin-sparse", False, False), ("sphere", False, False), ("mixed-dense", False, False), ("sin-dense", False, True)]:
    rng = np.random.default_rng(3)
    n, m = N, 64
    cons, A, kind, qw, b = _systems(ctx, n, m, rng)[system]
    c_, jac_, hdiag = ew_callables(A, kind, qw, b)
    target = 0.5 * rng.standard_normal(n)
    if bounds and os.environ.get("ACTIVE_BOUNDS") is None: target = np.clip(target, -0.8, 0.8)
    x0 = np.zeros(n)
    if system == "mixed-dense": x0 = 0.2 * rng.standard_normal(n)
    xl = xu = None
    if bounds:
        xl = np.where(np.arange(n) % 4 == 1, -1.0, np.where(np.arange(n) % 4 == 3, -1.0, -np.inf))
        xu = np.where(np.arange(n) % 4 == 2, 1.0, np.where(np.arange(n) % 4 == 3, 1.0, np.inf))
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target, xl=xl, xu=xu)
    p = L.LFPSQPParams(do_project_retract=project, maxiter=12, disp=L.DisplayOption.off)
    tr = []; xd, obj, lam, ti = prob.optimize(x0, p, trace=tr)
    f = lambda xx: float(np.sum((xx[:n] - target) ** 2))
    def grad_(g, xx): g[:n] = 2.0 * (xx[:n] - target)
    def hlv_(dest, src, xx, lam_): dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]
    p0 = R.LFPSQPParams(do_project_retract=project, maxiter=12, disp=R.DisplayOption.off)
    tr0 = []; R.optimize_core(f, grad_, c_, jac_, hlv_, x0, xl, xu, m, p0, trace=tr0)
    print(system, bounds, project, len(tr), len(tr0))
    for k, (a, bb) in enumerate(zip(tr, tr0)):
        print("  ", k, "%.2e" % (np.linalg.norm(a["x"] - bb["x"]) / max(np.linalg.norm(bb["x"]), 1)), [(a.get(key), bb.get(key)) for key in ("tn_iter", "retract_iter1", "alpha", "steptype")], "kkt %.3e %.3e" % (a["kkt_diff"], bb["kkt_diff"]))
