"""Where the FIRST outer iteration of `optimize` spends its host time at n = 1e7, m = 128 (config 3's shape): cProfile of a one-iteration run.
    python tools/profile_setup.py [n] [m]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = L.Context(0)
Jct = ctx.matrix(n, m, placed=True).hash_fill(1)
xs = ctx.vector(n).hash_fill(2)
b = ctx.vector(m); L.gemv_t(Jct, xs, b)
P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download())
x0 = np.ones(n)
par = L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=1)
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    P.optimize(x0, par)
    ctx.sync(); pr.disable()
    print(f"run {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms")
    if rep == 1:
        st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
