"""Timing of the replicated small step (lfpsqp_small_svd: device block Jacobi) on graded random matrices: python tools/time_small_svd.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import lfpsqp_jl_amd as L
ctx = L.Context(0)
for m, want_v in ((128, False), (128, True), (256, False), (512, False), (512, True)):
    rng = np.random.default_rng(m)
    A = np.linalg.cholesky((lambda X: X.T @ X)(rng.standard_normal((4 * m, m)))).T.copy()
    best = 1e9
    for rep in range(4):
        ctx.sync(); t = time.perf_counter(); U, S, V = L.small_svd_(ctx, A, want_v); dt = time.perf_counter() - t
        if rep: best = min(best, dt)
    S0 = np.linalg.svd(A, compute_uv=False)
    print(f"m={m} want_v={want_v}: {best*1e3:.2f} ms  max rel err sigma {np.max(np.abs(S-S0)/S0):.1e}  orth U {np.abs(U.T@U-np.eye(m)).max():.1e}")
