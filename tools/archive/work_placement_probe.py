"""Does the PLACEMENT of the n-vectors decide the speed of the fused projected-CG kernel F?  One process, one basis Z; K work sets (g, d, rp) and
x / b / a vectors allocated side by side; F timed (library profiler slots) on each set in turn, three rounds.   python tools/work_placement_probe.py [K]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = L.Context(0)
n, m = 10_000_000, 128
Z = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
L.orthonormalize_(Z, n_global=n)
U = L.DeviceBasis(Z)
A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0))
b = ctx.vector(n).hash_fill(4)
sets = []
pad = []
for k in range(K):
    sets.append((ctx.vector(n), L.ProjCGWork(ctx, n, m)))
    pad.append(ctx.vector(1_000_003 * (k + 1)))          # shifts the next set's placement
# warm the device
for _ in range(12):
    L.projcg_(sets[0][0], None, A, U, b, None, tol=1e-300, maxit=50, work=sets[0][1], n_global=n, want_lambda=False)
for rnd in range(3):
    out = []
    for k, (x, w) in enumerate(sets):
        L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=5, work=w, n_global=n, want_lambda=False)
        ctx.set_profiling(True)
        ctx.sync(); t0 = time.perf_counter()
        it, _ = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=40, work=w, n_global=n, want_lambda=False, resume=True)
        ctx.sync(); dt = time.perf_counter() - t0
        ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
        out.append(f"set {k}: F {ms[3] / max(cnt[3], 1):.3f} ms, {40 / dt:6.1f} it/s")
    print(f"round {rnd}: " + " | ".join(out), flush=True)
