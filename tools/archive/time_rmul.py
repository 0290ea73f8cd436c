"""rmul pass alone at full size (best of 5): python tools/time_rmul.py N M   (LFPSQP_LIB selects a variant build)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
ctx = L.Context(0, L.load_library(os.environ['LFPSQP_LIB']) if 'LFPSQP_LIB' in os.environ else None)
n, m = int(float(sys.argv[1])), int(sys.argv[2])
J = ctx.matrix(n, m).hash_fill(1); Z = ctx.matrix(n, m)
W = np.eye(m)
best = 1e9
for rep in range(6):
    ctx.sync(); t = time.perf_counter(); L.rmul(J, W, Z); ctx.sync(); dt = time.perf_counter() - t
    if rep: best = min(best, dt)
print(f"n={n} m={m} rmul {best*1e3:.2f} ms ({2*n*m*m/best/1e12:.1f} TF)")
