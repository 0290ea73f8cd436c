"""Gram pass alone at full size (best of 5): python tools/time_gram.py N M   (LFPSQP_LIB selects a variant build)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
ctx = L.Context(0, L.load_library(os.environ['LFPSQP_LIB']) if 'LFPSQP_LIB' in os.environ else None)
n, m = int(float(sys.argv[1])), int(sys.argv[2])
J = ctx.matrix(n, m).hash_fill(1)
best = 1e9
for rep in range(6):
    ctx.sync(); t = time.perf_counter(); G = L.gram(J); dt = time.perf_counter() - t
    if rep: best = min(best, dt)
print(f"n={n} m={m} gram {best*1e3:.2f} ms")
