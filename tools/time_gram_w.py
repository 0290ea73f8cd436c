"""Weighted against plain Gram kernel at (1e7, 128), kernel times from the profiling-free wall clock of lfpsqp_gram (best of 8), optionally on a
variant build:   python tools/time_gram_w.py [--lib variant.so]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
libpath = None
if '--lib' in sys.argv:
    k = sys.argv.index('--lib'); libpath = sys.argv[k + 1]; del sys.argv[k:k + 2]
n, m = 10_000_000, 128
ctx = L.Context(0, L.load_library(libpath) if libpath else None)
A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
w2 = ctx.vector(n).hash_fill(5, 0, 0.4, 0.6)
e1 = ctx.vector(n).hash_fill(8, 0, 1.0, 0.0)


def timed(fn, reps=8):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


print(f"{os.path.basename(libpath) if libpath else 'default':28s} gram {timed(lambda: L.gram(A)):7.3f} ms   weighted {timed(lambda: L.gram(A, w2=w2)):7.3f} ms   "
      f"gram + 1 column {timed(lambda: L.gram_rhs(A, [e1])):7.3f}   weighted + 1 column {timed(lambda: L.gram_rhs(A, [e1], w2=w2)):7.3f} ms")
