"""Gram matrix / tangent setup timings at n = 1e7 (one MI355X): plain, weighted (bounds), over a view (streamed gradients).
    python tools/time_gram.py [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

n, m = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = L.Context(0)
A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
w2 = ctx.vector(n).hash_fill(5, 0, 0.4, 0.6)              # in (0.2, 1.0)
rs = ctx.vector(n).hash_fill(6, 0, 1.0, 0.0)
u = ctx.vector(n).hash_fill(7, 0, 1.0, 0.0)
w = ctx.vector(m, 1e-3 * np.cos(np.arange(m)))
W = np.zeros((m, m), order='F')


def timed(fn, reps=6):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


S0, Vt0, r0 = L.ksvd_(A, None, W=W)
print(f"tangent setup (factored), plain: cold {timed(lambda: L.ksvd_(A, None, W=W)):7.3f} ms   warm (lfpsqp_factorize_hint) {timed(lambda: L.ksvd_(A, None, W=W, Vt_prev=Vt0)):7.3f} ms"
      f"   with weights: cold {timed(lambda: L.ksvd_(A, None, w2=w2, W=W)):7.3f}   warm from the unweighted Vt {timed(lambda: L.ksvd_(A, None, w2=w2, W=W, Vt_prev=Vt0)):7.3f} ms")
views = {"plain": A, "row scales": A.view(rs), "row scales + rank one": A.view(rs, u, w)}
for name, M in views.items():
    print(f"{name:24s} gram {timed(lambda: L.gram(M)):7.3f} ms   weighted {timed(lambda: L.gram(M, w2=w2)):7.3f} ms   "
          f"tangent setup (factored) {timed(lambda: L.ksvd_(M, None, W=W)):7.3f} ms   with weights {timed(lambda: L.ksvd_(M, None, w2=w2, W=W)):7.3f} ms")
# extra right-hand columns riding with the Gram pass (lfpsqp_gram_rhs): the outer iteration's Jct'd, and the view's own rank-one term
e1, e2 = ctx.vector(n).hash_fill(8, 0, 1.0, 0.0), ctx.vector(n).hash_fill(9, 0, 1.0, 0.0)
for name, M in views.items():
    print(f"{name:24s} gram + 1 column {timed(lambda: L.gram_rhs(M, [e1])):7.3f} ms   + 2 columns {timed(lambda: L.gram_rhs(M, [e1, e2])):7.3f} ms   "
          f"weighted + 1 column {timed(lambda: L.gram_rhs(M, [e1], w2=w2)):7.3f} ms   tangent setup with Jct'd {timed(lambda: L.ksvd_(M, None, W=W, rhs=e1)):7.3f} ms")

