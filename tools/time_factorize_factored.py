"""Phase times of the tangent setup with the basis left in factored form (LFPSQP_TRACE_FACTORIZE=1 prints the library's own phases)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
ctx = L.Context(0)
n, m = int(float(sys.argv[1])), int(sys.argv[2])
J = ctx.matrix(n, m).hash_fill(1)
W = np.zeros((m, m), order='F')
for rep in range(5):
    ctx.sync(); t = time.perf_counter(); S, Vt, r = L.ksvd_(J, None, W=W); ctx.sync(); print(f"factorize (factored basis) {1e3 * (time.perf_counter() - t):.3f} ms rank {r}", file=sys.stderr)
    ctx.sync(); t = time.perf_counter(); G = L.gram(J); print(f"   gram call alone {1e3 * (time.perf_counter() - t):.3f} ms", file=sys.stderr)
