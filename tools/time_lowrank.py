"""Iteration time of projcg! with a Hessian A = diag(a) + V diag(sigma) V' at n = 1e7, m = 128 (one MI355X): the fused ONE-pass iteration
(lfpsqp_projcg_lowrank) against the callback path with the same operator (lfpsqp_projcg_op: two passes over U per iteration), and the
diagonal operator alone.   python tools/time_lowrank.py [n] [m] [k,k,...]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
ctx = L.Context(0)
Z = ctx.matrix(n, m, placed=True).hash_fill(1)
L.orthonormalize_(Z)
work = L.ProjCGWork(ctx, n, m, against=Z, extra=1)
a = work.placed_extra[0].hash_fill(3, 0, 4.5, 5.5)
b = ctx.vector(n).hash_fill(4)
U = L.DeviceBasis(Z)
x = ctx.vector(n)
iters = 60


def run(A, **kw):
    best = 1e9
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        it, nr = L.projcg_(x, None, A, U, b, None, tol=0.0, maxit=iters, work=work, want_lambda=False, **kw)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3 / it)
    return best, it, nr


t_diag, it, nr0 = run(L.DiagOperator(0.0, a))
print(f"n={n} m={m}: diagonal operator                     {t_diag:7.3f} ms per iteration ({it} iterations, set-up included)")
for k in ks:
    V = ctx.matrix(n, k).hash_fill(17, 0, n, n ** -0.5)
    A = L.LowRankOperator(0.0, a, V, k, np.array([3.0, -0.4, 1.5, 0.7, -0.2, 2.2, 0.9, 1.1])[:k])
    t1, it1, nr1 = run(A)
    A.fused = False
    t2, it2, nr2 = run(A)
    print(f"n={n} m={m}: diag + rank-{k}: one pass (fused) {t1:7.3f} ms per iteration (incl. {k} thin set-up passes for U'V over {it1} iterations), "
          f"callback path {t2:7.3f} ms; nr {nr1:.6e} / {nr2:.6e}")
    V.free()
