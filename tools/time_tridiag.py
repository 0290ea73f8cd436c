"""Iteration time of projcg! with a TRIDIAGONAL Hessian at n = 1e7, m = 128 (one MI355X): the fused ONE-pass iteration (lfpsqp_projcg_tridiag)
against the callback path with the same operator (lfpsqp_projcg_op: two passes over U per iteration) and the diagonal operator alone; the
set-up of the reduced operator U'AU (once per solve) is separated from the iterations by timing two solve lengths.
    python tools/time_tridiag.py [n] [m] [--factored]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(float(args[0])) if len(args) > 0 else 10_000_000
m = int(args[1]) if len(args) > 1 else 128
factored = "--factored" in sys.argv
ctx = L.Context(0)
if factored:
    J = ctx.matrix(n, m, placed=True).hash_fill(1)
    W = np.zeros((m, m), order="F")
    S, Vt, rank = L.ksvd_(J, None, W=W)
    U = L.DeviceBasis(None, rank, generator=(J, W))
    work = L.ProjCGWork(ctx, n, m, against=J, extra=1)
else:
    Z = ctx.matrix(n, m, placed=True).hash_fill(1)
    L.orthonormalize_(Z)
    U = L.DeviceBasis(Z)
    work = L.ProjCGWork(ctx, n, m, against=Z, extra=1)
a = work.placed_extra[0].hash_fill(3, 0, 4.5, 6.5)            # 2 .. 11
off = ctx.vector(n).hash_fill(15, 0, 0.8, 0.0)                 # couplings of both signs, |off| <= 0.8: diagonally dominant (a_i - |off_i| - |off_i-1| >= 0.4)
offn = ctx.vector(n).hash_fill(15, 0, 3.0, 0.0)                # ... and not (a negative Gram weight part: a third set-up pass), with
an = ctx.vector(n).hash_fill(3, 0, 4.5, 10.0)                  # the diagonal 5.5 .. 14.5 (still positive definite)
b = ctx.vector(n).hash_fill(4)
x = ctx.vector(n)


def run(A, iters, **kw):
    best = 1e9
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        it, nr = L.projcg_(x, None, A, U, b, None, tol=0.0, maxit=iters, work=work, want_lambda=False, **kw)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, it, nr


def per_iteration(A, **kw):
    t1, i1, _ = run(A, 10, **kw)            # (both lengths end before the residual reaches rounding level: no early exit)
    t2, i2, nr = run(A, 40, **kw)
    per = (t2 - t1) / (i2 - i1)
    return per, t1 - per * i1, nr


tag = "factored basis U = J W" if factored else "materialised basis"
p0, s0, nr0 = per_iteration(L.DiagOperator(0.0, a))
print(f"n={n} m={m} ({tag}): diagonal operator            {p0:7.3f} ms per iteration, {s0:6.2f} ms per solve outside the iterations")
for name, dgv, o in (("diagonally dominant", a, off), ("not dominant (third Gram pass)", an, offn)):
    A = L.TridiagonalOperator(0.0, dgv, o)
    p1, s1, nr1 = per_iteration(A)
    line = f"n={n} m={m} ({tag}): tridiagonal, {name}: one pass {p1:7.3f} ms per iteration, {s1:6.2f} ms per solve outside the iterations (U'AU: two or three Gram passes)"
    if not factored:
        A.fused = False
        p2, s2, nr2 = per_iteration(A)
        line += f"; callback path {p2:7.3f} ms per iteration; nr {nr1:.6e} / {nr2:.6e}"
    print(line)
