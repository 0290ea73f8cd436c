"""Does the fused projected-CG iteration time depend on the leading dimension (column stride -> HBM channel mapping) or on
which allocation the basis landed in?  One process; n = 1e7 + k*2048 rows, two independently allocated bases per n."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
m = 128
ctx = L.Context(0)
def timeit(U, A, b, x, work, n, its=30):
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=work, n_global=n, want_lambda=False)
    ctx.sync(); t0 = time.perf_counter()
    it, nr = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=its, work=work, n_global=n, want_lambda=False)
    ctx.sync(); return (time.perf_counter() - t0) / it * 1e3
for k in (0, 1, 2, 3, 5, 8, 13, 16, 32):
    n = 10_000_000 + 2048 * k
    scale = 2.0 ** -11
    Z1 = ctx.matrix(n, m).hash_fill(1, 0, n, scale)
    Z2 = ctx.matrix(n, m).hash_fill(1, 0, n, scale)
    A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)); b = ctx.vector(n).hash_fill(4, 0)
    x = ctx.vector(n); work = L.ProjCGWork(ctx, n, m)
    t = [timeit(L.DeviceBasis(Z), A, b, x, work, n) for Z in (Z1, Z2, Z1, Z2)]
    print(f"k={k:2d} ld={n + (-n) % 2048}  ms/iter Z1 {t[0]:.4f} Z2 {t[1]:.4f} Z1 {t[2]:.4f} Z2 {t[3]:.4f}", flush=True)
    for o in (Z1, Z2, b, x): o.free()
