"""Several 10 GB bases allocated side by side in one process: does a plain streaming read (gemv_t) predict which of them the
fused projected-CG iteration runs fastest on?  (Background for placement probing in lfpsqp_mat_create.)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
m = 128; n = 10_000_000; K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = L.Context(0)
A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)); b = ctx.vector(n).hash_fill(4, 0)
x = ctx.vector(n); work = L.ProjCGWork(ctx, n, m); t = ctx.vector(m)
def cg(Z, its=30):
    U = L.DeviceBasis(Z)
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=work, n_global=n, want_lambda=False)
    ctx.sync(); t0 = time.perf_counter()
    it, nr = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=its, work=work, n_global=n, want_lambda=False)
    ctx.sync(); return (time.perf_counter() - t0) / it * 1e3
def rd(Z, its=5):
    L.gemv_t(Z, b, t); ctx.sync(); t0 = time.perf_counter()
    for _ in range(its): L.gemv_t(Z, b, t)
    ctx.sync(); return (time.perf_counter() - t0) / its * 1e3
Zs = [ctx.matrix(n, m).hash_fill(1, 0, n, 2.0 ** -11) for _ in range(K)]
for rep in range(2):
    for i, Z in enumerate(Zs):
        print(f"rep {rep} basis {i}: gemv_t {rd(Z):.4f} ms   projcg {cg(Z):.4f} ms/iter", flush=True)
