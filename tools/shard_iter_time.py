"""Per-iteration time of the fused projected-CG loop at the per-GPU shard sizes of an N-GPU strong-scaling run of n = 1e7
(n/N rows on ONE GPU), without a communicator and with a 1-rank RCCL communicator (the stream-ordered ncclAllReduce call of
every iteration is then in the loop: its launch cost, not its wire time)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
m = 128
def run(n, rccl, its=200):
    ctx = L.Context(0)
    if rccl: ctx.comm_init_rccl(0, 1, ctx.comm_unique_id())
    Z = ctx.matrix(n, m).hash_fill(1, 0, n, 2.0 ** -11)
    A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)); b = ctx.vector(n).hash_fill(4, 0)
    x = ctx.vector(n); work = L.ProjCGWork(ctx, n, m); U = L.DeviceBasis(Z)
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=5, work=work, n_global=n, want_lambda=False)
    best = 1e9
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        it, nr = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=its, work=work, n_global=n, want_lambda=False)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / it * 1e3)
    ctx.close()
    return best
for N in (1, 2, 4, 8):
    n = 10_000_000 // N
    a, b_ = run(n, False, 60 if N == 1 else 200), run(n, True, 60 if N == 1 else 200)
    print(f"N={N}: rows {n}: {a:.4f} ms/iter without communicator, {b_:.4f} ms/iter with 1-rank RCCL  (ideal {1.0 / a:.2f} k it/s)", flush=True)
