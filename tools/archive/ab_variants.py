"""Interleaved A/B of the streaming-kernel tuning variants on the SAME buffers in one process."""
import sys, math, statistics as st; sys.path.insert(0, '.')
import lfpsqp_jl_amd as L
n, m = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = L.Context(0)
Z = ctx.matrix(n, m).hash_fill(1, 0, n, 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n))))
A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)); b = ctx.vector(n).hash_fill(4)
x = ctx.vector(n); work = L.ProjCGWork(ctx, n, m); vv = ctx.vector(n).hash_fill(5); tt = ctx.vector(m).hash_fill(6); y = ctx.vector(n)
U = L.DeviceBasis(Z)
variants = [(2, False), (2, True), (4, False), (4, True)]
R = {v: dict(gt=[], gn=[], k1=[], k2=[], k3=[], it=[]) for v in variants}
for rnd in range(9):
    for v in (variants if rnd % 2 == 0 else variants[::-1]):
        ctx.set_tuning(*v); r = R[v]
        L.gemv_t(Z, vv, tt)
        ctx.timer_begin(); [L.gemv_t(Z, vv, tt) for _ in range(4)]; r['gt'].append(ctx.timer_end() / 4)
        ctx.timer_begin(); [L.gemv_n(Z, tt, y, 1.0, 1.0) for _ in range(4)]; r['gn'].append(ctx.timer_end() / 4)
        ctx.set_profiling(True)
        ctx.timer_begin(); L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=12, work=work, want_lambda=False); ms = ctx.timer_end()
        pm, pc = ctx.profile_read(); ctx.set_profiling(False)
        r['it'].append(ms / 12)
        for k, slot in (('k1', 0), ('k2', 1), ('k3', 2)):
            r[k].append(pm[slot] / pc[slot])
print(f"n={n} m={m}  median/min ms")
print(f"{'(ks,nt)':12s} " + " ".join(f"{k:>14s}" for k in ('gemv_t', 'gemv_n', 'K1', 'K2', 'K3', 'iter')))
for v in variants:
    r = R[v]
    print(f"{str(v):12s} " + " ".join(f"{st.median(r[k]):7.3f}/{min(r[k]):6.3f}" for k in ('gt', 'gn', 'k1', 'k2', 'k3', 'it')))
