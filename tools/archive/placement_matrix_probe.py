"""F (fused projected-CG kernel) over KZ allocations of the basis x KW allocations of the work vectors, one process:
which of the two placements decides its speed?   python tools/placement_matrix_probe.py [KZ] [KW]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import lfpsqp_jl_amd as L
KZ = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KW = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = L.Context(0, L.load_library(os.environ['LFPSQP_LIB']) if 'LFPSQP_LIB' in os.environ else None)      # (a variant build)
n, m = 10_000_000, 128
scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n)))
Zs, pads = [], []
for k in range(KZ):
    Zs.append(ctx.matrix(n, m).hash_fill(1, 0, n, scale))
    pads.append(ctx.vector(3_000_017 * (k + 1)))
A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0))
b = ctx.vector(n).hash_fill(4)
sets = []
for k in range(KW):
    sets.append((ctx.vector(n), L.ProjCGWork(ctx, n, m)))
    pads.append(ctx.vector(1_000_003 * (k + 1)))
for _ in range(20):                                   # warm the device
    L.projcg_(sets[0][0], None, A, L.DeviceBasis(Zs[0]), b, None, tol=1e-300, maxit=50, work=sets[0][1], n_global=n, want_lambda=False)
for rnd in range(2):
    for iz, Z in enumerate(Zs):
        U = L.DeviceBasis(Z)
        row = []
        for x, w in sets:
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=w, n_global=n, want_lambda=False)
            ctx.set_profiling(True)
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=16, work=w, n_global=n, want_lambda=False)
            ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
            row.append(ms[3] / max(cnt[3], 1))
        # does the plain GEMV-N (a store stream inside the matrix stream as well) rank the allocations of the matrix like F does?
        tt, yy = ctx.vector(m).hash_fill(6), sets[0][1].rp
        L.gemv_n(Z, tt, yy, 1.0, 1.0)
        ctx.timer_begin()
        for _ in range(4):
            L.gemv_n(Z, tt, yy, 1.0, 1.0)
        gn = ctx.timer_end() / 4
        ctx.timer_begin()
        for _ in range(4):
            L.gemv_t(Z, yy, tt)
        gt = ctx.timer_end() / 4
        print(f"round {rnd} Z{iz}: F ms over the work sets: " + "  ".join(f"{v:.3f}" for v in row) + f"   gemv_n {gn:.3f}  gemv_t {gt:.3f}", flush=True)
