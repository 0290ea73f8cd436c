"""Fused projected-CG iteration in its STACKED form (bound-projected basis Q = InequalityDecompProject, src/inequality_helper.jl:161-212; 2N-vectors)
at full size: ms per launch of the fused kernel and per iteration.    python tools/time_stacked.py [n] [m] [--lib path]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd.inequality import InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_, inequality_gradient_
lib = None
if "--lib" in sys.argv:
    k = sys.argv.index("--lib"); lib = L.load_library(sys.argv[k + 1]); del sys.argv[k:k + 2]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = L.Context(0, lib)
i = np.arange(n)
xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
idata = InequalityData(ctx, xl, xu)
xa = StackedVector(ctx, n)
xa.upload(0.6 * np.sin(0.001 * i), 0)
generate_initial_y_(xa, idata)
out = []
for rep in range(3):                                   # three allocations of the basis: the kernel's speed depends on where it lands
    Jct = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
    dec = InequalityDecomp(ctx, n, m, Jct)
    inequality_gradient_(dec, xa, idata)
    S, Vt, rank = L.ksvd_(Jct, dec.Z, w2=dec.sx)
    dec.rank = rank
    Q = InequalityDecompProject(dec)
    A = L.DiagOperator(0.0, StackedVector(ctx, n).upload2(np.concatenate([np.full(n, 5.0), np.full(n, 4.0)])))
    b = StackedVector(ctx, n)
    b.upload2(np.cos(0.002 * np.arange(2 * n)))
    x = StackedVector(ctx, n)
    work = L.ProjCGWork(ctx, 0, m, stacked_N=n)
    L.projcg_(x, None, A, Q, b, None, tol=1e-300, maxit=5, work=work, want_lambda=False)
    ctx.set_profiling(True)
    ctx.sync(); t0 = time.perf_counter()
    it, nr = L.projcg_(x, None, A, Q, b, None, tol=1e-300, maxit=30, work=work, want_lambda=False)
    ctx.sync(); wall = (time.perf_counter() - t0) * 1e3 / max(it, 1)
    ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
    out.append((ms[3] / max(cnt[3], 1), wall, it))
print(f"stacked projcg n={n} m={m} rank={rank}: fused kernel ms per launch / call ms per iteration on three allocations: " +
      "  ".join(f"{a:.3f}/{w:.3f}" for a, w, _ in out) + f"  (iterations {out[0][2]})")
