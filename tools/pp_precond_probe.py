"""Probe of lfpsqp_pcg_pre on config 4's stacked operator: residual after k iterations of one inner solve (exact preconditioner)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import _capi
from lfpsqp_jl_amd.inequality import InequalityData, InequalityDecomp, StackedVector, generate_initial_y_, inequality_gradient_
from lfpsqp_jl_amd.projpenalty import _JacStacked
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ctx = L.Context(0)
N, M = n + 1, m + 1
Jct = ctx.matrix(N, M).hash_fill(1, 0, n, 1.0, n, m)
i = np.arange(N)
xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
xl[n] = -np.inf; xu[n] = 0.0
idata = InequalityData(ctx, xl, xu)
xs = StackedVector(ctx, N); xs.upload(np.concatenate([0.5 * np.ones(n), [-0.1]]), 0)
generate_initial_y_(xs, idata)
# ball column of Jct: 2x on the variables, -1 on the slack row
col = np.zeros(N); col[:n] = 2 * 0.5; col[n] = -1.0
full = Jct.download(); full[:, m] = col; Jct.upload(np.asfortranarray(full))
dec = InequalityDecomp(ctx, N, M, Jct, factored=True)
inequality_gradient_(dec, xs, idata)
w = L.ProjPenaltyWork(ctx, M, N, True)
ctx.options.pp_precondition = True
w2 = L.ProjPenaltyWork(ctx, M, N, True)
Jop = _JacStacked(dec, w2); Jop.refresh()
a, b = w2.DxS.download(), w2.DyS.download()
det = mu * (a * a + b * b + mu)
i11, i12, i22 = (b * b + mu) / det, -(a * b) / det, (a * a + mu) / det
w2.i11.upload(i11); w2.i12.upload(i12); w2.i22.upload(i22)
G = L.gram(Jct, w2=w2.i11)
K = np.asfortranarray(np.linalg.inv(np.eye(M) + G))
print("cond(I + G) = %.3e, |G|max %.3e" % (np.linalg.cond(np.eye(M) + G), np.abs(G).max()))
rng = np.random.default_rng(1)
bh = rng.standard_normal(2 * N)
for maxit in (1, 2, 3, 5, 10):
    x, r = StackedVector(ctx, N), StackedVector(ctx, N); r.upload2(bh)
    pc = _capi.PcgPrecond(K.ctypes.data, w2.i11.h, w2.i12.h, w2.i22.h, w2.q.h)
    flag, iters = C.c_int(), _capi.c_i64()
    bs = Jop._basis()
    ctx.check(ctx.L.lfpsqp_pcg_pre(ctx.h, mu, C.byref(bs), C.byref(pc), x.h, r.h, w2.p.h, w2.z.h, 0.0, maxit, C.byref(flag), C.byref(iters)))
    print(f"mu={mu} maxit={maxit}: |r| = {L.nrm2(r):.3e}  (|b| = {np.linalg.norm(bh):.3e})")
