"""Random shapes through round 5's entry points: the one-pass projected CG with a tridiagonal and with a diagonal + low-rank Hessian
(lfpsqp_projcg_tridiag / _lowrank) against the callback path with the same operator (lfpsqp_projcg_op), materialised and factored bases;
lfpsqp_tridiag_mul and the Gram pass with extra right-hand columns (lfpsqp_gram_rhs) against numpy.
    python tools/fuzz_operators.py [cases] [seed]   (GPU, or LFPSQP_LIB=<emulator .so>)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
lib = L.load_library(os.environ["LFPSQP_LIB"]) if "LFPSQP_LIB" in os.environ else None
ctx = L.Context(0, lib)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = lib is None
bad = 0


def rel(u, v):
    return np.linalg.norm(u - v) / max(np.linalg.norm(v), 1e-300)


for case in range(cases):
    m = int(rng.choice([4, 5, 7, 8, 31, 32, 33, 64, 96, 127, 128, 129, 131, 132, 133, 200, 256, 257, 300, 512, 600][: (21 if big else 12)]))
    n = int(rng.integers(max(m + 3, 20), 120_000 if big else 2500))
    if rng.random() < 0.2:
        n = int(-(-n // 2048) * 2048) if big else int(-(-n // 64) * 64)           # (tile / padding multiples: the neighbour loads at the ends)
    J = ctx.matrix(n, m).hash_fill(1 + case, 0, n, 1.0)
    Z = ctx.matrix(n, m); W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(J, Z, W=W)
    factored = bool(rng.random() < 0.4)
    U = L.DeviceBasis(None, m, generator=(J, W)) if factored else L.DeviceBasis(Z)
    Um = L.DeviceBasis(Z)                                                          # (the callback path needs the materialised basis)
    b = ctx.vector(n).hash_fill(4 + case, 0)
    dominant = bool(rng.random() < 0.6)
    a = ctx.vector(n).hash_fill(3 + case, 0, 4.0, 6.0 if dominant else 10.5)       # 2 .. 10, or 6.5 .. 14.5
    off = ctx.vector(n).hash_fill(15 + case, 0, 0.9 if dominant else 3.0, float(rng.choice([0.0, 0.0, -0.3])))
    ok = True
    # --- the operator on its own
    vh = np.asarray(rng.standard_normal(n))
    ah, oh = a.download(), off.download()
    ref = ah * vh
    ref[:-1] += oh[:-1] * vh[1:]
    ref[1:] += oh[:-1] * vh[:-1]
    T = L.TridiagonalOperator(0.0, a, off)
    out = ctx.vector(n)
    T.mul_(out, ctx.vector(n, vh))
    ok = ok and np.abs(out.download() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
    # --- tridiagonal: one pass against the callback path
    res = {}
    for fused in (True, False):
        T.fused = fused
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, T, U if fused else Um, b, None, tol=1e-9, maxit=30)
        res[fused] = (it, nr, x.download(), lam.download())
    A1, A0 = res[True], res[False]
    ok_t = A1[0] == A0[0] and rel(A1[2], A0[2]) < 1e-9 and (np.isinf(A0[1]) or np.abs(A1[3] - A0[3]).max() < 1e-9 * max(1.0, np.abs(A0[3]).max()))
    # --- diagonal + low rank
    k = int(rng.integers(1, 9))
    V = ctx.matrix(n, k).hash_fill(17 + case, 0, n, n ** -0.5)
    sig = rng.uniform(-0.5, 3.0, size=k)
    LR = L.LowRankOperator(0.0, a, V, k, sig)
    res = {}
    for fused in (True, False):
        LR.fused = fused
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, LR, U if fused else Um, b, None, tol=1e-9, maxit=30)
        res[fused] = (it, nr, x.download(), lam.download())
    B1, B0 = res[True], res[False]
    ok_l = B1[0] == B0[0] and rel(B1[2], B0[2]) < 1e-9 and (np.isinf(B0[1]) or np.abs(B1[3] - B0[3]).max() < 1e-9 * max(1.0, np.abs(B0[3]).max()))
    # --- Gram pass with extra right-hand columns
    weighted = bool(rng.random() < 0.5)
    w2 = ctx.vector(n).hash_fill(9 + case, 0, 0.5, 0.6) if weighted else None
    nx = int(rng.integers(1, 3))
    es = [ctx.vector(n).hash_fill(21 + case + q, 0) for q in range(nx)]
    G, X = L.gram_rhs(J, es, m, w2)
    Jh = J.download()
    wh = w2.download() if weighted else np.ones(n)
    Gr = Jh.T @ (wh[:, None] * Jh)
    Xr = np.stack([Jh.T @ (np.sqrt(wh) * e.download()) for e in es], axis=1)
    ok_g = np.abs(G - Gr).max() <= 1e-11 * np.abs(Gr).max() and np.abs(np.asarray(X) - Xr).max() <= 1e-11 * max(1.0, np.abs(Xr).max())
    if not (ok and ok_t and ok_l and ok_g):
        bad += 1
        print(f"MISMATCH case {case}: n={n} m={m} factored={factored} dominant={dominant} k={k} weighted={weighted} nx={nx}: mul {ok}, tridiagonal {ok_t} "
              f"(its {A1[0]}/{A0[0]}, dx {rel(A1[2], A0[2]):.1e}), low rank {ok_l} (its {B1[0]}/{B0[0]}, dx {rel(B1[2], B0[2]):.1e}), gram {ok_g}")
    for v in (J, Z, V):
        v.free()
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
