"""Random shapes through the tangent-setup kernels: Gram (plain / weighted / leading columns), rmul (all three kernels) and
lfpsqp_factorize against numpy on the same inputs.   python tools/fuzz_factorize.py [cases] [seed]   (GPU, or LFPSQP_LIB=<emulator .so>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = L.load_library(os.environ["LFPSQP_LIB"]) if "LFPSQP_LIB" in os.environ else None
ctx = L.Context(0, lib)
emu = "emulator" in ctx.device_name
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    m = int(rng.choice([rng.integers(1, 20), rng.integers(20, 140), rng.integers(125, 135), rng.integers(140, 300 if emu else 640)]))
    n = int(rng.integers(max(m, 1), 3000 if emu else 60000))
    Mh = np.asfortranarray(rng.standard_normal((n, m)))
    wh = rng.random(n) + 0.05
    M, w = ctx.matrix(n, m, Mh), ctx.vector(n, wh)
    errs = {}
    G0 = Mh.T @ Mh
    errs["gram"] = np.abs(L.gram(M) - G0).max() / np.abs(G0).max()
    Gw = Mh.T @ (wh[:, None] * Mh)
    errs["gram_w"] = np.abs(L.gram(M, w2=w) - Gw).max() / np.abs(Gw).max()
    nc = int(rng.integers(1, m + 1))
    errs["gram_lead"] = np.abs(L.gram(M, ncols=nc) - G0[:nc, :nc]).max() / np.abs(G0).max()
    r = int(rng.integers(1, m + 1))
    W = rng.standard_normal((m, r))
    O = ctx.matrix(n, m)
    L.rmul(M, W, O)
    ref = Mh @ W
    errs["rmul"] = np.abs(O.download()[:, :r] - ref).max() / np.abs(ref).max()
    Z = ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(M, Z)
    S0 = np.linalg.svd(Mh, compute_uv=False)
    Zh = Z.download()
    errs["sigma"] = np.abs(S - S0).max() / S0[0]
    errs["orth"] = np.abs(Zh.T @ Zh - np.eye(m)).max()
    errs["recon"] = np.abs((Zh * S) @ Vt - Mh).max() / S0[0]
    worst = max(errs.values())
    ok = rank == m and worst <= 5e-12
    bad += not ok
    if not ok or case % 10 == 0:
        print(f"case {case}: n={n} m={m} rank={rank} " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()) + ("" if ok else "   <-- MISMATCH"))
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
