"""One process, KZ allocations of the basis x KW allocations of the work vectors: F (the fused projected-CG kernel) timed on every pair,
in a fixed launch order so that a rocprofv3 --pmc pass over the same process can be attributed pair by pair
(tools/gpu_placement_pmc.sh groups the F dispatches by this order).   python tools/placement_pmc_probe.py KZ KW WARM"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
KZ = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KW = int(sys.argv[2]) if len(sys.argv) > 2 else 3
WARM = int(sys.argv[3]) if len(sys.argv) > 3 else 4
TOUCH, TIMED = 3, 16
ctx = L.Context(0)
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
for _ in range(WARM):
    L.projcg_(sets[0][0], None, A, L.DeviceBasis(Zs[0]), b, None, tol=1e-300, maxit=50, work=sets[0][1], n_global=n, want_lambda=False)
out = {"warm_F": WARM * 50, "per_pair_F": TOUCH + TIMED, "pairs": []}
for rnd in range(2):
    for iz, Z in enumerate(Zs):
        U = L.DeviceBasis(Z)
        for iw, (x, w) in enumerate(sets):
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=TOUCH, work=w, n_global=n, want_lambda=False)
            ctx.set_profiling(True)
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=TIMED, work=w, n_global=n, want_lambda=False)
            ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
            f = ms[3] / max(cnt[3], 1)
            out["pairs"].append({"round": rnd, "iz": iz, "iw": iw, "F_ms": f})
            print(f"round {rnd} Z{iz} W{iw}: F {f:.4f} ms", flush=True)
print("PAIRS " + json.dumps(out), flush=True)
