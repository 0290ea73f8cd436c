"""Two builds of the library (tools/build_variant.py) timed on the SAME device buffers in one process: the speed of the fused
projected-CG kernel depends on the (matrix, work-vector) allocation pair (FINDINGS.md 6), so variants can only be compared on identical
buffers.  Handles are plain structs with the same layout in both builds: the buffers are allocated through build A and handed to build B.
    python tools/ab_same_buffers.py <libA | main> <libB> [KZ] [KW]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import _capi
import ctypes as C


def lib_of(tag):
    return None if tag == "main" else L.load_library(os.path.join(os.path.dirname(_capi.DEFAULT_LIB), "variants", f"liblfpsqp_{tag}.so"))


tagA, tagB = sys.argv[1], sys.argv[2]
KZ = int(sys.argv[3]) if len(sys.argv) > 3 else 3
KW = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = int(float(os.environ.get("AB_ROWS", "1e7"))); m = int(os.environ.get("AB_COLS", "128"))
ctxA, ctxB = L.Context(0, lib_of(tagA)), L.Context(0, lib_of(tagB))
if os.environ.get("AB_PING_B"):           # build B with the alternating residual buffers (lfpsqp_ctx_set_residual_buffers)
    ctxB.set_residual_buffers(1)
    tagB = tagB + "+alternating"
scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n)))
Zs, pads = [], []
for k in range(KZ):
    Zs.append(ctxA.matrix(n, m).hash_fill(1, 0, n, scale))
    pads.append(ctxA.vector(3_000_017 * (k + 1)))
dg = ctxA.vector(n).hash_fill(3, 0, 4.0, 5.0)
b = ctxA.vector(n).hash_fill(4)
sets = []
for k in range(KW):
    sets.append((ctxA.vector(n), L.ProjCGWork(ctxA, n, m)))
    pads.append(ctxA.vector(1_000_003 * (k + 1)))
ctxA.sync()


def run(ctx, Z, x, w, maxit):
    """lfpsqp_projcg through `ctx`'s build on buffers owned by build A."""
    A = _capi.DiagOp(0.0, dg.h)
    U = _capi.Basis(Z.h, m, None, None, None, None)
    wk = _capi.ProjCGWorkC(w.g.h, w.d.h, w.rp.h, w.Utr.h)
    it, nr = _capi.c_i64(), C.c_double()
    ctx.check(ctx.L.lfpsqp_projcg(ctx.h, x.h, None, C.byref(A), C.byref(U), b.h, None, 1e-300, maxit, n, 0, C.byref(wk), C.byref(it), C.byref(nr)))


for ctx in (ctxA, ctxB):
    for _ in range(6):
        run(ctx, Zs[0], sets[0][0], sets[0][1], 50)
res, k1 = {}, {}
for rnd in range(3):
    for iz, Z in enumerate(Zs):
        for iw, (x, w) in enumerate(sets):
            for tag, ctx in ((tagA, ctxA), (tagB, ctxB)) if rnd % 2 == 0 else ((tagB, ctxB), (tagA, ctxA)):
                run(ctx, Z, x, w, 3)
                ctx.set_profiling(True)
                run(ctx, Z, x, w, 16)
                ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
                res.setdefault((iz, iw), {}).setdefault(tag, []).append(ms[3] / max(cnt[3], 1))
                k1.setdefault(tag, []).append(ms[0] / max(cnt[0], 1))
print(f"F ms per (basis, work set) allocation pair, n={n} m={m}: {tagA} | {tagB}   (min of 3 rounds)")
tot = {tagA: 0.0, tagB: 0.0}
for (iz, iw), r in sorted(res.items()):
    a, bb = min(r[tagA]), min(r[tagB])
    tot[tagA] += a; tot[tagB] += bb
    print(f"  Z{iz} W{iw}: {a:.4f} | {bb:.4f}   ({(bb / a - 1) * 100:+.1f} %)")
print(f"  mean: {tot[tagA] / len(res):.4f} | {tot[tagB] / len(res):.4f}")
print(f"  K1 (x += alpha d; d = beta d - g) mean ms: {sum(k1[tagA]) / len(k1[tagA]):.4f} | {sum(k1[tagB]) / len(k1[tagB]):.4f}")
