"""Does the library's placement trial (the fused kernel on ZERO-filled buffers) rank allocation pairs like the real solve does?
For KZ matrices x KW vector sets: the trial's ms on zeros, then (after filling everything with the bench's data) the fused kernel's ms in a
real projcg run on the same pair, then the trial's arithmetic again on the now non-zero matrix with fresh zero vectors is not possible
(the vectors hold data), so: zeros-before vs real-after, per pair.      python tools/placement_probe_check.py [KZ] [KW]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
KZ = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KW = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = L.Context(0)
ctx.set_placement(1)
n, m = 10_000_000, 128
scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n)))
Zs = [ctx.matrix(n, m) for _ in range(KZ)]
sets = [ctx.vectors_placed(None, n, 5) for _ in range(KW)]          # g, d, a, rp, x from one slab each
b = ctx.vector(n).hash_fill(4)


def probe(Z, s, reps=3):
    ms = C.c_double()
    ctx.check(ctx.L.lfpsqp_placement_probe(ctx.h, Z.h, m, s[0].h, s[1].h, s[2].h, reps, C.byref(ms)))
    return ms.value


for _ in range(150):
    probe(Zs[0], sets[0])                        # warm the device (~0.8 s)
zero_ms = {}
for rnd in range(2):
    for iz, Z in enumerate(Zs):
        for iw, s in enumerate(sets):
            t = probe(Z, s)
            zero_ms[(iz, iw)] = min(t, zero_ms.get((iz, iw), 1e9))
# matrix still zero, vectors zero: zero matrix but REAL vector data?  fill the matrices only, probe again (vectors still zero)
for Z in Zs:
    Z.hash_fill(1, 0, n, scale)
zvec_ms = {(iz, iw): probe(Z, s) for iz, Z in enumerate(Zs) for iw, s in enumerate(sets)}
real_ms = {}
for iz, Z in enumerate(Zs):
    U = L.DeviceBasis(Z)
    for iw, s in enumerate(sets):
        w = L.ProjCGWork.__new__(L.ProjCGWork)
        w.g, w.d, w.rp, w.Utr, w.Av, w._extra = s[0], s[1], s[3], ctx.vector(m), None, None
        s[2].hash_fill(3, 0, 4.0, 5.0)
        A = L.DiagOperator(0.0, s[2])
        L.projcg_(s[4], None, A, U, b, None, tol=1e-300, maxit=3, work=w, n_global=n, want_lambda=False)
        ctx.set_profiling(True)
        L.projcg_(s[4], None, A, U, b, None, tol=1e-300, maxit=16, work=w, n_global=n, want_lambda=False)
        ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
        real_ms[(iz, iw)] = ms[3] / max(cnt[3], 1)
        for v in s[:5]:
            v.fill(0.0)
print("pair    zeros(all)  zeros(vectors only)  real")
for k in sorted(zero_ms):
    print(f"Z{k[0]} W{k[1]}   {zero_ms[k]:.4f}      {zvec_ms[k]:.4f}              {real_ms[k]:.4f}")
