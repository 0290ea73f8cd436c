"""How are the fast allocation pairs of the fused kernel distributed?  KZ matrices x KW vector slabs in one process, the library's own
placement trial (the fused kernel on zero-filled buffers) on every pair, printed as a KZ x KW table (ms).    python tools/placement_pairs_probe.py [KZ] [KW]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
KZ = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KW = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SPACER_GB = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # a buffer of this size allocated between the matrices and the vector slabs
PAD_GB = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0         # ... and one of this size between consecutive slabs
ctx = L.Context(0)
ctx.set_placement(1)
n, m = 10_000_000, 128
Zs = [ctx.matrix(n, m) for _ in range(KZ)]
spacer = ctx.vector(int(SPACER_GB * 1e9 / 8)) if SPACER_GB > 0 else None
sets, pads = [], []
for _ in range(KW):
    sets.append(ctx.vectors_placed(None, n, 5))
    if PAD_GB > 0:
        pads.append(ctx.vector(int(PAD_GB * 1e9 / 8)))


def probe(Z, s, reps=3):
    ms = C.c_double()
    ctx.check(ctx.L.lfpsqp_placement_probe(ctx.h, Z.h, m, s[0].h, s[1].h, s[2].h, reps, C.byref(ms)))
    return ms.value


for _ in range(200):
    probe(Zs[0], sets[0])
best = {}
for rnd in range(2):
    for iz, Z in enumerate(Zs):
        for iw, s in enumerate(sets):
            t = probe(Z, s)
            best[(iz, iw)] = min(t, best.get((iz, iw), 1e9))
lo = min(best.values())
for iz in range(KZ):
    print(f"Z{iz}: " + "  ".join(f"{best[(iz, iw)]:.3f}{'*' if best[(iz, iw)] < 1.012 * lo and lo < 1.61 else ' '}" for iw in range(KW)))
print(f"spacer {SPACER_GB} GB; " if SPACER_GB else "", end="")
print(f"pads of {PAD_GB} GB between slabs; " if PAD_GB else "", end="")
print(f"fast pairs (within 1.2 % of the best, best < 1.61 ms): {sum(1 for v in best.values() if v < 1.012 * lo and lo < 1.61)} of {KZ * KW}")
