"""Which of the fused kernel's n-vectors decides its speed on a given basis allocation?  One vector role at a time is swapped through NC
candidate allocations while the others stay fixed: F ms per candidate.   python tools/which_vector_probe.py [NC]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfpsqp_jl_amd as L
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = L.Context(0)
n, m = 10_000_000, 128
scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n)))
Zs = [ctx.matrix(n, m).hash_fill(1, 0, n, scale) for _ in range(2)]
b = ctx.vector(n).hash_fill(4)
x = ctx.vector(n)
base = L.ProjCGWork(ctx, n, m)
a0 = ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)
cands, pads = [], []
for k in range(NC):
    cands.append(ctx.vector(n))
    pads.append(ctx.vector(1_000_003 * (k + 1)))


def F(Z, work, a):
    U, A = L.DeviceBasis(Z), L.DiagOperator(0.0, a)
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=work, n_global=n, want_lambda=False)
    ctx.set_profiling(True)
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=16, work=work, n_global=n, want_lambda=False)
    ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
    return ms[3] / max(cnt[3], 1)


for _ in range(10):
    F(Zs[0], base, a0)
for iz, Z in enumerate(Zs):
    print(f"Z{iz}: base set {F(Z, base, a0):.4f}")
    for role in ("g", "d", "a"):
        row = []
        for c in cands:
            w = L.ProjCGWork.__new__(L.ProjCGWork)
            w.g, w.d, w.rp, w.Utr, w.Av, w._extra = base.g, base.d, base.rp, base.Utr, None, None
            a = a0
            if role == "a":
                c.copy_from(a0); a = c
            else:
                setattr(w, role, c)
            row.append(F(Z, w, a))
        print(f"   role {role}: " + "  ".join(f"{v:.4f}" for v in row), flush=True)
