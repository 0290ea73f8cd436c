"""EXPERIMENT: is the speed of the fused projected-CG kernel F a function of WHERE INSIDE one allocation its n-vectors start, at steps of
(ARCHIVED: the -DLFPSQP_VMM_EXPERIMENT variant of csrc/context.hip this probe needs was removed from the sources in round 5; git show a86abc9:lfpsqp.jl_amd/csrc/context.hip has it)
megabytes (steps up to 1 MB were tried in round 2: no)?  One basis, one set of over-sized vectors (x, g, d, rp, the diagonal of A), F timed with the
vectors' starts moved by s MB -- all together, then one at a time.  Needs the variant library built with -DLFPSQP_VMM_EXPERIMENT
(tools/gpu_vmm_probe.sh):   LFPSQP_LIB=lfpsqp.jl_amd/lib/variants/liblfpsqp_vmm.so python tools/skew_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import math
import lfpsqp_jl_amd as L
lib = L.load_library(os.environ["LFPSQP_LIB"])
lib.lib.lfpsqp_x_vec_skew.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
ctx = L.Context(0, lib)
n, m = 10_000_000, 128
MB = int(os.environ.get('SKEW_UNIT_DOUBLES', 1 << 17))      # doubles per step: 1 MB by default; 16 = one 128-byte line
SLACK = 160 * (1 << 17)
scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n)))
Z = ctx.matrix(n, m).hash_fill(1, 0, n, scale)
U = L.DeviceBasis(Z)
b = ctx.vector(n).hash_fill(4)
names = ["x", "g", "d", "rp", "a"]
vec = {k: ctx.vector(n + SLACK) for k in names}
work = L.ProjCGWork(ctx, 8, m)
for k in ("g", "d", "rp"):
    setattr(work, k, vec[k])
A = L.DiagOperator(0.0, vec["a"])


def place(skews):
    for k in names:
        lib.lib.lfpsqp_x_vec_skew(vec[k].h, int(skews.get(k, 0)) * MB, n)
        vec[k].n = n
    vec["a"].hash_fill(3, 0, 4.0, 5.0)


def timeF():
    x = vec["x"]
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=work, n_global=n, want_lambda=False)
    ctx.set_profiling(True)
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=16, work=work, n_global=n, want_lambda=False)
    ms, cnt = ctx.profile_read(); ctx.set_profiling(False)
    return ms[3] / max(cnt[3], 1)


place({})
for _ in range(20):
    L.projcg_(vec["x"], None, A, U, b, None, tol=1e-300, maxit=50, work=work, n_global=n, want_lambda=False)
S = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 0]
print("all vectors moved together:   " + "  ".join(f"{s}MB {timeF() if place({k: s for k in names}) is None else 0:.3f}" for s in S), flush=True)
for who in names:
    print(f"only {who:2s} moved:               " + "  ".join(f"{s}MB {timeF() if place({who: s}) is None else 0:.3f}" for s in S), flush=True)
# staggered: every vector its own offset
for base in (1, 2, 5, 16):
    sk = {k: base * i for i, k in enumerate(names)}
    place(sk)
    print(f"staggered by {base} MB ({sk}): {timeF():.3f}", flush=True)
