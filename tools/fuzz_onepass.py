"""Random shapes: the one-pass kernels (projcg, pcg!, Newton step incl. the batched form) against their two-pass forms on the
same inputs.  python tools/fuzz_onepass.py [cases] [seed]   (GPU, or LFPSQP_LIB=<emulator .so>)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd.projpenalty import _JacPlain
lib = L.load_library(os.environ["LFPSQP_LIB"]) if "LFPSQP_LIB" in os.environ else None
ctx = L.Context(0, lib)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = lib is None
bad = 0
for case in range(cases):
    m = int(rng.choice([4, 5, 7, 8, 31, 32, 33, 64, 96, 127, 128, 129, 131, 132, 133, 200, 256, 257, 300, 512, 600][: (21 if big else 12)]))
    n = int(rng.integers(max(m + 3, 20), 150_000 if big else 3000))
    J = ctx.matrix(n, m).hash_fill(1 + case, 0, n, 1.0)
    Z = ctx.matrix(n, m); W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(J, Z, W=W)
    a = ctx.vector(n).hash_fill(3 + case, 0, 4.0, 5.0); b = ctx.vector(n).hash_fill(4 + case, 0)
    out = {}
    for mode in (-1, 0):
        ctx.set_onepass(mode)
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, a), L.DeviceBasis(Z), b, None, tol=1e-11, maxit=25)
        # pcg!
        w = L.ProjPenaltyWork(ctx, m, n, False)
        xp, rp = ctx.vector(n), ctx.vector(n); rp.copy_from(b)
        fl, pit = L.pcg_(1e-2 * float(S[0]) ** 2, _JacPlain(J, w), L.no_precondition, xp, rp, w.p, w.z, None, 1e-9 * float(S[0]), 12)
        # Newton retraction on c(x) = J'x - J'xs from a perturbed point (generator known)
        xs = ctx.vector(n).hash_fill(2 + case, 0); bd = ctx.vector(m); L.gemv_t(J, xs, bd)
        cons = L.DeviceConstraints(J, m, bd.download())
        xt = ctx.vector(n); L.waxpby(1.0, xs, 1e-3, b, xt)
        nrm = L.NR(L.DeviceBasis(Z, generator=(J, W)), S, Vt, 1e-9, 20, L.NRWork(m), False, None)
        xn, cv = ctx.vector(n), np.zeros(m)
        nfl, nit, _ = L.retract_(cv, xn, cons, xt, xs, nrm)
        out[mode] = (it, nr, x.download(), fl, pit, xp.download(), nfl, nit, xn.download())
    ctx.set_onepass(0)
    A, B = out[-1], out[0]
    def rel(u, v): return np.linalg.norm(u - v) / max(np.linalg.norm(v), 1e-300)
    ok = (A[0] == B[0] and rel(B[2], A[2]) < 1e-10 and (A[3], A[4]) == (B[3], B[4]) and rel(B[5], A[5]) < 1e-9
          and (A[6], A[7]) == (B[6], B[7]) and rel(B[8], A[8]) < 1e-11)
    # batched retractions vs one by one (needs 4 <= m <= 256)
    if 4 <= m <= 256:
        xts = [ctx.vector(n) for _ in range(3)]; xns = [ctx.vector(n) for _ in range(3)]
        for j, v in enumerate(xts): L.waxpby(1.0, xs, 1e-3 * 0.5 ** j, b, v)
        cvs = np.zeros((3, m)); got = L.retract_nr_batch_(cvs, xns, cons, xts, xs, nrm)
        one = ctx.vector(n)
        for j in range(3):
            c1 = np.zeros(m); f1, i1, _ = L.retract_(c1, one, cons, xts[j], xs, nrm)
            ok = ok and got is not None and (got[j][0], got[j][1]) == (f1, i1) and rel(xns[j].download(), one.download()) < 1e-11
    # the basis in factored form (Z == NULL: projcg, projection and Newton retraction stream J) against the materialised basis
    if 4 <= m <= 1024:
        Uf = L.DeviceBasis(None, m, generator=(J, W))
        xf, lf = ctx.vector(n), ctx.vector(m)
        itf, nrf = L.projcg_(xf, lf, L.DiagOperator(0.0, a), Uf, b, None, tol=1e-11, maxit=25)
        tf, tm = ctx.vector(m), ctx.vector(m)
        Uf.adjoint().mul_(tf, b); L.gemv_t(Z, b, tm)
        cf = np.zeros(m); nff, nif, _ = L.retract_(cf, xn, cons, xt, xs, L.NR(Uf, S, Vt, 1e-9, 20, L.NRWork(m), False, None))
        okf = (itf == B[0] and rel(xf.download(), B[2]) < 1e-10 and np.abs(tf.download() - tm.download()).max() < 1e-11 * max(1.0, np.abs(tm.download()).max())
               and (nff, nif) == (B[6], B[7]) and rel(xn.download(), B[8]) < 1e-11)
        if not okf:
            print("FACTORED MISMATCH", case, n, m, itf, B[0], rel(xf.download(), B[2]), (nff, nif), (B[6], B[7]))
        ok = ok and okf
    if not ok:
        bad += 1
        print("MISMATCH", case, n, m, A[0], B[0], rel(B[2], A[2]), (A[3], A[4]), (B[3], B[4]), rel(B[5], A[5]), (A[6], A[7]), (B[6], B[7]), rel(B[8], A[8]))
    for o in (J, Z, a, b):
        o.free()
print("fuzz: bad", bad, "of", cases)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
