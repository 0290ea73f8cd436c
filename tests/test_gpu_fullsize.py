"""BASELINE.json's FULL size (n = 1e7, m = 128, one MI355X) through size-independent properties: the
oracle cannot run there in seconds, but linearity, projector idempotence, orthonormality and the CG
invariants must hold at any size (parity protocol)."""
import math

import numpy as np
import pytest

import lfpsqp_jl_amd as L

N, M = 10_000_000, 128


@pytest.fixture(scope="module")
def big():
    ctx = L.Context(0)
    Z = ctx.matrix(N, M).hash_fill(1)
    L.orthonormalize_(Z)
    yield ctx, Z
    ctx.close()


@pytest.mark.gpu
def test_full_size_basis_is_orthonormal_and_gemv_linear(big):
    ctx, Z = big
    G = L.gram(Z)
    assert np.abs(G - np.eye(M)).max() < 5e-14
    v1, v2 = ctx.vector(N).hash_fill(11), ctx.vector(N).hash_fill(12)
    t1, t2, t3 = ctx.vector(M), ctx.vector(M), ctx.vector(M)
    L.gemv_t(Z, v1, t1)
    L.gemv_t(Z, v2, t2)
    v3 = ctx.vector(N)
    L.waxpby(2.0, v1, -3.0, v2, v3)
    L.gemv_t(Z, v3, t3)
    np.testing.assert_allclose(t3.download(), 2.0 * t1.download() - 3.0 * t2.download(), atol=1e-11)
    # <Z t, v> == <t, Z'v>  (adjoint identity ties GEMV-N to GEMV-T)
    y = ctx.vector(N)
    L.gemv_n(Z, t2, y)
    assert L.dot(y, v1) == pytest.approx(float(t2.download() @ t1.download()), rel=1e-11)
    # projector idempotence: P = I - ZZ';  P(Pv) == Pv and Z'(Pv) == 0
    L.gemv_n(Z, t1, v1, -1.0, 1.0)                   # v1 <- P v1
    L.gemv_t(Z, v1, t3)
    assert np.abs(t3.download()).max() < 1e-10
    before = L.nrm2(v1)
    L.gemv_n(Z, t3, v1, -1.0, 1.0)
    assert L.nrm2(v1) == pytest.approx(before, rel=1e-13)


@pytest.mark.gpu
def test_full_size_projcg_invariants(big):
    """Solve the benchmark QP to convergence at n = 1e7 and check the KKT system the reference's test
    asserts (test_cg.jl:24-28): U'x = c = 0, nr < tol, (I - UU')(Ax - b) = 0, lambda = U'(b - Ax)."""
    ctx, Z = big
    a = ctx.vector(N).hash_fill(3, 0, 4.0, 5.0)
    b = ctx.vector(N).hash_fill(4)
    x, lam = ctx.vector(N), ctx.vector(M)
    tol = 1e-8
    it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, a), L.DeviceBasis(Z), b, None, tol=tol, maxit=500)
    assert nr < tol and 10 < it < 200
    t = ctx.vector(M)
    L.gemv_t(Z, x, t)
    assert np.abs(t.download()).max() < 1e-11              # U'x = 0
    r = ctx.vector(N)
    L.vmul(a, x, r)
    L.axpby(1.0, b, -1.0, r)                                # r = b - A x
    L.gemv_t(Z, r, t)
    np.testing.assert_allclose(t.download(), lam.download(), atol=1e-9)    # lambda = U'(b - Ax)
    L.gemv_n(Z, t, r, -1.0, 1.0)                            # projected residual
    assert L.nrm2(r) < 10 * tol
    # the same solve through the (2, non-NT) kernel variant agrees to rounding
    ctx.set_tuning(2, False)
    x2 = ctx.vector(N)
    it2, nr2 = L.projcg_(x2, None, L.DiagOperator(0.0, a), L.DeviceBasis(Z), b, None, tol=tol, maxit=500, want_lambda=False)
    ctx.set_tuning(0, True)
    assert it2 == it
    L.axpby(1.0, x, -1.0, x2)
    assert L.nrm2(x2) <= 1e-10 * L.nrm2(x)


@pytest.mark.gpu
def test_full_size_staged_stores_do_not_change_the_iterates(big, monkeypatch):
    """At n = 1e7 a workgroup of the fused kernel stores its span of the residual in two bursts (onepass_kernel STG).  With the bursts capped
    at 40 rounds (8 per span) the 12th iterate is the same, bit for bit: when a value is stored does not change what is stored."""
    ctx, Z = big
    res = []
    for cap in ("", "40"):
        if cap:
            monkeypatch.setenv("LFPSQP_STAGE_ROUNDS", cap)
        c = L.Context(0, ctx.L) if cap else ctx
        a = c.vector(N).hash_fill(3, 0, 4.0, 5.0)
        b = c.vector(N).hash_fill(4)
        x = c.vector(N)
        Zc = Z
        if c is not ctx:                     # (plain device buffers of the same process: the other context's kernels may read them)
            ctx.sync()
            Zc = c.matrix(N, M)
            Zc.copy_from(Z)
        it, nr = L.projcg_(x, None, L.DiagOperator(0.0, a), L.DeviceBasis(Zc), b, None, tol=1e-300, maxit=12, want_lambda=False)
        res.append((it, nr, x.download()))
        if c is not ctx:
            c.close()
    monkeypatch.delenv("LFPSQP_STAGE_ROUNDS", raising=False)
    assert res[0][0] == res[1][0] == 12 and res[0][1] == res[1][1]
    np.testing.assert_array_equal(res[0][2], res[1][2])


@pytest.mark.gpu
def test_projcg_against_the_c_oracle_port_at_2e6():
    """The largest size the C/OpenMP oracle port (oracle/projcg_port.c) finishes in seconds on the box's host
    cores: n = 2e6 at the headline m = 128 -- iterates within 1e-10 relative, equal iteration count, lambda to 1e-9."""
    from oracle import port
    n, m = 2_000_000, 128
    ctx = L.Context(0)
    Z = ctx.matrix(n, m).hash_fill(1)
    L.orthonormalize_(Z)
    Uh = Z.download()                                   # the SAME orthonormal basis for both sides
    a = port.hash_vector(3, n, 0, 4.0, 5.0)
    b = port.hash_vector(4, n)
    port.lib().port_set_num_threads(port.usable_cpus())
    x0, l0, it0, nr0 = port.projcg(a, np.asfortranarray(Uh), b, None, 1e-9, 500)
    x, lam = ctx.vector(n), ctx.vector(m)
    it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)), L.DeviceBasis(Z),
                       ctx.vector(n).hash_fill(4), None, tol=1e-9, maxit=500)
    assert it == it0 and nr == pytest.approx(nr0, rel=1e-5)
    xd = x.download()
    assert np.linalg.norm(xd - x0) <= 1e-10 * np.linalg.norm(x0)
    np.testing.assert_allclose(lam.download(), l0, atol=1e-9)
    ctx.close()


@pytest.fixture(scope="module")
def big_problem():
    """Config 3's constraint block at full size: Jct = hash matrix (seed 1), tangent setup through lfpsqp_factorize."""
    ctx = L.Context(0)
    J = ctx.matrix(N, M).hash_fill(1, 0, N, 1.0)
    Z = ctx.matrix(N, M)
    W = np.zeros((M, M), order="F")
    S, Vt, rank = L.ksvd_(J, Z, W=W)
    yield ctx, J, Z, S, Vt, W, rank
    ctx.close()


@pytest.mark.gpu
def test_full_size_tangent_setup_is_a_thin_svd(big_problem):
    """ksvd! (src/la_helper.jl:8-34) at n = 1e7: Z'Z = I, Sigma descending and positive, Vt orthogonal,
    Jct t = Z (Sigma .* (Vt t)) and Z t = Jct (W t) for a probe t."""
    ctx, J, Z, S, Vt, W, rank = big_problem
    assert rank == M and np.all(np.diff(S) <= 0) and S[-1] > 0
    assert np.abs(L.gram(Z) - np.eye(M)).max() < 5e-13
    assert np.abs(Vt @ Vt.T - np.eye(M)).max() < 1e-13
    t = np.cos(np.arange(M) * 0.37) + 0.1
    y1, y2 = ctx.vector(N), ctx.vector(N)
    L.gemv_n(J, ctx.vector(M, t), y1)
    L.gemv_n(Z, ctx.vector(M, S * (Vt @ t)), y2)
    ref = L.nrm2(y1)
    L.axpby(1.0, y1, -1.0, y2)
    assert L.nrm2(y2) <= 1e-12 * ref
    L.gemv_n(Z, ctx.vector(M, t), y1)
    L.gemv_n(J, ctx.vector(M, W @ t), y2)
    ref = L.nrm2(y1)
    L.axpby(1.0, y1, -1.0, y2)
    assert L.nrm2(y2) <= 1e-12 * ref


@pytest.mark.gpu
def test_full_size_newton_retraction(big_problem):
    """retract!(..., ::NR) (src/retractions.jl:75-177) at n = 1e7: converges onto c(x) = J x - b = 0 moving only inside
    range(U) (the properties of test_retractions.jl:94-101); the one-stream step, the two-stream step and the batched
    form (three trial points sharing the passes) give the same points in the same number of iterations."""
    ctx, J, Z, S, Vt, W, rank = big_problem
    xs = ctx.vector(N).hash_fill(2)
    bd = ctx.vector(M)
    L.gemv_t(J, xs, bd)
    cons = L.DeviceConstraints(J, M, bd.download())
    noise = ctx.vector(N).hash_fill(4)
    tol = 1e-7
    nrm = L.NR(L.DeviceBasis(Z, generator=(J, W)), S, Vt, tol, 30, L.NRWork(M), False, None)
    xts = [ctx.vector(N) for _ in range(3)]
    for j, v in enumerate(xts):
        L.waxpby(1.0, xs, 1e-2 * 0.5 ** j, noise, v)
    out = {}
    for mode in (0, -1):
        ctx.set_onepass(mode)
        xn, cv = ctx.vector(N), np.zeros(M)
        flag, it, _ = L.retract_(cv, xn, cons, xts[0], xs, nrm)
        out[mode] = (flag, it, xn, cv.copy())
    ctx.set_onepass(0)
    flag, it, xn, cv = out[0]
    assert flag == 0 and 1 <= it <= 10 and np.abs(cv).max() < tol
    c = ctx.vector(M)
    L.gemv_t(J, xn, c)
    assert np.abs(c.download() - bd.download()).max() < tol          # c(xnew) recomputed from scratch
    step = ctx.vector(N)
    L.waxpby(1.0, xn, -1.0, xts[0], step)                              # xnew - xtilde = U delta
    size = L.nrm2(step)
    L.gemv_t(Z, step, c)
    L.gemv_n(Z, c, step, -1.0, 1.0)
    assert size > 0 and L.nrm2(step) <= 1e-11 * size
    assert (out[-1][0], out[-1][1]) == (flag, it)
    L.waxpby(1.0, xn, -1.0, out[-1][2], step)
    assert L.nrm2(step) <= 1e-12 * L.nrm2(xn)
    # batched: three trial points, one pass per Newton iteration
    xns = [ctx.vector(N) for _ in range(3)]
    cvs = np.zeros((3, M))
    got = L.retract_nr_batch_(cvs, xns, cons, xts, xs, nrm)
    assert got is not None
    one = ctx.vector(N)
    for j in range(3):
        c1 = np.zeros(M)
        f1, i1, _ = L.retract_(c1, one, cons, xts[j], xs, nrm)
        assert (got[j][0], got[j][1]) == (f1, i1)
        L.axpby(1.0, xns[j], -1.0, one)
        assert L.nrm2(one) <= 1e-12 * L.nrm2(xns[j])


@pytest.mark.gpu
def test_full_size_pcg_solves_the_penalty_system(big_problem):
    """pcg! (src/retractions.jl:179-246) at n = 1e7: the returned x solves (J'J + mu I) x = b to the tolerance (true
    residual recomputed with plain GEMVs; the property of test_retractions.jl:121-139), one-pass iteration == two-pass."""
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    ctx, J, Z, S, Vt, W, rank = big_problem
    mu = 1e-2 * float(S[0]) ** 2
    b = ctx.vector(N).hash_fill(4)
    tol = 1e-8 * float(S[0])
    res = {}
    for mode in (0, -1):
        ctx.set_onepass(mode)
        w = L.ProjPenaltyWork(ctx, M, N, False)
        x, r = ctx.vector(N), ctx.vector(N)
        r.copy_from(b)
        flag, it = L.pcg_(mu, _JacPlain(J, w), L.no_precondition, x, r, w.p, w.z, None, tol, 60)
        res[mode] = (flag, it, x)
    ctx.set_onepass(0)
    flag, it, x = res[0]
    assert flag == 0 and 2 <= it < 60
    t, y = ctx.vector(M), ctx.vector(N)
    L.gemv_t(J, x, t)
    L.gemv_n(J, t, y)                       # y = Jct (Jct' x)
    L.axpby(mu, x, 1.0, y)                  # + mu x
    L.axpby(1.0, b, -1.0, y)                # b - (J'J + mu I) x
    assert L.nrm2(y) <= 2 * tol
    assert (res[-1][0], res[-1][1]) == (flag, it)
    L.waxpby(1.0, x, -1.0, res[-1][2], y)
    assert L.nrm2(y) <= 1e-10 * L.nrm2(x)


@pytest.mark.gpu
def test_full_size_config3_reaches_the_minimum_norm_point():
    """BASELINE config 3 end to end at n = 1e7, m = 128 (f = x'x s.t. J x = b, x0 = ones): terminates on kkt_tol at the
    feasible point of least norm -- c(x) = 0 and x in range(J') (its component outside range(U) vanishes)."""
    ctx = L.Context(0)
    Jct = ctx.matrix(N, M).hash_fill(1)
    xs = ctx.vector(N).hash_fill(2)
    b = ctx.vector(M)
    L.gemv_t(Jct, xs, b)
    P = L.QuadLinearBallBox(ctx, N, M, Jct, b.download())
    x, obj, lam, ti = P.optimize(np.ones(N), L.LFPSQPParams(disp=L.DisplayOption.off))
    assert ti.condition.name == "kkt_tol" and ti.iter <= 5 and ti.kkt_diff < 1e-6
    assert all(obj[k + 1] <= obj[k] for k in range(len(obj) - 1))
    xd = ctx.vector(N, x)
    c = ctx.vector(M)
    L.gemv_t(Jct, xd, c)
    assert np.abs(c.download() - b.download()).max() < 1e-6 * max(1.0, np.abs(b.download()).max())
    Z = ctx.matrix(N, M)
    L.ksvd_(Jct, Z)
    size = L.nrm2(xd)
    L.gemv_t(Z, xd, c)
    L.gemv_n(Z, c, xd, -1.0, 1.0)                     # (I - ZZ') x
    assert L.nrm2(xd) <= 1e-6 * size
    assert obj[-1] == pytest.approx(size * size, rel=1e-12)
    ctx.close()


@pytest.mark.gpu
def test_full_size_config4_ball_and_box():
    """BASELINE config 4 end to end at full size (N = 1e7 + 1, M = 129; ball + four-way bound pattern, Newton
    retraction, batched trial retractions): the objective decreases monotonically, the run ends on kkt_tol, the point is
    feasible for equalities, ball and box, and the objective is the one every earlier full-size run reached."""
    n, m = N, M
    ctx = L.Context(0)
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0)
    ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
    b = ctx.vector(m + 1)
    L.gemv_t(Jct, xs, b, ncols=m)
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    bh = b.download()[:m]
    P = L.QuadLinearBallBox(ctx, n, m, Jct, bh, R2=n / 2.0, xl=xl, xu=xu)
    x, obj, lam, ti = P.optimize(0.5 * np.ones(n), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off))
    assert ti.condition.name == "kkt_tol" and 10 <= ti.iter <= 30
    assert all(obj[k + 1] <= obj[k] for k in range(len(obj) - 1))
    assert obj[-1] == pytest.approx(47.71209, rel=1e-5)
    assert (x >= xl - 1e-9).all() and (x <= xu + 1e-9).all() and float(x @ x) <= n / 2.0 + 1e-6
    J2 = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
    c = ctx.vector(m)
    L.gemv_t(J2, ctx.vector(n, x), c)
    assert np.abs(c.download() - bh).max() < 1e-6
    ctx.close()


# ---- BASELINE configs[4] at its per-GPU shape: n = 4e7, m = 512 over 8 GPUs = 5e6 rows x 512 columns per rank ------------
N5, M5 = 5_000_000, 512


@pytest.fixture(scope="module")
def shard5():
    ctx = L.Context(0)
    J = ctx.matrix(N5, M5).hash_fill(1, 0, N5, 1.0)
    Z = ctx.matrix(N5, M5)
    W = np.zeros((M5, M5), order="F")
    S, Vt, rank = L.ksvd_(J, Z, W=W)
    yield ctx, J, Z, S, Vt, W, rank
    ctx.close()


@pytest.mark.gpu
def test_config5_shard_tangent_setup_is_a_thin_svd(shard5):
    """lfpsqp_factorize at 5e6 x 512 (four column panels in the Gram kernel, the tiled rmul): Z'Z = I, Sigma positive and
    descending, Vt orthogonal, Jct t = Z (Sigma .* (Vt t)), Z t = Jct (W t)."""
    ctx, J, Z, S, Vt, W, rank = shard5
    assert rank == M5 and np.all(np.diff(S) <= 0) and S[-1] > 0
    assert np.abs(L.gram(Z) - np.eye(M5)).max() < 2e-12
    assert np.abs(Vt @ Vt.T - np.eye(M5)).max() < 1e-12
    t = np.cos(np.arange(M5) * 0.37) + 0.1
    y1, y2 = ctx.vector(N5), ctx.vector(N5)
    L.gemv_n(J, ctx.vector(M5, t), y1)
    L.gemv_n(Z, ctx.vector(M5, S * (Vt @ t)), y2)
    ref = L.nrm2(y1)
    L.axpby(1.0, y1, -1.0, y2)
    assert L.nrm2(y2) <= 1e-11 * ref
    L.gemv_n(Z, ctx.vector(M5, t), y1)
    L.gemv_n(J, ctx.vector(M5, W @ t), y2)
    ref = L.nrm2(y1)
    L.axpby(1.0, y1, -1.0, y2)
    assert L.nrm2(y2) <= 1e-11 * ref


@pytest.mark.gpu
def test_config5_shard_projcg_wide_onepass_against_two_pass_and_kkt(shard5):
    """projcg! on the 512-column basis: the WIDE one-pass kernel (the four waves of a workgroup split the columns) and the
    two-pass kernels run the same solve -- equal iteration counts, iterates within 1e-10 -- and the result satisfies the
    KKT system of test_cg.jl:24-28 (U'x = 0, projected residual < tol, lambda = U'(b - A x))."""
    ctx, J, Z, S, Vt, W, rank = shard5
    a = ctx.vector(N5).hash_fill(3, 0, 4.0, 5.0)
    b = ctx.vector(N5).hash_fill(4)
    tol = 1e-8
    res = {}
    for mode in (0, -1):
        ctx.set_onepass(mode)
        x, lam = ctx.vector(N5), ctx.vector(M5)
        it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, a), L.DeviceBasis(Z), b, None, tol=tol, maxit=500)
        res[mode] = (it, nr, x, lam)
    ctx.set_onepass(0)
    it, nr, x, lam = res[0]
    assert nr < tol and 10 < it < 200
    assert res[-1][0] == it and res[-1][1] == pytest.approx(nr, rel=1e-5)
    d = ctx.vector(N5)
    L.waxpby(1.0, x, -1.0, res[-1][2], d)
    assert L.nrm2(d) <= 1e-10 * L.nrm2(x)
    np.testing.assert_allclose(lam.download(), res[-1][3].download(), atol=1e-9)
    t = ctx.vector(M5)
    L.gemv_t(Z, x, t)
    assert np.abs(t.download()).max() < 1e-11
    r = ctx.vector(N5)
    L.vmul(a, x, r)
    L.axpby(1.0, b, -1.0, r)
    L.gemv_t(Z, r, t)
    np.testing.assert_allclose(t.download(), lam.download(), atol=1e-9)
    L.gemv_n(Z, t, r, -1.0, 1.0)
    assert L.nrm2(r) < 10 * tol


@pytest.mark.gpu
def test_config5_shard_newton_retraction_and_pcg(shard5):
    """The retraction kernels at 512 columns (wide one-stream Newton step, wide one-pass pcg! iteration): the Newton
    retraction lands on J x = b moving inside range(U), one-stream == two-stream; pcg! solves (J'J + mu I) x = b."""
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    ctx, J, Z, S, Vt, W, rank = shard5
    xs = ctx.vector(N5).hash_fill(2)
    bd = ctx.vector(M5)
    L.gemv_t(J, xs, bd)
    cons = L.DeviceConstraints(J, M5, bd.download())
    noise = ctx.vector(N5).hash_fill(4)
    xt = ctx.vector(N5)
    L.waxpby(1.0, xs, 1e-2, noise, xt)
    tol = 1e-7
    nrm = L.NR(L.DeviceBasis(Z, generator=(J, W)), S, Vt, tol, 30, L.NRWork(M5), False, None)
    out = {}
    for mode in (0, -1):
        ctx.set_onepass(mode)
        xn, cv = ctx.vector(N5), np.zeros(M5)
        flag, it, _ = L.retract_(cv, xn, cons, xt, xs, nrm)
        out[mode] = (flag, it, xn, cv.copy())
    ctx.set_onepass(0)
    flag, it, xn, cv = out[0]
    assert flag == 0 and 1 <= it <= 10 and np.abs(cv).max() < tol
    assert (out[-1][0], out[-1][1]) == (flag, it)
    step = ctx.vector(N5)
    L.waxpby(1.0, xn, -1.0, out[-1][2], step)
    assert L.nrm2(step) <= 1e-12 * L.nrm2(xn)
    c = ctx.vector(M5)
    L.gemv_t(J, xn, c)
    assert np.abs(c.download() - bd.download()).max() < tol
    mu = 1e-2 * float(S[0]) ** 2
    b = ctx.vector(N5).hash_fill(4)
    ptol = 1e-8 * float(S[0])
    w = L.ProjPenaltyWork(ctx, M5, N5, False)
    x, r = ctx.vector(N5), ctx.vector(N5)
    r.copy_from(b)
    flag, it = L.pcg_(mu, _JacPlain(J, w), L.no_precondition, x, r, w.p, w.z, None, ptol, 60)
    assert flag == 0 and 2 <= it < 60
    t, y = ctx.vector(M5), ctx.vector(N5)
    L.gemv_t(J, x, t)
    L.gemv_n(J, t, y)
    L.axpby(mu, x, 1.0, y)
    L.axpby(1.0, b, -1.0, y)
    assert L.nrm2(y) <= 2 * ptol


@pytest.mark.gpu
def test_config5_shape_two_ranks_share_the_gpu():
    """The sharded path at 512 columns: bench.py --gpus 2 as the driver launches it (two processes, each 2.5e6 of the 5e6
    rows, the library's collectives staged through host gloo because both ranks sit on the one GPU of this box) runs the
    same solve as one rank: equal iteration count, ||x|| to 1e-10."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "6", "--warmup", "2", "--rows", str(N5), "--cols", str(M5), "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from .test_p2p_transport import _bench                      # (one more attempt with a fresh port if the launch of the ranks fails)
    two = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--comm", "host-gloo", "--device", "0",
                  *common], env, timeout=1200, port_flag=9)
    d2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][0])
    one = _bench([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", *common], env, timeout=1200)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["config"]["rows_per_gpu"] * 2 >= N5 and "configs[4]" in d2["config"]["workload"]
    assert d1["check"]["iters"] == d2["check"]["iters"] == 8
    assert abs(d1["check"]["x_norm"] - d2["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]
    assert abs(d1["check"]["nr"] - d2["check"]["nr"]) <= 1e-8 * d1["check"]["nr"]


# ---- the nonlinear (elementwise) constraint class at full size: size-independent properties ------------------------------
def _ew_big(ctx, sparse):
    n, m = N, M
    kind = (np.arange(n) % 3).astype(np.float64)                     # t, sin t, t^2 in turn
    if sparse:                                                        # four nonzeros per row, banded over the constraints
        rows = np.repeat(np.arange(n, dtype=np.int64), 4)
        cols = (np.repeat((np.arange(n, dtype=np.int64) * m) // n, 4) + np.tile(np.arange(4, dtype=np.int64), n)) % m
        vals = np.cos(0.37 * np.arange(4 * n)) + 1.5
        A = L.SparseMatrix(ctx, n, m, rows, cols, vals)
        cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=kind)
    else:
        A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
        cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=kind, qw=1e-7 * np.cos(np.arange(m)))
    return cons, n, m


@pytest.mark.gpu
@pytest.mark.parametrize("sparse", [False, True])
def test_full_size_elementwise_constraints(sparse):
    """c!, jac!, the Hessian diagonal and the Newton retraction of the nonlinear class at n = 1e7, m = 128: jac! is the derivative of c!
    (central differences along a random direction), hess_diag! the derivative of jac!'lam, the retraction lands on c = 0 inside
    range(U), and its cval is bit for bit c!(xnew) (test/test_retractions.jl:94-101 at full size)."""
    ctx = L.Context(0)
    cons, n, m = _ew_big(ctx, sparse)
    x = ctx.vector(n).hash_fill(31, 0, 0.5, 0.0)
    v = ctx.vector(n).hash_fill(32)
    c0, cp, cm = np.zeros(m), np.zeros(m), np.zeros(m)
    cons.jac_(cons.Jct, c0, x)
    t = ctx.vector(m)
    L.gemv_t(cons.Jct, v, t)
    jv = t.download()
    eps = 1e-5
    xp, xm = ctx.vector(n), ctx.vector(n)
    L.waxpby(1.0, x, eps, v, xp)
    L.waxpby(1.0, x, -eps, v, xm)
    cons.c_(cp, xp)
    cons.c_(cm, xm)
    np.testing.assert_allclose((cp - cm) / (2 * eps), jv, rtol=1e-6, atol=1e-6 * np.abs(jv).max())
    # Hessian diagonal: d/deps [Jct(x + eps v) lam] = hdiag .* v
    lam = np.cos(1.0 + np.arange(m))
    lam_d = ctx.vector(m, lam)
    gp, gm, hx, hv = ctx.vector(n), ctx.vector(n), ctx.vector(n), ctx.vector(n)
    ctmp = np.zeros(m)
    cons.jac_(cons.Jct, ctmp, xp)
    L.gemv_n(cons.Jct, lam_d, gp)
    cons.jac_(cons.Jct, ctmp, xm)
    L.gemv_n(cons.Jct, lam_d, gm)
    L.waxpby(1.0 / (2 * eps), gp, -1.0 / (2 * eps), gm, gp)                     # finite-difference Hessian-vector product
    cons.hess_diag_(hx, x, lam)
    L.vmul(hx, v, hv)
    L.axpby(-1.0, hv, 1.0, gp)
    assert L.nrm2(gp) <= 1e-6 * max(L.nrm2(hv), 1.0)
    # Newton retraction from a tangent step
    cons.jac_(cons.Jct, c0, x)
    cons.b = cons.b + c0                                                         # make x feasible
    Z = ctx.matrix(n, m)
    W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(cons.Jct, Z, W=W, Jsp=cons.Jsp)
    assert rank == m
    step = ctx.vector(n).hash_fill(33)
    L.gemv_t(Z, step, t)
    L.gemv_n(Z, t, step, -1.0, 1.0)
    xt, xnew = ctx.vector(n), ctx.vector(n)
    L.waxpby(1.0, x, 30.0 / L.nrm2(step), step, xt)                             # (a step long enough to leave the manifold by >> tol)
    nr = L.NR(L.DeviceBasis(Z, generator=(cons.Jct, W)), S, Vt, 1e-9, 100, L.NRWork(m), False, None)
    cval, cval2 = np.zeros(m), np.zeros(m)
    ctx.sync()
    import time
    t0 = time.perf_counter()
    flag, it, _ = L.retract_(cval, xnew, cons, xt, x, nr)
    dt = time.perf_counter() - t0
    cons.c_(cval2, xnew)
    assert flag == 0 and 0 < it < 30 and np.abs(cval).max() < 1e-9
    np.testing.assert_array_equal(cval, cval2)
    L.waxpby(1.0, xnew, -1.0, xt, gp)                                            # the correction lies in range(U)
    L.gemv_t(Z, gp, t)
    L.gemv_n(Z, t, gp, -1.0, 1.0)
    assert L.nrm2(gp) < 1e-9
    print(f"[elementwise n=1e7 m=128 {'sparse' if sparse else 'dense'}] Newton retraction: {it} iterations, {dt * 1e3 / max(it, 1):.2f} ms per iteration")
    ctx.close()
