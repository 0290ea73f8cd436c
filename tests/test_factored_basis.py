"""The tangent basis in FACTORED form U = Jct W (lfpsqp_basis.Z == NULL, FINDINGS.md §5.3): the n x m basis matrix of the reference's
ksvd! (src/la_helper.jl:8-34, used at src/optimize.jl:291-307, src/projcg.jl:55-118, src/retractions.jl:141) is never formed; the fused
projected-CG iteration, the projections and the Newton retraction stream Jct and apply the m x m factor W on the side.  Everything is
checked against the materialised basis Z = Jct W of the same factorisation and against the oracle."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .test_capi_parity import DiagOpRef


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def _setup(ctx, n, m, cond=1.0, seed=1):
    Jh = synth.hash_matrix(seed, n, m)
    if cond > 1.0:                                   # stretch the columns: an ill-conditioned block goes through the refinement rounds
        Jh = Jh * np.logspace(0, np.log10(cond), m)[None, :]
    Jct = ctx.matrix(n, m, np.asfortranarray(Jh))
    Z = ctx.matrix(n, m)
    W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(Jct, Z, W=W)
    W2 = np.zeros((m, m), order='F')
    S2, Vt2, rank2 = L.ksvd_(Jct, None, W=W2)        # the same factorisation without forming the basis
    return Jh, Jct, Z, W, S, Vt, rank, W2, S2, Vt2, rank2


@pytest.mark.parametrize("n,m,cond", [(1500, 12, 1.0), (2100, 128, 1.0), (1800, 20, 1e6)])
def test_factorisation_without_the_basis_returns_the_same_factors(dev_ctx, n, m, cond):
    ctx = dev_ctx
    Jh, Jct, Z, W, S, Vt, rank, W2, S2, Vt2, rank2 = _setup(ctx, n, m, cond)
    assert rank == rank2 == m
    np.testing.assert_array_equal(S, S2)
    np.testing.assert_array_equal(Vt, Vt2)
    np.testing.assert_array_equal(W, W2)
    np.testing.assert_allclose(Z.download(), Jh @ W, atol=1e-13 * max(1.0, cond))      # Z = Jct W is what the factored form applies


@pytest.mark.parametrize("n,m", [(2100, 4), (2500, 33), (1500, 128), (700, 300)])
def test_projcg_on_the_factored_basis_matches_the_materialised_one_and_the_oracle(dev_ctx, n, m):
    """projcg! (src/projcg.jl:40-121) with U = Jct W applied in factored form: counts, iterate and multipliers of the run on the
    materialised Z and of the oracle on Z's host copy -- zero and non-zero c, loose and tight tolerances."""
    ctx = dev_ctx
    Jh, Jct, Z, W, *_ = _setup(ctx, n, m)
    Zh = Z.download()
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    bh = synth.hash_vector(4, n)
    A = L.DiagOperator(0.0, ctx.vector(n, a))
    b = ctx.vector(n, bh)
    rng = np.random.default_rng(5)
    Uf, Um = L.DeviceBasis(None, m, generator=(Jct, W)), L.DeviceBasis(Z)
    tols = (1e-12,) if (_is_emu(ctx) and m >= 100) else (1e-6, 1e-12)      # (emulator: the wide shapes once)
    for ch in (None, rng.standard_normal(m)):
        for tol in tols:
            x0, l0 = np.zeros(n), np.zeros(m)
            i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Zh, bh, np.zeros(m) if ch is None else ch, tol=tol)
            out = {}
            for tag, U in (("factored", Uf), ("materialised", Um)):
                x, lam = ctx.vector(n), ctx.vector(m)
                c = None if ch is None else ctx.vector(m, ch)
                it, nr = L.projcg_(x, lam, A, U, b, c, tol=tol)
                out[tag] = (it, nr, x.download(), lam.download())
                assert it == i0 and nr < tol
                assert np.linalg.norm(out[tag][2] - x0) <= 1e-10 * np.linalg.norm(x0)
                assert np.abs(out[tag][3] - l0).max() < 1e-10
            assert out["factored"][0] == out["materialised"][0]
    # negative curvature: x = d / |d|, lambda = NaN, nr = Inf (src/projcg.jl:77-82)
    an = a.copy()
    an[::3] = -2.0
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(an), Zh, bh, np.zeros(m), tol=1e-10)
    x, lam = ctx.vector(n), ctx.vector(m)
    it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, an)), Uf, b, None, tol=1e-10)
    assert it == i0 and np.isinf(nr) and np.isinf(nr0) and np.all(np.isnan(lam.download()))
    np.testing.assert_allclose(x.download(), x0, atol=1e-10)


def test_projections_and_newton_retraction_on_the_factored_basis(dev_ctx):
    """kgemv!('T' / 'N') of the tangent projection (src/optimize.jl:305-308) and retract!(::NR) (src/retractions.jl:75-177) with the
    basis in factored form: equal to the materialised basis to rounding, flags and counts equal to the oracle's."""
    ctx = dev_ctx
    n, m = (900, 24) if _is_emu(ctx) else (6000, 64)
    Jh, Jct, Z, W, S, Vt, rank, *_ = _setup(ctx, n, m)
    Zh = Z.download()
    Uf = L.DeviceBasis(None, m, generator=(Jct, W))
    vh = synth.hash_vector(9, n)
    v, t = ctx.vector(n, vh), ctx.vector(m)
    Uf.adjoint().mul_(t, v)
    np.testing.assert_allclose(t.download(), Zh.T @ vh, atol=1e-12)
    y = ctx.vector(n, vh)
    Uf.mul_(y, t, -1.0, 1.0)                                           # y = v - U U'v
    np.testing.assert_allclose(y.download(), vh - Zh @ (Zh.T @ vh), atol=1e-12)
    # Newton retraction onto J x = b from a tangent step off a feasible point
    xs = synth.hash_vector(2, n)
    bvec = Jh.T @ xs
    cons = L.DeviceConstraints(Jct, m, bvec)
    step = y.download()
    xt_h = xs + 0.3 * step / np.linalg.norm(step) + 1e-3 * synth.hash_vector(7, n)

    def c_(cval, xx):
        cval[:m] = Jh.T @ xx - bvec

    x, xtilde, xnew = ctx.vector(n, xs), ctx.vector(n, xt_h), ctx.vector(n)
    nr0 = R.NR(Zh, S, Vt, 1e-9, 100, R.NRWork(m), False, R.InequalityData())
    xn0, cv0 = np.zeros(n), np.zeros(m)
    f0, i0, _ = R.retract_(cv0, xn0, c_, xt_h, xs, nr0)
    for U in (Uf, L.DeviceBasis(Z, generator=(Jct, W)), L.DeviceBasis(Z)):
        cval = np.zeros(m)
        flag, it, _ = L.retract_(cval, xnew, cons, xtilde, x, L.NR(U, S, Vt, 1e-9, 100, L.NRWork(m), False, None))
        assert (flag, it) == (f0, i0)
        np.testing.assert_allclose(xnew.download(), xn0, atol=1e-10)


@pytest.mark.parametrize("bounds", [False, True])
def test_optimize_is_the_same_with_and_without_the_materialised_basis(dev_ctx, bounds):
    """optimize (src/optimize.jl:119) on config 3 / config 4's shape: DeviceOptions.factored_basis on (default) and off give the same
    trajectory -- counts, step types, accepted steps, iterates to 1e-10."""
    ctx = dev_ctx
    n, m = (400, 8) if _is_emu(ctx) else (20000, 32)
    res = {}
    for factored in (True, False):
        ctx.options.factored_basis = factored
        if bounds:
            P0 = synth.BallBoxProblem(n, m)
            Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
            P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
            x0 = 0.97 * synth.hash_vector(2, n) + 0.015
        else:
            Jct = ctx.matrix(n, m).hash_fill(1)
            xs = ctx.vector(n).hash_fill(2)
            b = ctx.vector(m)
            L.gemv_t(Jct, xs, b)
            P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download())
            x0 = np.ones(n)
        tr = []
        x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=6), trace=tr)
        res[factored] = (tr, x, obj, ti)
    ctx.options.factored_basis = True
    (tr1, x1, o1, t1), (tr0, x0_, o0, t0) = res[True], res[False]
    assert t1.iter == t0.iter and len(tr1) == len(tr0)
    for a, b_ in zip(tr1, tr0):
        assert np.linalg.norm(a["x"] - b_["x"]) <= 1e-10 * max(1.0, np.linalg.norm(b_["x"]))
        for key in ("steptype", "mtype", "alpha", "ls_flag", "tn_iter", "rank"):
            assert a.get(key) == b_.get(key), (key, a.get(key), b_.get(key))
    np.testing.assert_allclose(o1, o0, rtol=1e-12)


@pytest.mark.parametrize("bounds", [False, True])
def test_sparse_twin_without_the_basis(dev_ctx, bounds):
    """Sparse constraint gradients (lfpsqp_spmat twin) with the basis in factored form: lfpsqp_factorize_sp(Z = NULL) returns the factors of the
    run that forms Z (no basis-forming product, no n x m basis in memory), and `optimize` on banded equalities -- projected CG, projection and
    Newton steps on the nonzeros -- follows the same trajectory with DeviceOptions.factored_basis on and off."""
    import scipy.sparse as sp
    ctx = dev_ctx
    n, m, k = (600, 8, 3) if _is_emu(ctx) else (30000, 32, 4)
    ii = np.arange(n)
    rows = np.repeat(ii, k)
    cols = ((((ii * m) // n)[:, None] + np.arange(k)[None, :]) % m).ravel()
    vals = (np.random.default_rng(5).standard_normal((n, k)) + 2.0 * (np.arange(k) == 0)).ravel()
    p = 1 if bounds else 0
    S = L.SparseMatrix(ctx, n + p, m, rows, cols, vals)
    Jct = ctx.matrix(n + p, m + p)
    S.to_dense(Jct)
    if not bounds:
        Z = ctx.matrix(n, m)
        W1, W2 = np.zeros((m, m), order='F'), np.zeros((m, m), order='F')
        S1, Vt1, r1 = L.ksvd_(Jct, Z, W=W1, Jsp=S)
        S2, Vt2, r2 = L.ksvd_(Jct, None, W=W2, Jsp=S)
        assert r1 == r2 == m
        np.testing.assert_array_equal(S1, S2)
        np.testing.assert_array_equal(W1, W2)
    xs = ctx.vector(n + p).hash_fill(2, 0, 1.0, 0.0)
    if bounds:
        ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
    b = ctx.vector(m + p)
    L.spmv_t(S, xs, b)
    res = {}
    for factored in (True, False):
        ctx.options.factored_basis = factored
        if bounds:
            xl = np.where((ii % 4 == 1) | (ii % 4 == 3), -1.0, -np.inf)
            xu = np.where((ii % 4 == 2) | (ii % 4 == 3), 1.0, np.inf)
            P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu, Jsp=S)
            x0 = 0.5 * np.ones(n)
        else:
            P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download(), Jsp=S)
            x0 = xs.download() + 0.05 * synth.hash_vector(6, n)
        tr = []
        maxiter = 2 if (_is_emu(ctx) and bounds) else 4                # (the bound problem's line searches: half a minute per iteration of launch emulation)
        x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter), trace=tr)
        res[factored] = (tr, obj, ti)
    ctx.options.factored_basis = True
    (tr1, o1, t1), (tr0, o0, t0) = res[True], res[False]
    assert t1.iter == t0.iter
    for a, b_ in zip(tr1, tr0):
        assert np.linalg.norm(a["x"] - b_["x"]) <= 1e-10 * max(1.0, np.linalg.norm(b_["x"]))
        for key in ("steptype", "mtype", "alpha", "ls_flag", "tn_iter", "rank", "retract_iter1"):
            assert a.get(key) == b_.get(key), (key, a.get(key), b_.get(key))
    np.testing.assert_allclose(o1, o0, rtol=1e-12)


@pytest.mark.parametrize("lib_kind", ["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("bounds", [False, True])
def test_optimize_with_the_one_pass_kernels_switched_off(request, lib_kind, bounds):
    """LFPSQP_ONEPASS=-1 (the documented cross-check path: every one-pass kernel replaced by its two-pass form) with the DEFAULT options:
    the factored basis needs the fused iteration, so the driver must ask the library (lfpsqp_factored_basis_supported), materialise Z and
    reach the same trajectory as with the one-pass kernels -- not fail with LFPSQP_ERR_UNSUPPORTED.  (src/optimize.jl:119, :291-307)"""
    lib = request.getfixturevalue("emu_lib" if lib_kind == "emu" else "gpu_lib")
    n, m = (2600, 6) if lib_kind == "emu" else (20000, 32)
    res = {}
    for onepass in (0, -1):
        ctx = L.Context(0, lib)
        try:
            ctx.set_onepass(onepass)
            assert ctx.options.factored_basis                     # the default
            Jprobe = ctx.matrix(n, m)
            assert ctx.factored_basis_supported(Jprobe) == (onepass == 0)
            Jprobe.free()
            if bounds:
                P0 = synth.BallBoxProblem(n, m)
                Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
                P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
                x0 = 0.97 * synth.hash_vector(2, n) + 0.015
            else:
                Jct = ctx.matrix(n, m).hash_fill(1)
                xs = ctx.vector(n).hash_fill(2)
                b = ctx.vector(m)
                L.gemv_t(Jct, xs, b)
                P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download())
                x0 = np.ones(n)
            tr = []
            x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=5), trace=tr)
            res[onepass] = (tr, x, obj, ti)
        finally:
            ctx.close()
    (tr1, x1, o1, t1), (tr0, x0_, o0, t0) = res[0], res[-1]
    assert t1.iter == t0.iter and len(tr1) == len(tr0)
    for a, b_ in zip(tr1, tr0):
        assert np.linalg.norm(a["x"] - b_["x"]) <= 1e-10 * max(1.0, np.linalg.norm(b_["x"]))
        for key in ("steptype", "mtype", "alpha", "ls_flag", "tn_iter", "rank"):
            assert a.get(key) == b_.get(key), (key, a.get(key), b_.get(key))
    np.testing.assert_allclose(o1, o0, rtol=1e-12)
