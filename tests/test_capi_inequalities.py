"""Device versions of the reference's test/test_inequalities.jl, checked against the oracle."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth


def _setup(ctx, n, m, seed=7):
    rng = np.random.default_rng(seed)
    i = np.arange(n)
    kind = i % 4                                   # none / lower / upper / both (test_inequalities.jl:6-9)
    lo = rng.standard_normal(n) - 1.0
    hi = lo + 0.5 + 2 * rng.random(n)
    xl = np.where((kind == 1) | (kind == 3), lo, -np.inf)
    xu = np.where((kind == 2) | (kind == 3), hi, np.inf)
    x = np.where(kind == 0, rng.standard_normal(n),
                 np.where(kind == 1, lo + 2 * rng.random(n), np.where(kind == 2, hi - 2 * rng.random(n), lo + rng.random(n) * (hi - lo))))
    xaug = np.zeros(2 * n)
    xaug[:n] = x
    idata0 = R.InequalityData(xl, xu)
    R.generate_initial_y_(xaug, idata0)
    idata = L.InequalityData(ctx, xl, xu)
    X = L.StackedVector(ctx, n).upload2(xaug)
    return rng, xl, xu, xaug, idata0, idata, X


@pytest.mark.parametrize("n,m", [(16, 5), (1027, 9), (3000, 4)])
def test_inequality_data_y_h_gradient(dev_ctx, n, m):
    ctx = dev_ctx
    rng, xl, xu, xaug, idata0, idata, X = _setup(ctx, n, m)
    for name in "qrst":
        np.testing.assert_array_equal(getattr(idata, name).download(), getattr(idata0, name))
    # generate_initial_y! on the device
    X2 = L.StackedVector(ctx, n).upload2(np.concatenate([xaug[:n], np.zeros(n)]))
    L.generate_initial_y_(X2, idata)
    np.testing.assert_allclose(X2.download2(), xaug, rtol=1e-15, atol=1e-15)
    # calculate_h!
    h = ctx.vector(n)
    hmax = L.calculate_h_(h, X, idata)
    c0 = np.zeros(n + m)
    R.calculate_h_(c0, xaug, idata0)
    np.testing.assert_allclose(h.download(), c0[:n], atol=4e-15)
    assert hmax == pytest.approx(np.abs(h.download()).max(), abs=0) and hmax < 1e-14
    # inequality_gradient!
    idc0 = R.InequalityDecomp(None, None, None, np.empty(n), np.empty(n), np.empty(n), None, m)
    R.inequality_gradient_(idc0, xaug, idata0)
    idc = L.InequalityDecomp(ctx, n, m)
    L.inequality_gradient_(idc, X, idata)
    np.testing.assert_allclose(idc.Dx.download(), idc0.Dx, rtol=1e-15, atol=1e-16)
    np.testing.assert_allclose(idc.Dy.download(), idc0.Dy, rtol=1e-15, atol=1e-16)
    np.testing.assert_allclose(idc.S.download(), idc0.S, rtol=1e-15)
    np.testing.assert_allclose(idc.sx.download(), idc0.Dy ** 2, rtol=1e-15, atol=1e-16)
    np.testing.assert_allclose(idc.sy.download(), -idc0.Dx * idc0.Dy, rtol=1e-15, atol=1e-16)


@pytest.mark.parametrize("n,m", [(16, 5), (2050, 12)])
def test_projection_operator_and_y_retraction(dev_ctx, n, m):
    """test_inequalities.jl:92-141 (Q mul!) and :180-199 (y_retract!) against the oracle's operators."""
    ctx = dev_ctx
    rng, xl, xu, xaug, idata0, idata, X = _setup(ctx, n, m)
    Jh = np.asfortranarray(rng.standard_normal((n, m)))
    # oracle decomposition (dgesvd of PJct)
    idc0 = R.InequalityDecomp(np.empty((2 * n, m), order='F'), np.empty(m), np.empty((m, m), order='F'),
                              np.empty(n), np.empty(n), np.empty(n), Jh, m)
    R.inequality_gradient_(idc0, xaug, idata0)
    PJ = np.asfortranarray(np.vstack([(1 - idc0.Dx ** 2)[:, None] * Jh, (-idc0.Dy * idc0.Dx)[:, None] * Jh]))
    R.ksvd_(PJ, idc0.U, idc0.Sigma, idc0.Vt)
    P0 = R.InequalityDecompProject(idc0)
    # device decomposition
    idc = L.InequalityDecomp(ctx, n, m, ctx.matrix(n, m, Jh))
    L.inequality_gradient_(idc, X, idata)
    w2 = ctx.vector(n, idc0.Dy ** 2)
    idc.Sigma, idc.Vt, idc.rank = L.ksvd_(idc.Jct, idc.Z, w2=w2)
    assert idc.rank == m
    np.testing.assert_allclose(idc.Sigma, idc0.Sigma, rtol=1e-11)
    P = L.InequalityDecompProject(idc)
    # projector I - QQ' applied to a random vector must agree (basis-rotation invariant)
    d = rng.standard_normal(2 * n)
    tmp0 = np.zeros(n + m)
    R.mul_(tmp0, R.adj(P0), d)
    d0 = d.copy()
    R.mul_(d0, P0, tmp0, -1.0, 1.0)
    D = L.StackedVector(ctx, n).upload2(d)
    w, t = ctx.vector(n), ctx.vector(m)
    P.mul_t(w, t, D)
    np.testing.assert_allclose(w.download(), tmp0[:n], atol=1e-14)
    np.testing.assert_allclose(np.linalg.norm(t.download()), np.linalg.norm(tmp0[n:]), rtol=1e-11)
    P.mul_n(D, w, t, -1.0, 1.0)
    np.testing.assert_allclose(D.download2(), d0, atol=1e-12)
    # the gap between the halves stays zero
    if D.hs > n:
        assert np.all(D.download(D.hs - n, n) == 0.0)
    # alpha/beta form with w = None (Newton-retraction update)
    Y = L.StackedVector(ctx, n).upload2(np.ones(2 * n))
    P.mul_n(Y, None, t, 2.0, 3.0)
    Zh = idc.Z.download()
    Ufull = np.vstack([idc.sx.download()[:, None] * Zh, idc.sy.download()[:, None] * Zh])
    np.testing.assert_allclose(Y.download2(), 2 * Ufull @ t.download() + 3, atol=1e-12)
    # y_retract! after a tangent step (test_inequalities.jl:180-199)
    xnew0 = xaug + d0
    R.y_retract_(xnew0, xaug, idata0)
    Xn = L.StackedVector(ctx, n).upload2(xaug + d0)
    L.y_retract_(Xn, X, idata)
    got = Xn.download2()
    np.testing.assert_allclose(got, xnew0, rtol=1e-12, atol=1e-12)
    h = ctx.vector(n)
    assert L.calculate_h_(h, Xn, idata) < 1e-11
    np.testing.assert_array_equal(X.download2(), xaug)


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("n,m", [(64, 5), (1500, 12)])
def test_projcg_with_bounds_matches_oracle(dev_ctx, n, m, sparse):
    """projcg! with U = InequalityDecompProject (src/optimize.jl:366-381) and the augmented diagonal
    Hessian (augmented_hess_lag_vec!, src/inequality_helper.jl:144-158): fused stacked kernels vs oracle.
    sparse: banded constraint gradients with a sparse twin -- the stacked basis applied in factored form on the nonzeros
    (lfpsqp_basis.SA: U t = [sx; sy] .* (Jct (W t)), same diagonal blocks)."""
    from tests.helpers import DiagOpRef
    ctx = dev_ctx
    rng, xl, xu, xaug, idata0, idata, X = _setup(ctx, n, m)
    Jh = synth.hash_matrix(1, n, m)
    Ssp = None
    if sparse:
        rows = np.repeat(np.arange(n), 3)
        cols = ((((np.arange(n) * m) // n)[:, None] + np.arange(3)[None, :]) % m).ravel()
        vals = (np.random.default_rng(17).standard_normal((n, 3)) + 2.0 * (np.arange(3) == 0)).ravel()
        Jh = np.zeros((n, m), order='F')
        np.add.at(Jh, (rows, cols), vals)
        Ssp = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    idc0 = R.InequalityDecomp(np.empty((2 * n, m), order='F'), np.empty(m), np.empty((m, m), order='F'),
                              np.empty(n), np.empty(n), np.empty(n), Jh, m)
    R.inequality_gradient_(idc0, xaug, idata0)
    PJ = np.asfortranarray(np.vstack([(1 - idc0.Dx ** 2)[:, None] * Jh, (-idc0.Dy * idc0.Dx)[:, None] * Jh]))
    R.ksvd_(PJ, idc0.U, idc0.Sigma, idc0.Vt)
    P0 = R.InequalityDecompProject(idc0)
    idc = L.InequalityDecomp(ctx, n, m, ctx.matrix(n, m, Jh))
    L.inequality_gradient_(idc, X, idata)
    Wg = np.zeros((m, m), order='F')
    idc.Sigma, idc.Vt, idc.rank = L.ksvd_(idc.Jct, idc.Z, w2=ctx.vector(n, idc0.Dy ** 2), W=Wg, Jsp=Ssp)
    if sparse:
        idc.W, idc.Jsp = Wg, Ssp
    P = L.InequalityDecompProject(idc)
    # A = diag: 2 + 2*lamy*q on the x-half, 2*lamy*s (+3 to keep it SPD) on the y-half
    lamy = 0.3 * rng.random(n)
    a = np.concatenate([2.0 + 2 * lamy * idata0.q, 3.0 + 2 * lamy * idata0.s])
    bh = rng.standard_normal(2 * n)
    tmp = np.zeros(n + m)
    R.mul_(tmp, R.adj(P0), bh)
    R.mul_(bh, P0, tmp, -1.0, 1.0)                      # rhs in the tangent space, like optimize's d
    Adev = L.DiagOperator(0.0, L.StackedVector(ctx, n).upload2(a))
    b = L.StackedVector(ctx, n).upload2(bh)
    work = L.ProjCGWork(ctx, 0, m, stacked_N=n)
    for tol, maxit in ((1e-8, None), (1e-300, 3)):
        x0, l0 = np.zeros(2 * n), np.zeros(n + m)
        i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), P0, bh, np.zeros(n + m), tol=tol, maxit=maxit)
        x, lam = L.StackedVector(ctx, n), ctx.vector(n + m)
        i1, nr1 = L.projcg_(x, lam, Adev, P, b, None, tol=tol, maxit=maxit, work=work)
        assert i1 == i0 and nr1 == pytest.approx(nr0, rel=1e-6)
        xd = x.download2()
        assert np.linalg.norm(xd - x0) <= 1e-10 * np.linalg.norm(x0)
        # Q'x = 0 (both blocks), checked with the oracle's operator
        R.mul_(tmp, R.adj(P0), xd)
        assert np.abs(tmp).max() < 1e-12
        np.testing.assert_allclose(lam.download()[:n], l0[:n], atol=1e-10)          # diagonal block of lambda
        assert np.linalg.norm(lam.download()[n:]) == pytest.approx(np.linalg.norm(l0[n:]), rel=1e-8, abs=1e-10)
    # negative curvature through the stacked path
    a2 = a.copy()
    a2[::2] *= -1
    x0, l0 = np.zeros(2 * n), np.zeros(n + m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a2), P0, bh, np.zeros(n + m), tol=1e-12)
    x, lam = L.StackedVector(ctx, n), ctx.vector(n + m)
    i1, nr1 = L.projcg_(x, lam, L.DiagOperator(0.0, L.StackedVector(ctx, n).upload2(a2)), P, b, None, tol=1e-12, work=work)
    assert i1 == i0 and np.isinf(nr1) and np.isinf(nr0) and np.all(np.isnan(lam.download()))
    assert np.linalg.norm(x.download2() - x0) < 1e-10


@pytest.mark.parametrize("n,m", [(16, 5), (1100, 7)])
def test_full_jacobian_operator_multipliers_and_hessian_diag(dev_ctx, n, m):
    """test_inequalities.jl:111-120 (InequalityDecomp mul! x3), :143-155 (calculate_lambda_kkt! == bigA \\ d)
    and :157-177 (augmented Hessian action, diagonal case) on the device."""
    ctx = dev_ctx
    rng, xl, xu, xaug, idata0, idata, X = _setup(ctx, n, m)
    Jh = np.asfortranarray(rng.standard_normal((n, m)))
    idc0 = R.InequalityDecomp(np.empty((2 * n, m), order='F'), np.empty(m), np.empty((m, m), order='F'),
                              np.empty(n), np.empty(n), np.empty(n), Jh, m)
    R.inequality_gradient_(idc0, xaug, idata0)
    PJ = np.asfortranarray(np.vstack([(1 - idc0.Dx ** 2)[:, None] * Jh, (-idc0.Dy * idc0.Dx)[:, None] * Jh]))
    R.ksvd_(PJ, idc0.U, idc0.Sigma, idc0.Vt)
    ghx, ghy = idc0.Dx * idc0.S, idc0.Dy * idc0.S
    bigA = np.block([[np.diag(ghx), Jh], [np.diag(ghy), np.zeros((n, m))]])
    idc = L.InequalityDecomp(ctx, n, m, ctx.matrix(n, m, Jh))
    L.inequality_gradient_(idc, X, idata)
    idc.Sigma, idc.Vt, idc.rank = L.ksvd_(idc.Jct, idc.Z, w2=idc.sx)
    op = L.InequalityDecompOp(idc)
    v = rng.standard_normal(n + m)
    w = rng.standard_normal(2 * n)
    vh, vc = ctx.vector(n, v[:n]), ctx.vector(m, v[n:])
    dest = L.StackedVector(ctx, n)
    op.mul_n(dest, vh, vc)
    np.testing.assert_allclose(dest.download2(), bigA @ v, atol=1e-13 * max(1, m))
    dest.upload2(np.ones(2 * n))
    op.mul_n(dest, vh, vc, 2.0, 3.0)
    np.testing.assert_allclose(dest.download2(), 2 * bigA @ v + 3, atol=1e-13 * max(1, m))
    dh, dc = ctx.vector(n), ctx.vector(m)
    op.mul_t(dh, dc, L.StackedVector(ctx, n).upload2(w))
    np.testing.assert_allclose(np.concatenate([dh.download(), dc.download()]), bigA.T @ w, atol=1e-12 * np.sqrt(n))
    # calculate_lambda_kkt!: [lamy; lam] == bigA \ d
    d = rng.standard_normal(2 * n)
    P = L.InequalityDecompProject(idc)
    qw, qt = ctx.vector(n), ctx.vector(m)
    P.mul_t(qw, qt, L.StackedVector(ctx, n).upload2(d))
    lam, lamy = np.zeros(m), ctx.vector(n)
    L.calculate_lambda_kkt_(lam, lamy, qw, qt, idc)
    ref, *_ = np.linalg.lstsq(bigA, d, rcond=None)
    np.testing.assert_allclose(np.concatenate([lamy.download(), lam]), ref, atol=1e-11)
    # augmented Hessian diagonal vs the oracle's operator applied to unit-free data
    hxh = 2.0 + rng.random(n)
    a = L.StackedVector(ctx, n)
    L.augmented_hess_diag_(a, ctx.vector(n, hxh), lamy, idata)
    src = rng.standard_normal(2 * n)
    dest0 = np.zeros(2 * n)
    R.augmented_hess_lag_vec_(dest0, src, lambda de, sr, xx, ll: de.__setitem__(slice(None), hxh * sr), xaug, lam, lamy.download(), idata0)
    np.testing.assert_allclose(a.download2() * src, dest0, rtol=1e-13, atol=1e-13)
