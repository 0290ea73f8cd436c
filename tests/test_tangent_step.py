"""The tangent step of an outer iteration with fewer passes over the constraint gradients.

Reference sequence (src/optimize.jl:286-343, 366-381; src/projcg.jl:55-62): ksvd!, tmp_m = U'd, d -= U tmp_m, lambda_kkt = V S^-1 tmp_m,
hess_lag_vec!'s constraint term, then projcg! starts with r = A x0 - b, Utr = U'r.  On the device
  * lfpsqp_gram_rhs / lfpsqp_factorize_rhs sum Jct'd during the Gram pass (extra right-hand columns of the MFMA kernel),
  * lfpsqp_tangent_step projects d, completes the Hessian diagonal and forms U'r0 in one pass,
  * lfpsqp_projcg(LFPSQP_PROJCG_START_GIVEN) starts from that state.
Checked here against numpy restatements of the reference statements (products) and against the statement-by-statement device sequence
(solver results)."""
import ctypes as C

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import _capi
from oracle import synth


def _view_parts(ctx, n, m, rs, ru):
    rs_h = (synth.hash_vector(11, n) * 1.5) if rs else None          # both signs, and an exact zero below
    if rs_h is not None:
        rs_h[5] = 0.0
    u_h = synth.hash_vector(12, n) if ru else None
    w_h = 0.3 * synth.hash_vector(13, m) if ru else None
    return rs_h, u_h, w_h


@pytest.mark.parametrize("n,m,weighted,rs,ru,nx", [
    (1500, 12, False, False, False, 1), (2100, 128, False, False, False, 2), (1900, 130, True, False, False, 1),
    (1700, 64, True, True, False, 2), (2300, 128, False, True, True, 1), (2300, 40, True, True, True, 2), (1300, 260, False, False, True, 2),
    (1100, 131, True, True, True, 1), (700, 300, True, False, False, 2)])
def test_gram_with_extra_right_hand_columns(dev_ctx, n, m, weighted, rs, ru, nx):
    """X[:, k] = V'(sqrt(w2) .* e_k) from the Gram pass itself, for plain matrices and views V = diag(rs) A + u w', weighted or not,
    panels and border columns."""
    ctx = dev_ctx
    Ah = synth.hash_matrix(1, n, m)
    A = ctx.matrix(n, m, np.asfortranarray(Ah))
    rs_h, u_h, w_h = _view_parts(ctx, n, m, rs, ru)
    Vh = Ah.copy()
    M = A
    if rs or ru:
        M = A.view(ctx.vector(n, rs_h) if rs else None, ctx.vector(n, u_h) if ru else None, ctx.vector(m, w_h) if ru else None)
        if rs:
            Vh = rs_h[:, None] * Vh
        if ru:
            Vh = Vh + np.outer(u_h, w_h)
    w2h = synth.hash_vector(5, n) ** 2 if weighted else None          # (includes tiny weights)
    if weighted:
        w2h[7] = 0.0
    w2 = ctx.vector(n, w2h) if weighted else None
    es_h = [synth.hash_vector(20 + k, n) for k in range(nx)]
    es = [ctx.vector(n, e) for e in es_h]
    G, X = L.gram_rhs(M, es, w2=w2)
    sw = np.sqrt(w2h) if weighted else np.ones(n)
    Gref = (sw[:, None] * Vh).T @ (sw[:, None] * Vh)
    scale = np.abs(Gref).max()
    np.testing.assert_allclose(G, Gref, atol=1e-12 * scale)
    np.testing.assert_allclose(G, L.gram(M, w2=w2), atol=1e-13 * scale)
    for k in range(nx):
        ref = Vh.T @ (sw * es_h[k])
        np.testing.assert_allclose(X[:, k], ref, atol=1e-12 * max(np.abs(ref).max(), 1.0) * np.sqrt(n))


def test_gram_extra_columns_ignore_stale_pad_rows(dev_ctx):
    """One diverged iterate must not poison later solves on the same context: the extra right-hand columns of the Gram pass are read in 16-row
    steps, and for a VIEW they live in a scratch slot of the context whose rows beyond n keep whatever an earlier, larger call left there.  After a
    call with NaN row scales on 4000 rows, a fully valid call on a 500-row view must return finite, correct products (the kernel masks the
    column's entries of rows >= n exactly like the weights; 0 * NaN would otherwise reach X, and G through a rank-one column)."""
    ctx = dev_ctx
    m = 8
    for ru in (False, True):
        big = ctx.matrix(4000, m, np.asfortranarray(synth.hash_matrix(1, 4000, m)))
        bad = big.view(ctx.vector(4000, np.full(4000, np.nan)), ctx.vector(4000, np.full(4000, np.nan)) if ru else None,
                       ctx.vector(m, 0.3 * synth.hash_vector(13, m)) if ru else None)
        Gb, Xb = L.gram_rhs(bad, [ctx.vector(4000, np.full(4000, np.nan))])
        assert not np.isfinite(Xb).all()                                   # (the poisoned call itself)
        n = 500
        Ah = synth.hash_matrix(2, n, m)
        rs_h = 1.0 + 0.5 * synth.hash_vector(11, n)
        u_h = synth.hash_vector(12, n) if ru else None
        w_h = 0.3 * synth.hash_vector(13, m) if ru else None
        V = ctx.matrix(n, m, np.asfortranarray(Ah)).view(ctx.vector(n, rs_h), ctx.vector(n, u_h) if ru else None, ctx.vector(m, w_h) if ru else None)
        Vh = rs_h[:, None] * Ah + (np.outer(u_h, w_h) if ru else 0.0)
        e_h = synth.hash_vector(20, n)
        G, X = L.gram_rhs(V, [ctx.vector(n, e_h)])
        assert np.isfinite(G).all() and np.isfinite(X).all()
        np.testing.assert_allclose(G, Vh.T @ Vh, atol=1e-12 * np.abs(Vh.T @ Vh).max())
        np.testing.assert_allclose(X[:, 0], Vh.T @ e_h, atol=1e-11)
        W = np.zeros((m, m), order='F')
        S, Vt, rank, Jtd = L.ksvd_(V, None, W=W, rhs=ctx.vector(n, e_h))      # (the factorisation that failed with "non-finite Gram matrix")
        assert rank == m and np.isfinite(S).all()
        np.testing.assert_allclose(Jtd, Vh.T @ e_h, atol=1e-11)


@pytest.mark.parametrize("n,m", [(1800, 16), (2500, 128), (1200, 129)])
def test_factorize_with_right_hand_side_returns_the_same_factors(dev_ctx, n, m):
    ctx = dev_ctx
    Mh = synth.hash_matrix(3, n, m)
    M = ctx.matrix(n, m, np.asfortranarray(Mh))
    dh = synth.hash_vector(8, n)
    d = ctx.vector(n, dh)
    W0, W1 = np.zeros((m, m), order='F'), np.zeros((m, m), order='F')
    S0, Vt0, r0 = L.ksvd_(M, None, W=W0)
    S1, Vt1, r1, Jtd = L.ksvd_(M, None, W=W1, rhs=d)
    assert r0 == r1 == m
    np.testing.assert_array_equal(S0, S1)
    np.testing.assert_array_equal(W0, W1)
    np.testing.assert_array_equal(Vt0, Vt1)
    np.testing.assert_allclose(Jtd, Mh.T @ dh, atol=1e-12 * np.sqrt(n))


def _tangent_step(ctx, U, S, Vt, m, Jtd, d, cons, x, hd, work, G=None):
    """G given: LFPSQP_TANGENT_INIT_PROJCG (the pass is also projcg!'s initial projection)."""
    Utd, lam = np.zeros(m), np.zeros(m)
    ss = C.c_double()
    bs, wc = U._c(), work._c()
    cc = cons._c() if cons is not None else None
    ctx.check(ctx.L.lfpsqp_tangent_step(ctx.h, C.byref(bs), S.ctypes.data, Vt.ctypes.data, m, Jtd.ctypes.data,
                                        G.ctypes.data if G is not None else None, d.h,
                                        C.byref(cc) if cc is not None else None, x.h if x is not None else None,
                                        hd.h if hd is not None else None, None, None, None, None, C.byref(wc), 1 if G is not None else 0,
                                        Utd.ctypes.data, lam.ctypes.data, C.byref(ss)))
    return Utd, lam, ss.value


@pytest.mark.parametrize("n,m", [(1800, 16), (2600, 128), (1500, 300)])
def test_tangent_step_matches_the_statement_sequence(dev_ctx, n, m):
    """Linear constraints: projection, multipliers, r0 and U'r0 against numpy; then projcg_ from the given start against projcg_ from scratch."""
    ctx = dev_ctx
    Jh = synth.hash_matrix(3, n, m)
    Jct = ctx.matrix(n, m, np.asfortranarray(Jh))
    dh = synth.hash_vector(8, n)
    d = ctx.vector(n, dh)
    W = np.zeros((m, m), order='F')
    S, Vt, rank, Jtd = L.ksvd_(Jct, None, W=W, rhs=d)
    assert rank == m
    U = L.DeviceBasis(None, rank, generator=(Jct, W))
    work = L.ProjCGWork(ctx, n, m)
    Utd, lam, ss = _tangent_step(ctx, U, S, Vt, m, Jtd, d, None, None, None, work)
    Zh = Jh @ W
    tref = Zh.T @ dh
    np.testing.assert_allclose(Utd, tref, atol=1e-12 * np.linalg.norm(dh))
    np.testing.assert_allclose(lam, Vt.T @ (tref / S), atol=1e-11 * np.linalg.norm(dh) / S[-1])
    dref = dh - Zh @ tref
    np.testing.assert_allclose(d.download(), dref, atol=1e-13 * np.linalg.norm(dh))
    np.testing.assert_allclose(work.rp.download(), -dref, atol=1e-13 * np.linalg.norm(dh))
    np.testing.assert_allclose(ss, dref @ dref, rtol=1e-12)
    np.testing.assert_allclose(work.Utr.download()[:m], Zh.T @ (-d.download()), atol=1e-12 * np.linalg.norm(dh))
    # the projected-CG solve from the given start == the solve from scratch
    a = ctx.vector(n, 4.0 * synth.hash_vector(9, n) + 5.0)
    A = L.DiagOperator(0.0, a)
    x1, x2 = ctx.vector(n), ctx.vector(n)
    tol = 1e-9 * np.sqrt(ss)
    iters = _capi.c_i64()
    nrv = C.c_double()
    a_c, u_c, w_c = A._c(), U._c(), work._c()
    ctx.check(ctx.L.lfpsqp_projcg(ctx.h, x1.h, None, C.byref(a_c), C.byref(u_c), d.h, None, float(tol), 200, n, 4, C.byref(w_c),
                                  C.byref(iters), C.byref(nrv)))
    work2 = L.ProjCGWork(ctx, n, m)
    i2, nr2 = L.projcg_(x2, None, A, U, d, None, tol=tol, maxit=200, work=work2, want_lambda=False)
    assert iters.value == i2 and iters.value > 3
    np.testing.assert_allclose(x1.download(), x2.download(), atol=1e-11 * np.linalg.norm(x2.download()))
    np.testing.assert_allclose(nrv.value, nr2, rtol=1e-6)


@pytest.mark.parametrize("n,m,qw", [(2200, 16, False), (2600, 128, True), (1900, 24, True)])
def test_tangent_step_completes_the_hessian_diagonal_of_the_streamed_class(dev_ctx, n, m, qw):
    """Nonlinear class with streamed gradients (Jct a view of A): the projection runs over the view, phi''(x) .* (A lam) over the plain A, in
    the same pass; against lfpsqp_constraints_hess_diag and the statement-by-statement products."""
    ctx = dev_ctx
    A = ctx.matrix(n, m, np.asfortranarray(2.0 ** -3 * synth.hash_matrix(21, n, m)))
    kind = (np.arange(n) % 3).astype(np.float64)
    cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=kind, qw=(1e-2 * np.cos(np.arange(m)) if qw else None), stream=True)
    assert cons.streamed
    xh = 0.5 * synth.hash_vector(31, n)
    x = ctx.vector(n, xh)
    cv = np.zeros(m)
    cons.jac_(cons.Jct, cv, x)
    Jh = np.asfortranarray(np.zeros((n, m)))
    tmp = ctx.matrix(n, m)
    tmp.copy_from(cons.Jct)                                         # (materialises the view)
    Jh = tmp.download()
    dh = synth.hash_vector(8, n)
    d = ctx.vector(n, dh)
    W = np.zeros((m, m), order='F')
    S, Vt, rank, Jtd = L.ksvd_(cons.Jct, None, W=W, rhs=d)
    assert rank == m
    np.testing.assert_allclose(Jtd, Jh.T @ dh, atol=1e-12 * np.sqrt(n))
    U = L.DeviceBasis(None, rank, generator=(cons.Jct, W))
    work = L.ProjCGWork(ctx, n, m)
    base = 2.0 + synth.hash_vector(41, n) ** 2
    hd = ctx.vector(n, base)
    Utd, lam, ss = _tangent_step(ctx, U, S, Vt, m, Jtd, d, cons, x, hd, work)
    Zh = Jh @ W
    tref = Zh.T @ dh
    dref = dh - Zh @ tref
    np.testing.assert_allclose(Utd, tref, atol=1e-12 * np.linalg.norm(dh))
    np.testing.assert_allclose(d.download(), dref, atol=1e-13 * np.linalg.norm(dh))
    np.testing.assert_allclose(work.Utr.download()[:m], Zh.T @ (-d.download()), atol=1e-12 * np.linalg.norm(dh))
    href = ctx.vector(n, base)
    cons.hess_diag_(href, x, lam)
    np.testing.assert_allclose(hd.download(), href.download(), rtol=1e-13, atol=1e-13)


class _LowRankRef:
    """A = diag(a) + V diag(sigma) V' as an oracle operator (mul! protocol of oracle/lfpsqp_ref.py)."""

    def __init__(self, a, V, sigma):
        self.a, self.V, self.sigma = a, V, sigma

    def _apply(self, v):
        return self.a * v + self.V @ (self.sigma * (self.V.T @ v))

    def mul_(self, dest, v, al=None, be=None):
        if al is None:
            dest[:] = self._apply(v)
        else:
            dest[:] = al * self._apply(v) + be * dest
        return dest

    def adjoint(self):
        return self


@pytest.mark.parametrize("n,m,k,factored", [(1800, 10, 1, False), (2600, 128, 3, False), (2200, 33, 8, False), (2400, 130, 2, True), (1500, 300, 4, False)])
def test_projcg_with_a_diagonal_plus_low_rank_operator_on_one_pass(dev_ctx, n, m, k, factored):
    """lfpsqp_projcg_lowrank: A = diag(a) + V diag(sigma) V' (sigma of both signs) on the fused ONE-pass iteration -- counts, iterates and
    multipliers of the oracle's projcg! (src/projcg.jl:40-121) with A as a LinearMap; c != 0; the negative-curvature exit; and against the
    callback path (lfpsqp_projcg_op, two passes per iteration) with the same operator."""
    from oracle import lfpsqp_ref as R
    ctx = dev_ctx
    Uh, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
    Uh = np.asfortranarray(Uh)
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    bh = synth.hash_vector(4, n)
    Vh = np.asfortranarray(synth.hash_matrix(17, n, k) / np.sqrt(n))
    sigma = np.array([3.0, -0.4, 1.5, 0.7, -0.2, 2.2, 0.9, 1.1])[:k]          # A stays positive definite (|sigma_j| |V_j|^2 < min a)
    if factored:                                   # the basis kept as U = J W (lfpsqp_basis.Z == NULL)
        Jh = synth.hash_matrix(5, n, m)
        J = ctx.matrix(n, m, np.asfortranarray(Jh))
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(J, None, W=W)
        U = L.DeviceBasis(None, rank, generator=(J, W))
        Uh = np.asfortranarray(Jh @ W)
    else:
        U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = L.LowRankOperator(0.0, ctx.vector(n, a), ctx.matrix(n, k, Vh), k, sigma)
    Aref = _LowRankRef(a, Vh, sigma)
    b = ctx.vector(n, bh)
    work = L.ProjCGWork(ctx, n, m)
    for ch, tol in ((None, 1e-10), (np.linspace(-1, 1, m), 1e-12)):
        x0, l0 = np.zeros(n), np.zeros(m)
        i0, nr0 = R.projcg_(x0, l0, Aref, Uh, bh, np.zeros(m) if ch is None else ch, tol=tol)
        x, lam = ctx.vector(n), ctx.vector(m)
        i1, nr1 = L.projcg_(x, lam, A, U, b, None if ch is None else ctx.vector(m, ch), tol=tol, work=work)
        assert i1 == i0 and i1 > 3 and nr1 == pytest.approx(nr0, rel=1e-5)
        assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
        assert np.abs(lam.download() - l0).max() < 1e-10
        if factored:
            continue                               # (the callback path needs a materialised basis)
        # the callback path with the same operator: the same solve
        A.fused = False
        x2, lam2 = ctx.vector(n), ctx.vector(m)
        i2, nr2 = L.projcg_(x2, lam2, A, U, b, None if ch is None else ctx.vector(m, ch), tol=tol)
        A.fused = True
        assert i2 == i1
        assert np.linalg.norm(x.download() - x2.download()) <= 1e-10 * np.linalg.norm(x0)
    # negative curvature (src/projcg.jl:77-82): one strongly negative direction
    sneg = sigma.copy()
    sneg[0] = -40.0 * n
    A2 = L.LowRankOperator(0.0, ctx.vector(n, a), ctx.matrix(n, k, Vh), k, sneg)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, _LowRankRef(a, Vh, sneg), Uh, bh, np.zeros(m), tol=1e-10)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, A2, U, b, None, tol=1e-10, work=work)
    assert (i1, nr1) == (i0, nr0) and np.isinf(nr1)
    assert np.linalg.norm(x.download() - x0) <= 1e-10 and np.all(np.isnan(lam.download()))


@pytest.mark.parametrize("n,m", [(1700, 8), (2400, 129)])
def test_tangent_step_with_bounds_matches_the_statement_sequence(dev_ctx, n, m):
    """Bound-stacked form: against the statement-by-statement device calls (mul!(tmp, Q', d), mul!(d, Q, tmp, -1, 1), calculate_lambda_kkt!,
    augmented_hess_lag_vec!'s diagonal, projcg!'s first Q'r; src/optimize.jl:312-343, src/inequality_helper.jl:144-158, 286-308)."""
    from lfpsqp_jl_amd.inequality import (InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_,
                                          inequality_gradient_)
    ctx = dev_ctx
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    idata = InequalityData(ctx, xl, xu)
    x = StackedVector(ctx, n)
    x.upload(0.6 * synth.hash_vector(2, n), 0)
    generate_initial_y_(x, idata)
    Jct = ctx.matrix(n, m, np.asfortranarray(synth.hash_matrix(3, n, m)))
    dec = InequalityDecomp(ctx, n, m, Jct, factored=True)
    inequality_gradient_(dec, x, idata)
    dh = synth.hash_vector(8, 2 * n)
    d1, d2 = StackedVector(ctx, n), StackedVector(ctx, n)
    d1.upload2(dh); d2.upload2(dh)
    # the statement sequence
    W = np.zeros((m, m), order='F')
    S0, Vt0, rank = L.ksvd_(Jct, None, w2=dec.sx, W=W)
    assert rank == m
    dec.Sigma[:] = S0; dec.Vt[:, :] = Vt0; dec.W = W; dec.rank = rank
    Q = InequalityDecompProject(dec)
    tw, tm = ctx.vector(n), ctx.vector(m)
    Q.mul_t(tw, tm, d1)
    Q.mul_n(d1, tw, tm, -1.0, 1.0)
    th = tm.download(m) / S0
    lam0 = Vt0.T @ th
    lamy0 = ctx.vector(n)
    lamdev = ctx.vector(m, lam0)
    ctx.check(ctx.L.lfpsqp_calculate_lambda_y(ctx.h, Jct.h, m, lamdev.h, dec.Dx.h, dec.S.h, tw.h, lamy0.h))
    hxh = 2.0 + synth.hash_vector(41, n) ** 2
    hx = ctx.vector(n, hxh)
    a0 = StackedVector(ctx, n)
    idc = idata._c()
    ctx.check(ctx.L.lfpsqp_augmented_diag(ctx.h, hx.h, lamy0.h, C.byref(idc), a0.h))
    r0 = StackedVector(ctx, n)
    L.waxpby(-1.0, d1, 0.0, d1, r0)
    tw2, tm2 = ctx.vector(n), ctx.vector(m)
    Q.mul_t(tw2, tm2, r0)
    # the fused pass
    e = ctx.vector(n)
    ctx.check(ctx.L.lfpsqp_ineq_rhs(ctx.h, d2.h, dec.Dx.h, dec.Dy.h, e.h))
    W2 = np.zeros((m, m), order='F')
    S1, Vt1, rank1, Jtd = L.ksvd_(Jct, None, w2=dec.sx, W=W2, rhs=e)
    np.testing.assert_array_equal(S1, S0)
    dec.W = W2
    work = L.ProjCGWork(ctx, n, m, n)
    a1, lamy1 = StackedVector(ctx, n), ctx.vector(n)
    Utd, lam = np.zeros(m), np.zeros(m)
    ss = C.c_double()
    bs, wc = Q._c(), work._c()
    ctx.check(ctx.L.lfpsqp_tangent_step(ctx.h, C.byref(bs), S1.ctypes.data, np.asfortranarray(Vt1).ctypes.data, m, Jtd.ctypes.data, None, d2.h, None, x.h,
                                        a1.h, C.byref(idc), hx.h, dec.S.h, lamy1.h, C.byref(wc), 0, Utd.ctypes.data, lam.ctypes.data, C.byref(ss)))
    scale = np.linalg.norm(dh)
    np.testing.assert_allclose(Utd, tm.download(m), atol=1e-12 * scale)
    np.testing.assert_allclose(lam, lam0, atol=1e-11 * scale / S0[-1])
    np.testing.assert_allclose(d2.download2(), d1.download2(), atol=1e-13 * scale)
    np.testing.assert_allclose(work.rp.download2(), -d1.download2(), atol=1e-13 * scale)
    np.testing.assert_allclose(lamy1.download(), lamy0.download(), rtol=1e-10, atol=1e-11 * scale)
    np.testing.assert_allclose(a1.download2(), a0.download2(), rtol=1e-10, atol=1e-10 * scale)
    np.testing.assert_allclose(work.Utr.download()[:m], tm2.download(m), atol=1e-12 * scale)
    np.testing.assert_allclose(ss.value, np.sum(d1.download2() ** 2), rtol=1e-12)


@pytest.mark.parametrize("n,m,view", [(1800, 16, False), (2600, 128, False), (2300, 24, True), (1500, 300, False)])
def test_tangent_step_as_the_initial_projection_of_projcg(dev_ctx, n, m, view):
    """LFPSQP_TANGENT_INIT_PROJCG + LFPSQP_PROJCG_START_PROJECTED: the tangent-step pass also makes projcg!'s initial projection
    (src/projcg.jl:58-62), with U'r0 = -(I - U'U) U'd from the Gram matrix of the factorisation.  The solve that follows is the solve from
    scratch: equal counts, iterates to 1e-11; for a plain matrix and for a view (streamed gradients with the quadratic term)."""
    ctx = dev_ctx
    if view:
        A0 = ctx.matrix(n, m, np.asfortranarray(2.0 ** -3 * synth.hash_matrix(21, n, m)))
        cons = L.ElementwiseConstraints(ctx, A0, np.zeros(m), kind=(np.arange(n) % 3).astype(np.float64), qw=1e-2 * np.cos(np.arange(m)), stream=True)
        xv = ctx.vector(n, 0.5 * synth.hash_vector(31, n))
        cons.jac_(cons.Jct, np.zeros(m), xv)
        Jct = cons.Jct
    else:
        cons, xv = None, None
        Jct = ctx.matrix(n, m, np.asfortranarray(synth.hash_matrix(3, n, m)))
    dh = synth.hash_vector(8, n)
    W = np.zeros((m, m), order='F')
    G = np.zeros((m, m), order='F')
    d1, d2 = ctx.vector(n, dh), ctx.vector(n, dh)
    S, Vt, rank, Jtd = L.ksvd_(Jct, None, W=W, rhs=d1, G_out=G)
    assert rank == m
    U = L.DeviceBasis(None, rank, generator=(Jct, W))
    base = 5.0 + 4.0 * synth.hash_vector(9, n)
    # (a) the tangent step, then projcg! from the given start (its own initial projection)
    w1 = L.ProjCGWork(ctx, n, m)
    h1 = ctx.vector(n, base)
    _tangent_step(ctx, U, S, Vt, m, Jtd, d1, cons, xv, h1, w1)
    x1 = ctx.vector(n)
    i1, nr1 = L.projcg_(x1, None, L.DiagOperator(0.0, h1), U, d1, None, tol=1e-9 * np.linalg.norm(dh), maxit=300, work=w1, want_lambda=False,
                        start_given=True)
    # (b) the tangent step that is the initial projection as well
    w2 = L.ProjCGWork(ctx, n, m)
    h2 = ctx.vector(n, base)
    _tangent_step(ctx, U, S, Vt, m, Jtd, d2, cons, xv, h2, w2, G=G)
    np.testing.assert_array_equal(d2.download(), d1.download())
    np.testing.assert_array_equal(h2.download(), h1.download())
    x2 = ctx.vector(n)
    i2, nr2 = L.projcg_(x2, None, L.DiagOperator(0.0, h2), U, d2, None, tol=1e-9 * np.linalg.norm(dh), maxit=300, work=w2, want_lambda=False,
                        start_projected=True)
    assert i2 == i1 and i1 > 3
    assert nr2 == pytest.approx(nr1, rel=1e-6)
    np.testing.assert_allclose(x2.download(), x1.download(), atol=1e-11 * np.linalg.norm(x1.download()))
