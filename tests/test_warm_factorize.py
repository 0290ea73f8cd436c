"""Warm start of the tangent setup's small eigenproblem (lfpsqp_factorize_hint).

The reference factorises the constraint gradients afresh at every outer iteration (ksvd!, src/la_helper.jl:8-34 at src/optimize.jl:286-302).
Between outer iterations they move little -- with linear constraints and no bounds not at all -- so the eigenvectors of the previous Gram matrix
almost diagonalise the next one: the one-sided Jacobi starts from L' V_prev instead of L.  The hint is an optimisation only: same Sigma, rank and
column space; the basis may rotate inside clusters of equal singular values (every use is invariant to that); a hint that is not an orthogonal
matrix of the right size is ignored."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import synth


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def _check_factors(Mh, S, Vt, rank, Zh, W, w2h=None):
    n, m = Mh.shape
    assert rank == m
    sw = np.ones(n) if w2h is None else np.sqrt(w2h)
    np.testing.assert_allclose(S, np.linalg.svd(sw[:, None] * Mh, compute_uv=False), rtol=1e-11)
    if Zh is not None:
        np.testing.assert_allclose((sw[:, None] * Zh).T @ (sw[:, None] * Zh), np.eye(m), atol=1e-12)
        np.testing.assert_allclose(Zh, Mh @ W, atol=1e-12)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(m), atol=1e-12)
    # diag(sqrt(w2)) Jct = U S Vt with U = diag(sqrt(w2)) Z = diag(sqrt(w2)) Jct W
    np.testing.assert_allclose((sw[:, None] * (Mh @ W)) * S @ Vt, sw[:, None] * Mh, atol=1e-11 * np.abs(Mh).max() * np.sqrt(n))


@pytest.mark.parametrize("n,m,weighted", [(1500, 12, False), (2100, 128, False), (1900, 130, True), (1700, 64, True), (900, 200, False)])
def test_warm_factorisation_gives_the_cold_factors(dev_ctx, n, m, weighted):
    ctx = dev_ctx
    Mh = synth.hash_matrix(1, n, m)
    M = ctx.matrix(n, m, np.asfortranarray(Mh))
    w2h = synth.hash_vector(5, n) ** 2 + 0.1 if weighted else None
    w2 = ctx.vector(n, w2h) if weighted else None
    Z = ctx.matrix(n, m)
    Wc, Ww, Wp = (np.zeros((m, m), order='F') for _ in range(3))
    Sc, Vtc, rc = L.ksvd_(M, Z, w2=w2, W=Wc)
    Zc = Z.download()
    _check_factors(Mh, Sc, Vtc, rc, Zc, Wc, w2h)
    # the same matrix again, warm: the Gram matrix is already diagonal in the hinted basis
    Sw, Vtw, rw = L.ksvd_(M, Z, w2=w2, W=Ww, Vt_prev=Vtc)
    Zw = Z.download()
    _check_factors(Mh, Sw, Vtw, rw, Zw, Ww, w2h)
    np.testing.assert_allclose(Sw, Sc, rtol=1e-12)
    sw = np.ones(n) if w2h is None else np.sqrt(w2h)
    Pc = lambda v: Zc @ (Zc.T @ ((sw ** 2) * v))
    Pw = lambda v: Zw @ (Zw.T @ ((sw ** 2) * v))
    v = synth.hash_vector(9, n)
    np.testing.assert_allclose(Pw(v), Pc(v), atol=1e-11 * np.linalg.norm(v))        # the same (weighted) projector
    # a nearby matrix (the next outer iteration's gradients), warm from the old eigenvectors, factored form
    Mh2 = Mh + 1e-3 * synth.hash_matrix(7, n, m)
    M2 = ctx.matrix(n, m, np.asfortranarray(Mh2))
    Sp, Vtp, rp = L.ksvd_(M2, None, w2=w2, W=Wp, Vt_prev=Vtc)
    _check_factors(Mh2, Sp, Vtp, rp, None, Wp, w2h)


def test_a_hint_that_is_not_an_orthogonal_matrix_is_ignored(dev_ctx):
    ctx = dev_ctx
    n, m = 1300, 24
    Mh = synth.hash_matrix(1, n, m)
    M = ctx.matrix(n, m, np.asfortranarray(Mh))
    Wc, W1, W2, W3 = (np.zeros((m, m), order='F') for _ in range(4))
    Sc, Vtc, rc = L.ksvd_(M, None, W=Wc)
    bad = np.asfortranarray(np.random.default_rng(3).standard_normal((m, m)))
    S1, Vt1, r1 = L.ksvd_(M, None, W=W1, Vt_prev=bad)                                  # not orthogonal
    rankdef = Vtc.copy(order='F')
    rankdef[m - 1, :] = 0.0                                                            # the Vt of a rank-deficient factorisation
    S2, Vt2, r2 = L.ksvd_(M, None, W=W2, Vt_prev=rankdef)
    S3, Vt3, r3 = L.ksvd_(M, None, W=W3, Vt_prev=np.eye(m + 1, order='F'))             # another size
    for S_, Vt_, W_ in ((S1, Vt1, W1), (S2, Vt2, W2), (S3, Vt3, W3)):
        np.testing.assert_array_equal(S_, Sc)                                          # bit for bit the cold call
        np.testing.assert_array_equal(Vt_, Vtc)
        np.testing.assert_array_equal(W_, Wc)
    # a hint is consumed by the first factorisation that follows: the call after a warm one is cold again
    ctx.check(ctx.L.lfpsqp_factorize_hint(ctx.h, np.asfortranarray(Vtc).ctypes.data, m))
    L.ksvd_(M, None, W=W1)
    S4, Vt4, r4 = L.ksvd_(M, None, W=W2)
    np.testing.assert_array_equal(Vt4, Vtc)


@pytest.mark.parametrize("bounds", [False, True])
def test_optimize_is_the_same_with_and_without_the_warm_start(dev_ctx, bounds):
    """optimize (src/optimize.jl:119) on config 3 / config 4's shape with DeviceOptions.warm_factorize on (default) and off: same counts, step
    types, accepted steps; iterates to 1e-10."""
    ctx = dev_ctx
    n, m = (400, 8) if _is_emu(ctx) else (20000, 64)
    res = {}
    for warm in (True, False):
        ctx.options.warm_factorize = warm
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
        res[warm] = (tr, x, obj, ti)
    ctx.options.warm_factorize = True
    (tr1, x1, o1, t1), (tr0, x0_, o0, t0) = res[True], res[False]
    assert t1.iter == t0.iter and t1.condition.name == t0.condition.name and len(tr1) == len(tr0)
    for k, (a, b_) in enumerate(zip(tr1, tr0)):
        nb = max(np.linalg.norm(b_["x"]), 1.0)
        assert np.linalg.norm(a["x"] - b_["x"]) <= 1e-10 * nb, (k, np.linalg.norm(a["x"] - b_["x"]) / nb)
        for key in ("steptype", "mtype", "retract_iter1", "alpha", "ls_flag", "tn_iter", "rank"):
            assert a.get(key) == b_.get(key), (k, key, a.get(key), b_.get(key))
    np.testing.assert_allclose(o1, o0, rtol=1e-10)
