"""Row-scaled matrix views (lfpsqp_mat_rowscaled_view) and the STREAMED gradients of the nonlinear constraint class built on them.

The reference's jac! (src/autodiff_generators.jl:60-66, called at src/optimize.jl:284 and at every iteration of the ProjPenalty retraction,
src/retractions.jl:340) rewrites the whole n x m matrix Jct(x).  For constraints c(x) = A' phi(x) - b that matrix is diag(phi'(x)) A: with a
view, jac! rewrites one n-vector and the tangent setup, projcg!, both retractions and the multiplier products stream the constant A.  Checked
here: every product kernel on a view against numpy and against the materialised matrix; the refusals; the solvers on a view against the same
solvers on the materialised matrix and against the oracle; `optimize` streamed against `optimize` with a materialised Jct (same trajectory)."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .test_capi_parity import DiagOpRef
from .test_elementwise import _phi, _trace_compare, ew_callables


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def _close(a, b, tol):
    """|a - b| <= tol entry by entry (tol: array of rounding yardsticks)"""
    err = np.abs(np.asarray(a) - np.asarray(b))
    assert np.all(err <= tol), (float(np.max(err / np.maximum(tol, 1e-300))), "x the yardstick")


def _view(ctx, n, m, seed=1, spread=3.0, parts="both"):
    """A, the view diag(rs) A + u w' (parts: "scale", "rank1", "both") and the host matrix it stands for."""
    Ah = synth.hash_matrix(seed, n, m)
    rsh = np.exp(spread * synth.hash_vector(seed + 10, n)) * np.where(synth.hash_vector(seed + 11, n) > 0.3, -1.0, 1.0)   # both signs, 1e-1 .. 1e1
    rsh[::7] = 0.0                                                                     # (phi' vanishes somewhere)
    uh, wh = 2.0 * synth.hash_vector(seed + 12, n), 0.3 * synth.hash_vector(seed + 13, m)
    A = ctx.matrix(n, m, np.asfortranarray(Ah))
    rs = ctx.vector(n, rsh) if parts != "rank1" else None
    u, w = (ctx.vector(n, uh), ctx.vector(m, wh)) if parts != "scale" else (None, None)
    Mh = (rsh[:, None] * Ah if rs is not None else Ah.copy()) + (np.outer(uh, wh) if u is not None else 0.0)
    _view.mag = (np.abs(rsh[:, None] * Ah) if rs is not None else np.abs(Ah)) + (np.abs(np.outer(uh, wh)) if u is not None else 0.0)   # rounding yardstick
    return Ah, Mh, A, (rs, u, w), A.view(rs, u, w)


@pytest.mark.parametrize("parts", ["scale", "rank1", "both"])
@pytest.mark.parametrize("n,m", [(1, 1), (777, 5), (2100, 33), (1500, 128), (4100, 100), (900, 300)])
def test_product_kernels_on_a_view(dev_ctx, n, m, parts):
    """GEMV-T / GEMV-N (both forms of each), Gram (plain and weighted), the basis-forming product and the materialising copy of
    diag(rs) A + u w'."""
    ctx = dev_ctx
    Ah, Mh, A, (rs, u, w), V = _view(ctx, n, m, parts=parts)
    _close(V.download(), Mh, 4e-16 * _view.mag)                                        # (one multiply-add per entry)
    vh, th = synth.hash_vector(3, n), synth.hash_vector(4, m)
    v, t, y = ctx.vector(n, vh), ctx.vector(m, th), ctx.vector(n, vh)
    out = ctx.vector(m)
    L.gemv_t(V, v, out)
    _close(out.download(), Mh.T @ vh, 1e-12 * np.abs(Mh).T @ np.abs(vh) + 1e-300)
    L.gemv_n(V, t, y, alpha=-0.5, beta=2.0)
    _close(y.download(), -0.5 * (Mh @ th) + 2.0 * vh, 1e-12 * (np.abs(Mh) @ np.abs(th) + np.abs(vh)))
    if m > 1:                                                                          # a leading block of the columns
        L.gemv_t(V, v, out, ncols=m - 1)
        _close(out.download()[:m - 1], Mh[:, :m - 1].T @ vh, 1e-12 * np.abs(Mh[:, :m - 1]).T @ np.abs(vh) + 1e-300)
    scale = np.sqrt(np.outer(np.sum(Mh * Mh, 0), np.sum(Mh * Mh, 0))) + 1e-300
    assert np.max(np.abs(L.gram(V) - Mh.T @ Mh) / scale) < 1e-13
    w2h = synth.hash_vector(5, n) ** 2 + 0.1
    assert np.max(np.abs(L.gram(V, w2=ctx.vector(n, w2h)) - Mh.T @ (w2h[:, None] * Mh)) / (scale * 1.1)) < 1e-13
    r = min(m, 7)
    Wh = np.asfortranarray(np.random.default_rng(2).standard_normal((m, r)))
    Out = ctx.matrix(n, m)
    L.rmul(V, Wh, Out)
    _close(Out.download()[:, :r], Mh @ Wh, 1e-12 * (np.abs(Mh) @ np.abs(Wh)) + 1e-300)


def test_what_a_view_refuses(dev_ctx):
    ctx = dev_ctx
    Ah, Mh, A, (rs, u, w), V = _view(ctx, 600, 8)
    with pytest.raises(L.LfpsqpError):
        V.upload(Ah)
    with pytest.raises(L.LfpsqpError):
        V.hash_fill(3)
    with pytest.raises(L.LfpsqpError):
        L.rmul(A, np.eye(8, order='F'), V)                                            # as an output
    with pytest.raises(L.LfpsqpError):
        A.rowscaled_view(ctx.vector(599))                                             # scale vector too short
    with pytest.raises(L.LfpsqpError):
        A.view(None, u, None)                                                         # half a rank-one term
    with pytest.raises(L.LfpsqpError):
        A.view(None, None, None)                                                      # nothing to view
    with pytest.raises(L.LfpsqpError):
        V.rowscaled_view(rs)                                                          # no views of views
    cons = L.DeviceConstraints(V, 8, np.zeros(8))
    U = L.DeviceBasis(None, 8, generator=(V, np.eye(8, order='F')))
    nr = L.NR(U, np.ones(8), np.eye(8, order='F'), 1e-9, 10, L.NRWork(8), False, None)
    assert L.retract_nr_batch_width_(cons, nr) == 0                                   # the batched Newton retraction: one by one on a view
    with pytest.raises(L.LfpsqpError):                                                # the ball column would land in the borrowed storage
        c2 = L.DeviceConstraints(V, 7, np.zeros(7), has_ball=True, R2=1.0)
        c2.c_(np.zeros(8), ctx.vector(600))
    # the vectors are read at launch time: changing them changes the matrix
    rs.upload(np.full(600, 2.0))
    u.fill(0.0)
    np.testing.assert_array_equal(V.download(), 2.0 * Ah)


@pytest.mark.parametrize("n,m,cond", [(1500, 12, 1.0), (2100, 128, 1.0), (1800, 20, 1e6)])
def test_factorisation_of_a_view(dev_ctx, n, m, cond):
    """ksvd! (src/la_helper.jl:8-34) of diag(rs) A from the view against the same call on the materialised matrix: same rank, singular values
    and factor to rounding, with and without the basis, through the refinement rounds of an ill-conditioned block too."""
    ctx = dev_ctx
    Ah, Mh, A, (rs, u, w), V = _view(ctx, n, m, spread=1.0)
    if cond > 1.0:
        Ah = Ah * np.logspace(0, np.log10(cond), m)[None, :]
        A.upload(Ah)
        Mh = rs.download()[:, None] * Ah + np.outer(u.download(), w.download())
    M = ctx.matrix(n, m, np.asfortranarray(Mh))
    Zv, Zm = ctx.matrix(n, m), ctx.matrix(n, m)
    Wv, Wm, Wf = (np.zeros((m, m), order='F') for _ in range(3))
    Sv, Vtv, rv = L.ksvd_(V, Zv, W=Wv)
    Sm, Vtm, rm = L.ksvd_(M, Zm, W=Wm)
    Sf, Vtf, rf = L.ksvd_(V, None, W=Wf)
    assert rv == rm == rf == m
    np.testing.assert_allclose(Sv, Sm, rtol=1e-9 if cond > 1 else 1e-12)
    np.testing.assert_allclose(Sf, Sv, rtol=1e-9 if cond > 1 else 1e-12)
    Zh = Zv.download()
    np.testing.assert_allclose(Zh.T @ Zh, np.eye(m), atol=1e-10 if cond > 1 else 1e-12)      # an orthonormal basis ...
    np.testing.assert_allclose(Zh @ (Zh.T @ Mh), Mh, atol=1e-9 * np.abs(Mh).max())          # ... of the range of diag(rs) A
    np.testing.assert_allclose(Zh, Mh @ Wv, atol=1e-12 * max(1.0, cond))                      # Z = (diag(rs) A) W


@pytest.mark.parametrize("n,m,bounds", [(2100, 4, False), (2500, 33, False), (1500, 128, False), (1300, 24, True), (700, 300, False)])
def test_projcg_on_a_view_matches_the_materialised_matrix_and_the_oracle(dev_ctx, n, m, bounds):
    """projcg! (src/projcg.jl:40-121) with the basis U = (diag(rs) A) W in factored form over the VIEW: counts equal and iterates within 1e-10
    of the run over the materialised matrix and of the oracle; plain and bound-stacked bases."""
    ctx = dev_ctx
    Ah, Mh, A, (rs, u, w), V = _view(ctx, n, m, spread=1.0)
    M = ctx.matrix(n, m, np.asfortranarray(Mh))
    rng = np.random.default_rng(5)
    if not bounds:
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(V, None, W=W)
        Zh = Mh @ W
        a = 4.0 * synth.hash_vector(3, n) + 5.0
        bh = synth.hash_vector(4, n)
        Aop = L.DiagOperator(0.0, ctx.vector(n, a))
        b = ctx.vector(n, bh)
        tols = (1e-12,) if (_is_emu(ctx) and m >= 100) else (1e-6, 1e-12)      # (emulator: the wide shapes once)
        for ch in (None, rng.standard_normal(m)):
            for tol in tols:
                x0, l0 = np.zeros(n), np.zeros(m)
                i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Zh, bh, np.zeros(m) if ch is None else ch, tol=tol)
                for Mat in (V, M):
                    x, lam = ctx.vector(n), ctx.vector(m)
                    it, nr = L.projcg_(x, lam, Aop, L.DeviceBasis(None, m, generator=(Mat, W)), b, None if ch is None else ctx.vector(m, ch), tol=tol)
                    assert it == i0 and nr < tol
                    assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
                    assert np.abs(lam.download() - l0).max() < 1e-10
        return
    # bound-stacked basis (InequalityDecompProject, src/inequality_helper.jl:161-212) over the view: projections and projcg
    xl = np.where(np.arange(n) % 3 == 0, -1.0, -np.inf)
    xu = np.where(np.arange(n) % 3 == 1, 1.0, np.inf)
    idata = L.InequalityData(ctx, xl, xu)
    xh = 0.5 * synth.hash_vector(8, n)
    res = {}
    for tag, Mat in (("view", V), ("materialised", M)):
        xa = L.StackedVector(ctx, n)
        xa.upload(xh, 0)
        L.generate_initial_y_(xa, idata)
        idc = L.InequalityDecomp(ctx, n, m, Mat, factored=True)
        L.inequality_gradient_(idc, xa, idata)
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(Mat, None, w2=idc.sx, W=W)
        idc.W, idc.rank = W, rank
        Q = L.InequalityDecompProject(idc)
        hs = L.half_stride(ctx, n)
        gh = np.concatenate([synth.hash_vector(21, n), synth.hash_vector(22, n)])
        d = L.StackedVector(ctx, n)
        d.upload2(gh)
        tw, tm = ctx.vector(n), ctx.vector(m)
        Q.mul_t(tw, tm, d)
        Q.mul_n(d, tw, tm, -1.0, 1.0)
        proj = d.download2()
        a = L.StackedVector(ctx, n)
        a.upload2(np.concatenate([4.0 * synth.hash_vector(3, n) + 5.0, 3.0 * synth.hash_vector(6, n) + 4.0]))
        b = L.StackedVector(ctx, n)
        b.upload2(gh)
        x = L.StackedVector(ctx, n)
        it, nr = L.projcg_(x, None, L.DiagOperator(0.0, a), Q, b, None, tol=1e-10, work=L.ProjCGWork(ctx, n, m, n), n_global=2 * n, want_lambda=False)
        res[tag] = (rank, S, proj, it, x.download2())
        assert nr < 1e-10
    assert res["view"][0] == res["materialised"][0] == m
    np.testing.assert_allclose(res["view"][1], res["materialised"][1], rtol=1e-12)
    np.testing.assert_allclose(res["view"][2], res["materialised"][2], atol=1e-11)
    assert res["view"][3] == res["materialised"][3]
    np.testing.assert_allclose(res["view"][4], res["materialised"][4], atol=1e-10 * max(1.0, np.linalg.norm(res["materialised"][4])))


def _mixed(ctx, n, m, seed, stream, quad):
    """mixed kinds, with (quad) or without the common quadratic term"""
    rng = np.random.default_rng(seed)
    Ah = rng.standard_normal((n, m)) / np.sqrt(n)
    kind = rng.integers(0, 3, n).astype(np.float64)
    bh = rng.standard_normal(m) * 0.1
    qw = 0.01 * rng.standard_normal(m) if quad else None
    cons = L.ElementwiseConstraints(ctx, ctx.matrix(n, m, np.asfortranarray(Ah)), bh, kind=kind, qw=qw, stream=stream)
    return cons, Ah, kind, qw, bh, rng


@pytest.mark.parametrize("quad", [False, True])
def test_retractions_with_streamed_gradients(dev_ctx, quad):
    """retract!(::NR) (src/retractions.jl:75-177) and retract!(::ProjPenalty) (:266-440, without and with the exact preconditioner) of a
    mixed-kind system away from phi' = 1, streamed against materialised gradients and against the oracle: flags and counts equal, cval bit
    for bit c!(xnew), iterates within 1e-10."""
    ctx = dev_ctx
    n, m = (400, 16) if _is_emu(ctx) else (3000, 64)
    out = {}
    for stream in (True, False):
        cons, Ah, kind, qw, bh, rng = _mixed(ctx, n, m, 31, stream, quad)
        assert cons.streamed == stream
        x0 = 0.4 * rng.standard_normal(n)
        c_, jac_, _ = ew_callables(Ah, kind, qw, bh)
        cv0 = np.zeros(m)
        c_(cv0, x0)
        cons.b = cons.b + cv0                                                         # x0 feasible
        bh = bh + cv0
        c_, jac_, _ = ew_callables(Ah, kind, qw, bh)
        x = ctx.vector(n, x0)
        cv = np.zeros(m)
        cons.jac_(cons.Jct, cv, x)
        Jh = _phi(kind, x0, 1)[:, None] * Ah + (np.outer(2.0 * x0, qw) if quad else 0.0)
        np.testing.assert_allclose(cons.Jct.download(), Jh, rtol=1e-14, atol=1e-17)       # (the device's own cos)
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(cons.Jct, None, W=W)
        assert rank == m
        Zh = Jh @ W
        step = rng.standard_normal(n)
        step -= Zh @ (Zh.T @ step)
        step *= 0.05 / np.linalg.norm(step)
        xt_h = x0 + step
        xtilde, xnew = ctx.vector(n, xt_h), ctx.vector(n)
        res = []
        nr0 = R.NR(Zh, S, Vt, 1e-9, 100, R.NRWork(m), False, R.InequalityData())
        xn0, cvr = np.zeros(n), np.zeros(m)
        f0, i0, _ = R.retract_(cvr, xn0, c_, xt_h, x0, nr0)
        cval, cval2 = np.zeros(m), np.zeros(m)
        nr = L.NR(L.DeviceBasis(None, m, generator=(cons.Jct, W)), S, Vt, 1e-9, 100, L.NRWork(m), False, None)
        flag, it, _ = L.retract_(cval, xnew, cons, xtilde, x, nr)
        cons.c_(cval2, xnew)
        np.testing.assert_array_equal(cval, cval2)
        assert (flag, it) == (f0, i0)
        np.testing.assert_allclose(xnew.download(), xn0, atol=1e-10 * max(1.0, np.linalg.norm(xn0)))
        res.append(("NR", flag, it, 0, xnew.download()))
        pp0 = R.ProjPenalty(jac_, Zh, S, Vt, m, 0.01, 1e-9, 100, 200, R.ProjPenaltyWork(m, n, m, n), False,
                            R.InequalityDecomp(np.zeros((0, 0)), *(np.zeros(0) for _ in range(5)), np.zeros((0, 0)), 0), R.InequalityData())
        f0, i0, p0 = R.retract_(cvr, xn0, c_, xt_h, x0, pp0)
        for precond in (False, True):
            idc = L.InequalityDecomp(ctx, n, m, cons.Jct, factored=True)
            ctx.options.pp_precondition = precond
            wk = L.ProjPenaltyWork(ctx, m, n, False)
            ctx.options.pp_precondition = False
            pp = L.ProjPenalty(cons.jac_, None, S, Vt, m, 0.01, 1e-9, 100, 200, wk, False, idc, None)
            flag, it, pcg_i = L.retract_(cval, xnew, cons, xtilde, x, pp)
            assert flag == 0 and np.max(np.abs(cval)) < 1e-9
            if not precond:
                assert (flag, it) == (f0, i0) and abs(pcg_i - p0) <= 2
                np.testing.assert_allclose(xnew.download(), xn0, atol=1e-10 * max(1.0, np.linalg.norm(xn0)))
            res.append(("PP-pre" if precond else "PP", flag, it, pcg_i, xnew.download()))
            cons.jac_(cons.Jct, cv, x)                                                # (ProjPenalty leaves the gradients at its last iterate)
        out[stream] = res
    for (na, fa, ia, pa, xa), (nb, fb, ib, pb, xb) in zip(out[True], out[False]):
        assert (na, fa, ia) == (nb, fb, ib) and abs(pa - pb) <= 2, (na, (fa, ia, pa), (fb, ib, pb))
        np.testing.assert_allclose(xa, xb, atol=1e-10 * max(1.0, np.linalg.norm(xb)))


@pytest.mark.parametrize("bounds,project,quad", [(False, False, False), (True, False, True), (False, True, True), (True, True, False), (False, False, True)])
def test_optimize_with_streamed_gradients_against_materialised_ones_and_the_oracle(dev_ctx, bounds, project, quad):
    """optimize (src/optimize.jl:119) of f = |x - target|^2 on a mixed-kind system: the streamed run, the run with a materialised Jct and the
    oracle's run with host callables walk the same trajectory (counts, step types, retraction iterations, iterates to 1e-10)."""
    ctx = dev_ctx
    n, m = (240, 12) if _is_emu(ctx) else (10000, 64)
    runs = {}
    for stream in (True, False):
        cons, Ah, kind, qw, bh, rng = _mixed(ctx, n, m, 41, stream, quad)
        target = 0.5 * rng.standard_normal(n)
        x0 = 0.2 * rng.standard_normal(n)
        # start on the manifold: from an infeasible start the first accepted Newton retractions take hundreds of Broyden iterations at n = 1e4, and
        # such a path amplifies rounding differences to 4e-8 in BOTH device runs alike (tools/ew_stream_diffs.py prints the deviations per iteration)
        cv0 = np.zeros(m)
        ew_callables(Ah, kind, qw, bh)[0](cv0, x0)
        bh = bh + cv0
        cons.b = cons.b + cv0
        xl = xu = None
        if bounds:
            target = np.clip(target, -0.8, 0.8)
            xl = np.where(np.arange(n) % 4 == 1, -1.0, np.where(np.arange(n) % 4 == 3, -1.0, -np.inf))
            xu = np.where(np.arange(n) % 4 == 2, 1.0, np.where(np.arange(n) % 4 == 3, 1.0, np.inf))
        prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target, xl=xl, xu=xu)
        tr = []
        xd, obj, lam, ti = prob.optimize(x0, L.LFPSQPParams(do_project_retract=project, maxiter=10, disp=L.DisplayOption.off), trace=tr)
        assert cons.streamed == stream
        runs[stream] = (tr, xd, obj, ti)
    (tr1, x1, o1, t1), (tr0, x0_, o0, t0) = runs[True], runs[False]
    assert t1.iter == t0.iter and t1.condition.name == t0.condition.name
    _trace_compare(tr1, tr0)
    np.testing.assert_allclose(o1, o0, rtol=1e-10)
    # the oracle
    c_, jac_, hdiag = ew_callables(Ah, kind, qw, bh)
    f = lambda xx: float(np.sum((xx[:n] - target) ** 2))

    def grad_(g, xx):
        g[:n] = 2.0 * (xx[:n] - target)

    def hlv_(dest, src, xx, lam_):
        dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]

    trr = []
    xr, objr, lamr, tir = R.optimize_core(f, grad_, c_, jac_, hlv_, x0, xl, xu, m,
                                          R.LFPSQPParams(do_project_retract=project, maxiter=10, disp=R.DisplayOption.off), trace=trr)
    assert t1.iter == tir.iter and t1.condition.name == tir.condition.name
    _trace_compare(tr1, trr)
    np.testing.assert_allclose(x1, xr, atol=1e-10 * max(1.0, np.linalg.norm(xr)))
