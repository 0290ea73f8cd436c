"""Staged stores of the one-pass kernels (csrc/kernels.h, onepass_kernel STG): the row update's output vector waits in LDS and the
workgroup stores it in a few bursts over its span.  WHEN a value is stored must not change WHAT is stored: with the rounds per burst
capped at 1, 2 or 3 (test hook LFPSQP_STAGE_ROUNDS, read when a context is created) every workgroup goes through many bursts and a
ragged last one, and the iterates have to be bit-identical to the uncapped run (one or two bursts per span), which the parity tests
compare with the oracle.  Covers every staged tile width (8 / 16 / 24 / 32 / 33 column groups per wave, and the wide form at 24 / 32) of the staged kernels:
the fused projected-CG iteration (src/projcg.jl:84-103), the pcg! iteration (src/retractions.jl:215-229) and the one-stream Newton
step (src/retractions.jl:141-148)."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import synth


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def _with_caps(ctx0, monkeypatch, run, caps=("", "1", "3")):
    out = []
    for cap in caps:
        if cap:
            monkeypatch.setenv("LFPSQP_STAGE_ROUNDS", cap)
        else:
            monkeypatch.delenv("LFPSQP_STAGE_ROUNDS", raising=False)
        ctx = L.Context(0, ctx0.L)
        out.append(run(ctx))
        ctx.close()
    monkeypatch.delenv("LFPSQP_STAGE_ROUNDS", raising=False)
    return out


@pytest.mark.parametrize("m", [20, 40, 90, 128, 130, 300, 500])
def test_projcg_iterates_do_not_depend_on_the_burst_length(dev_ctx, monkeypatch, m):
    emu = _is_emu(dev_ctx)
    n = (5003 if m < 200 else 1603) if emu else 400_003     # not a multiple of the 64-row round: the last round of the last workgroup is ragged

    def run(ctx):
        scale = 2.0 ** np.floor(np.log2(np.sqrt(3.0 / n)))
        Z = ctx.matrix(n, m).hash_fill(1, 0, n, scale)
        dg = ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)
        b = ctx.vector(n).hash_fill(4)
        x = ctx.vector(n)
        it, nr = L.projcg_(x, None, L.DiagOperator(0.0, dg), L.DeviceBasis(Z), b, None, tol=1e-300, maxit=9, want_lambda=False)
        return it, nr, x.download()

    res = _with_caps(dev_ctx, monkeypatch, run)
    it0, nr0, x0 = res[0]
    assert it0 == 9 and np.isfinite(nr0) and np.linalg.norm(x0) > 0
    for it, nr, x in res[1:]:
        assert it == it0 and nr == nr0
        np.testing.assert_array_equal(x, x0)


@pytest.mark.parametrize("m", [24, 100, 320])
def test_pcg_iterates_do_not_depend_on_the_burst_length(dev_ctx, monkeypatch, m):
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    emu = _is_emu(dev_ctx)
    n = (4001 if m < 200 else 1501) if emu else 300_001
    rng = np.random.default_rng(3 + m)
    Jt = np.asfortranarray(rng.standard_normal((n, m)))
    bh = rng.standard_normal(n)

    def run(ctx):
        Jd = ctx.matrix(n, m, Jt)
        w = L.ProjPenaltyWork(ctx, m, n, False)
        x, r = ctx.vector(n), ctx.vector(n, bh)
        flag, i = L.pcg_(1e-2, _JacPlain(Jd, w), L.no_precondition, x, r, w.p, w.z, None, 1e-30, 7)
        return flag, i, x.download(), r.download()

    res = _with_caps(dev_ctx, monkeypatch, run)
    assert res[0][1] == 7
    for flag, i, x, r in res[1:]:
        assert (flag, i) == res[0][:2]
        np.testing.assert_array_equal(x, res[0][2])
        np.testing.assert_array_equal(r, res[0][3])


@pytest.mark.parametrize("m_lin", [12, 60, 128, 300])
def test_newton_step_iterates_do_not_depend_on_the_burst_length(dev_ctx, monkeypatch, m_lin):
    emu = _is_emu(dev_ctx)
    n = (3001 if m_lin < 200 else 901) if emu else 300_001
    m = m_lin
    pert = 1e-2 * np.random.default_rng(5 + m).standard_normal(n)

    def run(ctx):
        Jct = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0, n, m_lin)
        xs_h = synth.hash_vector(2, n)
        b = Jct.download().T @ xs_h
        cons = L.DeviceConstraints(Jct, m_lin, b, False, 0.0, n, -1)
        xs = ctx.vector(n, xs_h)
        Z = ctx.matrix(n, m)
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(Jct, Z, W=W)
        assert rank == m
        xt, xnew = ctx.vector(n, xs_h + pert), ctx.vector(n)
        nr = L.NR(L.DeviceBasis(Z, generator=(Jct, W)), S, Vt, 1e-11, 50, L.NRWork(m), False, None)
        cval = np.zeros(m)
        flag, it, _ = L.retract_(cval, xnew, cons, xt, xs, nr)
        return flag, it, xnew.download(), cval.copy()

    res = _with_caps(dev_ctx, monkeypatch, run)
    assert res[0][0] == 0 and res[0][1] >= 1
    for flag, it, x, cv in res[1:]:
        assert (flag, it) == res[0][:2]
        np.testing.assert_array_equal(x, res[0][2])
        np.testing.assert_array_equal(cv, res[0][3])


@pytest.mark.parametrize("m", [12, 60, 128])
def test_stacked_projcg_iterates_do_not_depend_on_the_burst_length(dev_ctx, monkeypatch, m):
    """The stacked (bound-projected) forms stage TWO vectors, the x and the y half of the residual, in bursts half as long."""
    from lfpsqp_jl_amd.inequality import (InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_,
                                           inequality_gradient_)
    emu = _is_emu(dev_ctx)
    n = 3001 if emu else 200_001
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    xh = 0.6 * synth.hash_vector(2, n)
    a = 4.0 * synth.hash_vector(3, 2 * n) + 5.0
    bh = synth.hash_vector(4, 2 * n)

    def run(ctx):
        idata = InequalityData(ctx, xl, xu)
        xa = StackedVector(ctx, n)
        xa.upload(xh, 0)
        generate_initial_y_(xa, idata)
        Jct = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
        dec = InequalityDecomp(ctx, n, m, Jct)
        inequality_gradient_(dec, xa, idata)
        S, Vt, rank = L.ksvd_(Jct, dec.Z, w2=dec.sx)
        assert rank == m
        dec.rank = rank
        Q = InequalityDecompProject(dec)
        A = L.DiagOperator(0.0, StackedVector(ctx, n).upload2(a))
        b = StackedVector(ctx, n).upload2(bh)
        x = StackedVector(ctx, n)
        it, nr = L.projcg_(x, None, A, Q, b, None, tol=1e-300, maxit=7, work=L.ProjCGWork(ctx, 0, m, stacked_N=n), want_lambda=False)
        return it, nr, x.download2()

    res = _with_caps(dev_ctx, monkeypatch, run)
    assert res[0][0] == 7 and np.isfinite(res[0][1]) and np.linalg.norm(res[0][2]) > 0
    for it, nr, x in res[1:]:
        assert (it, nr) == res[0][:2]
        np.testing.assert_array_equal(x, res[0][2])


@pytest.mark.parametrize("m,nonlinear", [(40, False), (128, True), (130, False), (90, True), (40, "bounds"), (128, "bounds")])
def test_tangent_step_outputs_do_not_depend_on_the_burst_length(dev_ctx, monkeypatch, m, nonlinear):
    """The one-pass tangent step with projcg!'s initial projection folded in (lfpsqp_tangent_step, LFPSQP_TANGENT_INIT_PROJCG; src/optimize.jl:305-343,
    src/projcg.jl:55-62) stages THREE or FOUR vectors -- the projected step, g0, -g0 and, for a class with a constraint term in its Hessian,
    the completed diagonal -- and has two or three first products: the trajectory of an `optimize` run that goes through it every outer
    iteration must be bit-identical whatever the burst length."""
    emu = _is_emu(dev_ctx)
    n = 1537 if emu else 300_001
    rng = np.random.default_rng(7 + m)
    Ah = np.asfortranarray(rng.standard_normal((n, m)) / np.sqrt(n))
    target = 0.5 * rng.standard_normal(n)
    x0 = 0.2 * rng.standard_normal(n)

    def run(ctx):
        assert ctx.options.fused_tangent_step
        tr = []
        if nonlinear == "bounds":            # config 4's class: ball in slack form + four-way bounds -- the stacked form, NINE staged vectors
            P0 = synth.BallBoxProblem(n, m)
            Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
            prob = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
            xs = 0.9 * synth.hash_vector(2, n) + 0.1 * P0.x0
            x, obj, lam, ti = prob.optimize(xs, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3), trace=tr)
            return x, obj, [t['x'] for t in tr], [t.get('tn_iter') for t in tr]
        if nonlinear:
            kind = (np.arange(n) % 3).astype(np.float64)
            cons = L.ElementwiseConstraints(ctx, ctx.matrix(n, m, Ah), np.zeros(m), kind=kind)
            cv = np.zeros(m)
            cons.c_(cv, ctx.vector(n, x0))
            cons = L.ElementwiseConstraints(ctx, ctx.matrix(n, m, Ah), cv, kind=kind)            # x0 on the manifold
            prob = L.SeparableElementwiseBox(ctx, cons, 1, 0.3, target)
        else:
            prob = L.SeparableLinearBallBox(ctx, n, m, ctx.matrix(n, m, Ah), Ah.T @ x0, 1, 0.3, target)
        x, obj, lam, ti = prob.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3), trace=tr)
        return x, obj, [t['x'] for t in tr], [t.get('tn_iter') for t in tr]

    res = _with_caps(dev_ctx, monkeypatch, run, caps=("", "1", "2") if not emu else ("", "1"))
    x0_, obj0, xs0, tn0 = res[0]
    assert len(xs0) >= 3 and any((t or 0) >= 1 for t in tn0)
    for x, obj, xs, tn in res[1:]:
        assert tn == tn0
        np.testing.assert_array_equal(x, x0_)
        np.testing.assert_array_equal(obj, obj0)
        for a, b in zip(xs, xs0):
            np.testing.assert_array_equal(a, b)
