"""Committed golden vectors (tests/golden/oracle_golden.json, provenance in make_golden.py):
the oracle must keep reproducing them (CPU), and the device path must match them without consulting
the oracle at run time (`-m gpu`; also run on the emulator build)."""
import json
import os

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .helpers import DiagOpRef
from .test_oracle_reference_properties import rosenbrock

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.json")))
OFF0 = dict(disp=R.DisplayOption.off)


def _check(out, tr, gold, rel=1e-10, pcg_slack=0):
    x, obj, lam, ti = out
    assert ti.iter == gold["iters"] and ti.condition.name == gold["condition"]
    if rel > 1e-10:        # a check looser than north_star's 1e-10 says what it measured
        import warnings
        wn = max((abs(np.linalg.norm(t["x"]) - g["x_norm"]) / g["x_norm"] for t, g in zip(tr, gold["trace"])), default=0.0)
        wo = float(np.max(np.abs(np.asarray(obj) - np.asarray(gold["obj_values"])) / np.maximum(np.abs(gold["obj_values"]), 1e-300)))
        msg = f"[golden parity] worst relative deviation: |x_k| {wn:.2e}, f_k {wo:.2e} (tolerance {rel:g})"
        print(msg)
        warnings.warn(msg)
    np.testing.assert_allclose(obj, gold["obj_values"], rtol=rel, atol=1e-13)
    np.testing.assert_allclose(lam, gold["lam"], rtol=1e-7, atol=1e-10)
    assert np.linalg.norm(x) == pytest.approx(gold["x_norm"], rel=rel)
    np.testing.assert_allclose(x[:8], gold["x_head"], rtol=1e-8, atol=1e-10)
    for t, g in zip(tr, gold["trace"]):
        for k in ("tn_iter", "steptype", "mtype", "retract_iter1", "ls_flag", "rank"):
            assert t.get(k) == g[k], (k, t.get(k), g[k])
        if g["retract_iter2"] is not None:
            assert abs(t["retract_iter2"] - g["retract_iter2"]) <= pcg_slack
        assert t.get("alpha") == g["alpha"]
        assert np.linalg.norm(t["x"]) == pytest.approx(g["x_norm"], rel=rel)


def test_oracle_reproduces_goldens():
    f, dv = rosenbrock()
    tr = []
    _check(R.optimize(f, np.zeros(2), R.LFPSQPParams(**OFF0), derivatives=dv, trace=tr), tr, G["config1_rosenbrock"], rel=1e-12)
    for tag, dpr in (("nr", False), ("pp", True)):
        prob, x0 = synth.config3(2000, 8)
        tr = []
        out = R.optimize(prob.f, prob.grad_, prob.c_, prob.jac_, prob.hess_lag_vec_, x0, None, None, 8,
                         R.LFPSQPParams(do_project_retract=dpr, **OFF0), trace=tr)
        _check(out, tr, G[f"config3_n2000_m8_{tag}"], rel=1e-12)
    P = synth.BallBoxProblem(400, 6)
    tr = []
    out = R.optimize(P.f, P.c_, P.d_, P.x0, P.xl, P.xu, P.m, P.p, R.LFPSQPParams(do_project_retract=False, **OFF0),
                     derivatives=P.derivatives(), trace=tr)
    _check(out, tr, G["config4_n400_m6_nr"], rel=1e-12)


def test_device_projcg_matches_golden(dev_ctx):
    ctx = dev_ctx
    n, m = 1000, 10
    U, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
    Ub = L.DeviceBasis(ctx.matrix(n, m, np.asfortranarray(U)))
    A = L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0))
    b = ctx.vector(n).hash_fill(4)
    for key, g in G["projcg_n1000_m10"].items():
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, A, Ub, b, None, tol=float(key[3:]))
        assert it == g["iters"] and nr == pytest.approx(g["nr"], rel=1e-6)
        assert L.nrm2(x) == pytest.approx(g["x_norm"], rel=1e-12)
        np.testing.assert_allclose(x.download()[:8], g["x_head"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(lam.download(), g["lam"], atol=1e-11)


@pytest.mark.parametrize("tag,dpr", [("nr", False), ("pp", True)])
def test_device_config2_and_3_match_golden(dev_ctx, tag, dpr):
    ctx = dev_ctx
    par = L.LFPSQPParams(do_project_retract=dpr, disp=L.DisplayOption.off)
    J = np.zeros((50, 1), order='F')
    J[0, 0] = 1.0
    tr = []
    out = L.QuadLinearBallBox(ctx, 50, 1, ctx.matrix(50, 1, J), np.array([0.75])).optimize(np.ones(50), par, trace=tr)
    _check(out, tr, G[f"config2_n50_{tag}"], pcg_slack=2)
    n, m = 2000, 8
    Jct = ctx.matrix(n, m).hash_fill(1)
    bvec = ctx.vector(m)
    L.gemv_t(Jct, ctx.vector(n).hash_fill(2), bvec)
    tr = []
    out = L.QuadLinearBallBox(ctx, n, m, Jct, bvec.download()).optimize(np.ones(n), par, trace=tr)
    _check(out, tr, G[f"config3_n2000_m8_{tag}"], pcg_slack=2)


@pytest.mark.gpu
def test_device_config4_matches_golden():
    """~2300 Newton-retraction iterations: GPU only (the emulator runs config 4 in test_capi_retractions)."""
    ctx = L.Context(0)
    n, m = 400, 6
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    xs = ctx.vector(n + 1).hash_fill(2)
    ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
    bvec = ctx.vector(m + 1)
    L.gemv_t(Jct, xs, bvec, ncols=m)
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    tr = []
    out = L.QuadLinearBallBox(ctx, n, m, Jct, bvec.download()[:m], R2=n / 2.0, xl=xl, xu=xu).optimize(
        0.5 * np.ones(n), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off), trace=tr)
    _check(out, tr, G["config4_n400_m6_nr"])         # (1e-10; measured on the GPU: |x_k| 7.1e-13, f_k 2.6e-11)
    ctx.close()
