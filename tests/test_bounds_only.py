"""retract!(::YRetract) (reference src/retractions.jl:67-72) and the bounds-only branch of the driver (src/optimize.jl:404-411: m == 0 with
bounds present -> YRetract; the 6-argument surface optimize(f, c!, x0, xl, xu, m) of src/optimize.jl:88 with m = 0) against the oracle,
on the reference test's four-way bound pattern (test/test_inequalities.jl:6-9: none / lower / upper / both)."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .test_capi_retractions import _compare_traces, _note


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def _four_way(n):
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    return xl, xu


def test_yretract_is_copy_then_y_retract(dev_ctx):
    """retract!(cval, xnew, c!, xtilde, x, ::YRetract): xnew = xtilde, y_retract!(xnew, x, idata), returns (0, 0, 0); x and xtilde untouched."""
    ctx = dev_ctx
    n = 3000 if not _is_emu(ctx) else 700
    xl, xu = _four_way(n)
    idata0 = R.InequalityData(xl, xu)
    idata = L.InequalityData(ctx, xl, xu)
    xaug = np.zeros(2 * n)
    xaug[:n] = 0.9 * synth.hash_vector(2, n)                          # inside the bounds
    R.generate_initial_y_(xaug, idata0)
    X = L.StackedVector(ctx, n).upload2(xaug)
    step = 0.3 * synth.hash_vector(3, 2 * n)                          # a step off the manifold: parabola and circle branches do real work
    xt0 = xaug + step
    Xt = L.StackedVector(ctx, n).upload2(xt0)
    Xn = L.StackedVector(ctx, n)
    xn0 = np.zeros(2 * n)
    out0 = R.retract_(np.zeros(0), xn0, None, xt0, xaug, R.YRetract(idata0))
    out = L.retract_(np.zeros(0), Xn, None, Xt, X, L.YRetract(idata))
    assert out == out0 == (0, 0, 0)
    got = Xn.download2()
    assert np.linalg.norm(got - xn0) <= 1e-10 * np.linalg.norm(xn0)
    np.testing.assert_allclose(got, xn0, rtol=1e-13, atol=1e-13)      # (measured: last-bit agreement)
    h = ctx.vector(n)
    assert L.calculate_h_(h, Xn, idata) < 1e-11                       # the new point is on the bound manifold
    np.testing.assert_array_equal(X.download2(), xaug)
    np.testing.assert_array_equal(Xt.download2(), xt0)
    # free variables (line): x = y after the retraction (src/retractions.jl:459-461)
    free = np.isinf(xl) & np.isinf(xu)
    assert np.array_equal(got[:n][free], got[n:][free])


def _weighted_quadratic(n, scale):
    tgt = scale * synth.hash_vector(5, n)
    a = 1.0 + 4.5 * (synth.hash_vector(8, n) + 1.0)                    # Hessian diag in [2, 20]: several truncated-Newton iterations per step
    f = lambda x: float(np.sum(a * (x - tgt) ** 2))

    def grad_(g, x):
        g[:] = 2 * a * (x - tgt)

    def hlv_(dest, src, x, lam):
        dest[:] = 2 * a * src
    return f, grad_, hlv_, tgt


def test_bounds_only_driver_follows_the_oracle(dev_ctx):
    """optimize(f, c!, x0, xl, xu, 0): no equalities, bounds present -- the driver's YRetract branch (src/optimize.jl:404-411), the tangent
    setup without a Jacobian, projcg! over the doubled variables with the augmented Hessian (src/inequality_helper.jl:144-158).  The optimum
    is strictly inside the box (no bound active at the solution): strict parity -- equal counts, step types, accepted steps, 1e-10 on every
    iterate."""
    ctx = dev_ctx
    n = 4000 if not _is_emu(ctx) else 400
    xl, xu = _four_way(n)
    f, grad_, hlv_, tgt = _weighted_quadratic(n, 0.8)
    x0 = 0.5 * synth.hash_vector(6, n)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(f, None, x0, xl, xu, 0, R.LFPSQPParams(disp=R.DisplayOption.off),
                                     derivatives=R.Derivatives(grad_, hlv_), trace=tr0)
    x, obj, lam, ti = L.optimize(f, None, x0, xl, xu, 0, L.LFPSQPParams(disp=L.DisplayOption.off),
                                 derivatives=L.Derivatives(grad_, hlv_), ctx=ctx, trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter and ti.iter >= 3
    assert all(t.get('mtype') == 0 and t.get('retract_iter1') == 0 for t in tr[:-1])        # YRetract reports (0, 0, 0)
    assert _compare_traces(tr, tr0) is None
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(obj, objr, rtol=1e-10, atol=1e-12 * objr[0])      # (the objective ends at 1e-14: absolute floor)
    assert len(lam) == len(lamr) == 0
    assert np.all(x >= xl - 1e-9) and np.all(x <= xu + 1e-9)
    assert np.linalg.norm(x - tgt) <= 1e-3 * np.linalg.norm(tgt)                             # (the unconstrained minimiser, inside the box)


def test_bounds_only_driver_with_active_bounds(dev_ctx):
    """The same with the minimiser OUTSIDE the box for a fifth of the variables.  Squared slacks make that solution degenerate (y -> 0 on an
    active bound, the reduced Hessian loses rank), and the trajectory becomes sensitive to the last bit well before it converges: the ORACLE
    started one ulp away from x0 departs from itself by 1e-10 ... 1e-9 after five outer iterations.  So: equal counts, step types and
    accepted steps throughout; 1e-10 on the iterates while the oracle's own one-ulp sensitivity stays below 1e-12, and ten times that
    sensitivity afterwards (the note prints both)."""
    ctx = dev_ctx
    n = 4000 if not _is_emu(ctx) else 400
    xl, xu = _four_way(n)
    f, grad_, hlv_, tgt = _weighted_quadratic(n, 1.5)
    x0 = 0.5 * synth.hash_vector(6, n)
    tr0, tr1, tr = [], [], []
    p0 = R.LFPSQPParams(disp=R.DisplayOption.off)
    xr, objr, lamr, tir = R.optimize(f, None, x0, xl, xu, 0, p0, derivatives=R.Derivatives(grad_, hlv_), trace=tr0)
    R.optimize(f, None, np.nextafter(x0, np.inf), xl, xu, 0, p0, derivatives=R.Derivatives(grad_, hlv_), trace=tr1)
    x, obj, lam, ti = L.optimize(f, None, x0, xl, xu, 0, L.LFPSQPParams(disp=L.DisplayOption.off),
                                 derivatives=L.Derivatives(grad_, hlv_), ctx=ctx, trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter and len(tr) == len(tr0)
    sens = [np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x']) for a, b in zip(tr1, tr0)] + [np.inf] * (len(tr0) - len(tr1))
    dev = [np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x']) for a, b in zip(tr, tr0)]
    _note("bounds only, active bounds: deviation per outer iteration " + " ".join(f"{v:.1e}" for v in dev)
          + " | the oracle's own one-ulp sensitivity " + " ".join(f"{v:.1e}" for v in sens))
    for k, (a, b) in enumerate(zip(tr, tr0)):
        for key in ('tn_iter', 'steptype', 'mtype', 'retract_iter1', 'alpha', 'ls_flag', 'rank'):
            assert a.get(key) == b.get(key), (k, key, a.get(key), b.get(key))
        assert dev[k] <= max(1e-10, 10.0 * max(sens[:k + 1])), (k, dev[k], sens[k])
    strict = [k for k in range(len(dev)) if max(sens[:k + 1]) < 1e-11]          # (n = 4000 on the GPU: the oracle's sensitivity is 1.5e-12 after ONE iteration)
    assert len(strict) >= 2 and all(dev[k] <= 1e-10 for k in strict)
    active = np.sum(np.abs(x - xl) < 1e-5) + np.sum(np.abs(x - xu) < 1e-5)
    assert active >= n // 10
    assert abs(obj[-1] - objr[-1]) <= max(1e-9, 20.0 * max(sens)) * abs(objr[-1])      # (f is quadratic in an iterate known to `sens`)
    # (on an active bound y -> 0 and the iterate sits within ~1e-8 of the bound, on either side -- the oracle's 3e-9 / 7e-9 outside, like the device's)
    assert np.all(x >= xl - 1e-7) and np.all(x <= xu + 1e-7)
    assert np.max(xl - xr) < 1e-7 and np.max(xr - xu) < 1e-7
