"""The device-resident NONLINEAR constraint class (lfpsqp_elementwise) against the oracle:
c(x) = A' phi(x) + qw x'x - b with phi elementwise from {t, sin t, t^2}.  The reference's own nonlinear test systems are of
this form -- the sin system (test/test_retractions.jl:34-54) and the sphere system (:1-31) -- and the tests below are the
reference's retraction tests (:90-103 Newton, :144-157 ProjPenalty) with c! / jac! resident on the device, plus `optimize`
end to end (src/optimize.jl:119) against the oracle's run with host callables of the same functions."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R

from .test_oracle_reference_properties import sin_system


def _is_emu(ctx):
    return "emulator" in ctx.device_name


# ---- numpy statement of the class (the oracle side's user callbacks) --------------------------------------------------------------
def _phi(kind, x, order=0):
    k0, k1 = kind == 0, kind == 1
    if order == 0:
        return np.where(k0, x, np.where(k1, np.sin(x), x * x))
    if order == 1:
        return np.where(k0, 1.0, np.where(k1, np.cos(x), 2.0 * x))
    return np.where(k0, 0.0, np.where(k1, -np.sin(x), 2.0))


def ew_callables(A, kind, qw, b):
    """(c!, jac!, hessian-diagonal) of c(x) = A' phi(x) + qw x'x - b on host arrays; A dense (n, m)."""
    n, m = A.shape
    kind = np.zeros(n) if kind is None else kind
    qw = np.zeros(m) if qw is None else qw

    def c_(cval, x):
        cval[:m] = A.T @ _phi(kind, x[:n]) + qw * (x[:n] @ x[:n]) - b

    def jac_(J, cval, x):
        c_(cval, x)
        J[:m, :n] = (_phi(kind, x[:n], 1)[:, None] * A).T + np.outer(qw, 2.0 * x[:n])

    def hdiag(x, lam):
        return _phi(kind, x[:n], 2) * (A @ lam[:m]) + 2.0 * (qw @ lam[:m])

    return c_, jac_, hdiag


def ew_test_data(n, m, seed=17):
    """A dense mixed-kind system with the common quadratic term, an objective target and a start -- shared with the 2-rank worker
    (tests/mp_worker.py), which takes the row shards of the same arrays."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, m)) / np.sqrt(n)
    kind = rng.integers(0, 3, n).astype(np.float64)
    qw = rng.standard_normal(m) * 0.01
    b = rng.standard_normal(m) * 0.1
    return A, kind, qw, b, 0.5 * rng.standard_normal(n), 0.2 * rng.standard_normal(n)


def sphere_system(n, m, rng):
    """generate_sphere_system (test/test_retractions.jl:1-31), seeded."""
    Rs = rng.random(m) + 1.0
    dirs = rng.standard_normal((n, m))
    dirs /= np.linalg.norm(dirs, axis=0)
    centers = dirs * Rs            # x0 = 0 lies on every sphere
    return centers, Rs


def _systems(ctx, n, m, rng):
    """name -> (device constraints, dense A, kind, qw, b)"""
    out = {}
    cons = L.sin_system_constraints(ctx, n, m)
    A = np.zeros((n, m))
    i = np.arange(m)
    A[2 * i + 1, i], A[2 * i, i] = 1.0, -1.0
    kind = np.zeros(n)
    kind[0:2 * m:2] = 1
    out["sin-sparse"] = (cons, A, kind, None, np.zeros(m))
    out["sin-dense"] = (L.ElementwiseConstraints(ctx, ctx.matrix(n, m, np.asfortranarray(A)), np.zeros(m), kind=kind), A, kind, None, np.zeros(m))
    centers, Rs = sphere_system(n, m, rng)
    cons = L.sphere_system_constraints(ctx, centers, Rs)
    out["sphere"] = (cons, -2.0 * centers, None, np.ones(m), cons.b.copy())
    Ar = rng.standard_normal((n, m)) / np.sqrt(n)
    kr = rng.integers(0, 3, n).astype(np.float64)
    br = rng.standard_normal(m) * 0.1
    qr = rng.standard_normal(m) * 0.01
    out["mixed-dense"] = (L.ElementwiseConstraints(ctx, ctx.matrix(n, m, np.asfortranarray(Ar)), br, kind=kr, qw=qr), Ar, kr, qr, br)
    import scipy.sparse as sp
    As = sp.random(n, m, density=3.0 / m, random_state=np.random.RandomState(5), data_rvs=lambda k: rng.standard_normal(k)).tocsr()
    out["mixed-sparse"] = (L.ElementwiseConstraints(ctx, L.SparseMatrix.from_scipy(ctx, As), br, kind=kr), As.toarray(), kr, None, br)
    return out


def test_c_jac_and_hessian_diagonal_match_numpy(dev_ctx):
    ctx = dev_ctx
    rng = np.random.default_rng(7)
    n, m = (700, 24) if _is_emu(ctx) else (5000, 100)
    for name, (cons, A, kind, qw, b) in _systems(ctx, n, m, rng).items():
        c_, jac_, hdiag = ew_callables(A, kind, qw, b)
        xh = rng.standard_normal(n)
        x = ctx.vector(n, xh)
        cv, cv0 = np.zeros(m), np.zeros(m)
        cons.c_(cv, x)
        c_(cv0, xh)
        scale = np.abs(A).T @ np.abs(_phi(np.zeros(n) if kind is None else kind, xh)) + 1.0
        assert np.max(np.abs(cv - cv0) / scale) < 1e-13, name
        J0 = np.zeros((m, n))
        jac_(J0, cv0, xh)
        cvj = np.zeros(m)
        cons.jac_(cons.Jct, cvj, x)
        np.testing.assert_array_equal(cvj, cv)                                      # jac! evaluates the same c!
        np.testing.assert_allclose(cons.Jct.download(), J0.T, rtol=1e-14, atol=1e-15, err_msg=name)
        if cons.Jsp is not None:                                                     # the sparse twin holds the same entries
            np.testing.assert_array_equal(cons.Jsp.to_dense().download(), cons.Jct.download()[:, :m])
        lam = rng.standard_normal(m)
        h0 = rng.standard_normal(n)
        hx = ctx.vector(n, h0)
        cons.hess_diag_(hx, x, lam)
        np.testing.assert_allclose(hx.download(), h0 + hdiag(xh, lam), rtol=1e-12, atol=1e-12, err_msg=name)


def _tangent_step(Zh, n, rng, length=5.0):
    step = rng.standard_normal(n)
    step -= Zh @ (Zh.T @ step)
    return step * (length / np.linalg.norm(step))


@pytest.mark.parametrize("system", ["sin-sparse", "sin-dense", "sphere", "mixed-dense"])
def test_retractions_of_the_reference_tests_with_device_resident_constraints(dev_ctx, system):
    """test/test_retractions.jl:90-103 (Newton) and :144-157 (ProjPenalty) with c! / jac! on the device: flags and counts equal to
    the oracle's, cval bit for bit c!(xnew), xtilde untouched, the reference's geometric assertions, iterates to 1e-10."""
    ctx = dev_ctx
    rng = np.random.default_rng(99)
    n, m = (300, 20) if _is_emu(ctx) else (1000, 100)
    cons, A, kind, qw, b = _systems(ctx, n, m, rng)[system]
    c_, jac_, _ = ew_callables(A, kind, qw, b)
    x0 = np.zeros(n) if system != "mixed-dense" else 0.3 * rng.standard_normal(n)
    if system.startswith("sin"):
        c_, jac_ = sin_system(n, m)[1:]                                              # the reference's own statement of the system
    x = ctx.vector(n, x0)
    cv = np.zeros(m)
    cons.jac_(cons.Jct, cv, x)
    Z = ctx.matrix(n, m)
    W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(cons.Jct, Z, W=W, Jsp=cons.Jsp)
    assert rank == m
    Zh = Z.download()
    length = 5.0 if system.startswith("sin") else 0.05
    step = _tangent_step(Zh, n, rng, length)
    xt_h = x0 + step
    if system == "mixed-dense":                                                      # start the retraction from a point near the manifold
        cv0 = np.zeros(m)
        c_(cv0, x0)
        cons.b = cons.b + cv0                                                        # ... by making x0 feasible
        b = b + cv0
        c_, jac_, _ = ew_callables(A, kind, qw, b)
    xtilde, xnew = ctx.vector(n, xt_h), ctx.vector(n)
    cval, cval2 = np.zeros(m), np.zeros(m)
    nr0 = R.NR(Zh, S, Vt, 1.0, 1000, R.NRWork(m), False, R.InequalityData())
    for U in (L.DeviceBasis(Z), L.DeviceBasis(Z, generator=(cons.Jct, W))):
        nr = L.NR(U, S, Vt, 1.0, 1000, L.NRWork(m), False, None)
        for tol in (1e-6, 1e-8, 1e-10):
            nr.tol = nr0.tol = tol
            flag, i, _ = L.retract_(cval, xnew, cons, xtilde, x, nr)
            xn = xnew.download()
            cons.c_(cval2, xnew)
            assert flag == 0 and np.max(np.abs(cval)) < tol
            np.testing.assert_array_equal(cval, cval2)                               # cval is c! evaluated at xnew (:97)
            np.testing.assert_array_equal(xtilde.download(), xt_h)                   # (:98)
            assert abs(step @ (xn - xt_h)) < 1e-6 * max(1.0, length)                 # (:99)
            xn0, cv0 = np.zeros(n), np.zeros(m)
            f0, i0, _ = R.retract_(cv0, xn0, c_, xt_h, x0, nr0)
            assert (flag, i) == (f0, i0), (system, tol)
            np.testing.assert_allclose(xn, xn0, atol=1e-10 * max(1.0, np.linalg.norm(xn0)))
            np.testing.assert_allclose(cval, cv0, atol=1e-12)

    # ProjPenalty (the reference's default retraction) with the device-resident jac!
    idc = L.InequalityDecomp(ctx, n, m, cons.Jct)
    pp = L.ProjPenalty(cons.jac_, None, S, Vt, m, 0.01, 1.0, 100, 200, L.ProjPenaltyWork(ctx, m, n, False), False, idc, None)
    pp0 = R.ProjPenalty(jac_, Zh, S, Vt, m, 0.01, 1.0, 100, 200, R.ProjPenaltyWork(m, n, m, n), False,
                        R.InequalityDecomp(np.zeros((0, 0)), *(np.zeros(0) for _ in range(5)), np.zeros((0, 0)), 0), R.InequalityData())
    for tol in (1e-6, 1e-8, 1e-10):
        pp.tol = pp0.tol = tol
        flag, i, pcg_i = L.retract_(cval, xnew, cons, xtilde, x, pp)
        xn = xnew.download()
        cons.c_(cval2, xnew)
        assert flag == 0 and np.max(np.abs(cval)) < tol
        np.testing.assert_array_equal(cval, cval2)
        np.testing.assert_array_equal(xtilde.download(), xt_h)
        assert np.linalg.norm(step) >= np.linalg.norm(xn - x0) - tol
        xn0, cv0 = np.zeros(n), np.zeros(m)
        f0, i0, p0 = R.retract_(cv0, xn0, c_, xt_h, x0, pp0)
        assert (flag, i) == (f0, i0) and abs(pcg_i - p0) <= 2, (system, tol, (flag, i, pcg_i), (f0, i0, p0))
        np.testing.assert_allclose(xn, xn0, atol=1e-10 * max(1.0, np.linalg.norm(xn0)))


def _trace_compare(tr, tr0, tol=1e-10):
    assert len(tr) == len(tr0)
    for k, (a, b) in enumerate(zip(tr, tr0)):
        nb = max(np.linalg.norm(b["x"]), 1.0)
        assert np.linalg.norm(a["x"] - b["x"]) <= tol * nb, (k, np.linalg.norm(a["x"] - b["x"]) / nb)
        for key in ("steptype", "mtype", "retract_iter1", "alpha", "ls_flag", "tn_iter"):
            if key in b:
                assert a.get(key) == b[key], (k, key, a.get(key), b[key])


@pytest.mark.parametrize("system,bounds,project", [("sin-sparse", False, False), ("sin-sparse", True, False), ("sphere", False, False),
                                                   ("mixed-dense", False, False), ("sin-dense", False, True), ("sin-dense", True, False)])
def test_optimize_end_to_end_against_the_oracle(dev_ctx, monkeypatch, system, bounds, project):
    """optimize(f, grad!, c!, jac!, hess_lag_vec!, x0, xl, xu, m) (src/optimize.jl:119) with everything resident on the device against
    the oracle's run with host callables of the same functions: equal counts, step types, retraction iterations, accepted steps;
    iterates within 1e-10 after every outer iteration."""
    ctx = dev_ctx
    rng = np.random.default_rng(3)
    n, m = (240, 12) if _is_emu(ctx) else (10000, 64)
    cons, A, kind, qw, b = _systems(ctx, n, m, rng)[system]
    c_, jac_, hdiag = ew_callables(A, kind, qw, b)
    target = 0.5 * rng.standard_normal(n)
    if bounds:
        # bounds present but inactive at the optimum: with ACTIVE bounds the squared-slack Hessian degenerates, the truncated-Newton solves take
        # 5 .. 34 iterations and amplify rounding differences to 6e-10 by the seventh outer iteration -- on the emulator (no FMA, the host's own
        # sin) as on the GPU, so a property of the problem's conditioning, not of the kernels (tools/ew_diffs.py prints the growth)
        target = np.clip(target, -0.8, 0.8)
    x0 = np.zeros(n)
    if system == "mixed-dense":
        x0 = 0.2 * rng.standard_normal(n)
    xl = xu = None
    if bounds:
        xl = np.where(np.arange(n) % 4 == 1, -1.0, np.where(np.arange(n) % 4 == 3, -1.0, -np.inf))
        xu = np.where(np.arange(n) % 4 == 2, 1.0, np.where(np.arange(n) % 4 == 3, 1.0, np.inf))
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target, xl=xl, xu=xu)          # f = sum (x - target)^2
    p = L.LFPSQPParams(do_project_retract=project, maxiter=12, disp=L.DisplayOption.off)
    tr = []
    # every successful retraction of the run returns c!(xnew) bit for bit (test/test_retractions.jl:97) -- with bounds too, where the step is the
    # stacked instantiation of the one-pass kernel and c! the plain one
    import lfpsqp_jl_amd.linesearch as LS
    checked = []
    real_retract = LS.retract_

    def checking_retract(cval, xnew, c_dev, xtilde, x, method):
        out = real_retract(cval, xnew, c_dev, xtilde, x, method)
        if out[0] == 0 and isinstance(method, L.NR):
            cv2 = np.zeros_like(cval)
            c_dev.c_(cv2, xnew)
            np.testing.assert_array_equal(cval, cv2)
            checked.append(1)
        return out
    monkeypatch.setattr(LS, "retract_", checking_retract)
    ctx.options.ls_batch = 1                    # (one by one: every retraction goes through the wrapper)
    xd, obj, lam, ti = prob.optimize(x0, p, trace=tr)
    ctx.options.ls_batch = 4
    monkeypatch.setattr(LS, "retract_", real_retract)
    assert project or len(checked) > 0

    f = lambda xx: float(np.sum((xx[:n] - target) ** 2))

    def grad_(g, xx):
        g[:n] = 2.0 * (xx[:n] - target)

    def hlv_(dest, src, xx, lam_):
        dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]

    p0 = R.LFPSQPParams(do_project_retract=project, maxiter=12, disp=R.DisplayOption.off)
    tr0 = []
    x0h, obj0, lam0, ti0 = R.optimize_core(f, grad_, c_, jac_, hlv_, x0, xl, xu, m, p0, trace=tr0)
    assert ti.iter == ti0.iter and ti.condition.name == ti0.condition.name
    _trace_compare(tr, tr0)
    np.testing.assert_allclose(obj, obj0, rtol=1e-10)
    np.testing.assert_allclose(xd, x0h, atol=1e-10 * max(1.0, np.linalg.norm(x0h)))
