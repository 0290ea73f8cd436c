"""Device versions of the reference's test/test_retractions.jl + test_linesearch.jl, and
end-to-end `optimize` trajectories of BASELINE configs 1-4 against the oracle."""
import math
import warnings

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .test_oracle_reference_properties import rosenbrock, sin_system


def _is_emu(ctx):
    return "emulator" in ctx.device_name


@pytest.fixture
def sin_setup(dev_ctx):
    ctx = dev_ctx
    rng = np.random.default_rng(99)
    n, m = (1000, 100) if not _is_emu(ctx) else (300, 20)
    x0, c_, jac_ = sin_system(n, m)
    cval = np.zeros(m)
    J = np.zeros((m, n), order='F')
    jac_(J, cval, x0)
    Jct = ctx.matrix(n, m, np.asfortranarray(J.T))
    Z = ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(Jct, Z)
    assert rank == m
    Zh = Z.download()
    step = rng.standard_normal(n)
    step -= Zh @ (Zh.T @ step)
    step *= 5.0 / np.linalg.norm(step)
    return ctx, n, m, x0, c_, jac_, Jct, Z, S, Vt, step, rng


def test_newton_retraction_with_host_callback(sin_setup):
    """test_retractions.jl:90-103: nonlinear c! as a host callable (x downloaded per evaluation)."""
    ctx, n, m, x0, c_, jac_, Jct, Z, S, Vt, step, _ = sin_setup
    nr = L.NR(L.DeviceBasis(Z), S, Vt, 1.0, 1000, L.NRWork(m), False, None)
    xt_h = x0 + step
    xtilde, x, xnew = ctx.vector(n, xt_h), ctx.vector(n, x0), ctx.vector(n)
    cval, cval2 = np.zeros(m), np.zeros(m)
    # oracle with the same factors
    nr0 = R.NR(Z.download(), S, Vt, 1.0, 1000, R.NRWork(m), False, R.InequalityData())
    for tol in (1e-6, 1e-8):
        nr.tol = nr0.tol = tol
        flag, i, _ = L.retract_(cval, xnew, c_, xtilde, x, nr)
        xn = xnew.download()
        c_(cval2, xn)
        assert flag == 0 and np.max(np.abs(cval)) < tol
        assert np.all(cval == cval2)                       # cval is c! evaluated at xnew
        np.testing.assert_array_equal(xtilde.download(), xt_h)
        assert abs(step @ (xn - xt_h)) < 1e-6
        xn0, cv0 = np.zeros(n), np.zeros(m)
        f0, i0, _ = R.retract_(cv0, xn0, c_, xt_h, x0, nr0)
        assert (flag, i) == (f0, i0)
        np.testing.assert_allclose(xn, xn0, atol=1e-12)


def test_pcg_and_projection_penalty(sin_setup):
    """test_retractions.jl:105-141 (pcg!) and :144-157 (ProjPenalty) on the device."""
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    ctx, n, m, x0, c_, jac_, Jct, Z, S, Vt, step, rng = sin_setup
    # pcg!: solve (mu I + J'J) x = b with a random J
    Jr = rng.standard_normal((m, n))
    Jd = ctx.matrix(n, m, np.asfortranarray(Jr.T))
    w = L.ProjPenaltyWork(ctx, m, n, False)
    for mu in (1e-1, 1e-2, 1e-4):
        bh = rng.standard_normal(n)
        x, r = ctx.vector(n), ctx.vector(n, bh)
        flag, i = L.pcg_(mu, _JacPlain(Jd, w), L.no_precondition, x, r, w.p, w.z, None, 1e-6, 100)
        x0h, r0 = np.zeros(n), bh.copy()
        f0, i0 = R.pcg_(mu, Jr, R.no_precondition, x0h, r0, np.zeros(n), np.zeros(n), np.zeros(m), 1e-6, 100)
        assert (flag, i) == (f0, i0) and flag == 0
        xh = x.download()
        assert np.linalg.norm(r.download()) < 1e-6
        assert np.linalg.norm(mu * xh + Jr.T @ (Jr @ xh) - bh) < 1e-6
        np.testing.assert_allclose(xh, x0h, atol=1e-10)

    # ProjPenalty with a device-side jac! adapter around the host Jacobian
    def jac_dev(Jct_, cval, xdev):
        J = np.zeros((m, n), order='F')
        jac_(J, cval, xdev.download(n, 0))
        Jct_.upload(np.asfortranarray(J.T))

    idc = L.InequalityDecomp(ctx, n, m, Jct)
    pp = L.ProjPenalty(jac_dev, None, S, Vt, m, 0.01, 1.0, 100, 200, L.ProjPenaltyWork(ctx, m, n, False), False, idc, None)
    J0 = np.zeros((m, n), order='F')
    pp0 = R.ProjPenalty(jac_, Z.download(), S, Vt, m, 0.01, 1.0, 100, 200, R.ProjPenaltyWork(m, n, m, n), False,
                        R.InequalityDecomp(np.zeros((0, 0)), *(np.zeros(0) for _ in range(5)), np.zeros((0, 0)), 0), R.InequalityData())
    xt_h = x0 + step
    xtilde, x, xnew = ctx.vector(n, xt_h), ctx.vector(n, x0), ctx.vector(n)
    cval, cval2 = np.zeros(m), np.zeros(m)
    for tol in (1e-6, 1e-8, 1e-10):
        pp.tol = pp0.tol = tol
        flag, i, pcg_i = L.retract_(cval, xnew, c_, xtilde, x, pp)
        xn = xnew.download()
        c_(cval2, xn)
        assert flag == 0 and np.max(np.abs(cval)) < tol
        assert np.all(cval == cval2)
        np.testing.assert_array_equal(xtilde.download(), xt_h)
        assert np.linalg.norm(step) >= np.linalg.norm(xn - x0) - tol
        xn0, cv0 = np.zeros(n), np.zeros(m)
        f0, i0, p0 = R.retract_(cv0, xn0, c_, xt_h, x0, pp0)
        assert (flag, i, pcg_i) == (f0, i0, p0)
        np.testing.assert_allclose(xn, xn0, atol=1e-11)


def test_linesearch_known_answers(dev_ctx):
    """test_linesearch.jl: f = x^2, x = -0.23, d = 1 => Armijo alpha = 0.25, exact alpha = 0.23."""
    ctx = dev_ctx
    f = lambda v: float(v.download()[0] ** 2)
    x, xnew, d, g = ctx.vector(1, [-0.23]), ctx.vector(1), ctx.vector(1, [1.0]), ctx.vector(1, [-0.46])
    fval = 0.23 ** 2
    p = L.LFPSQPParams()
    flag, t1, t2, newf, f_diff, step_diff, alpha = L.armijo_(xnew, x, 1, d, g, f, fval, L.Euclidean(), np.zeros(0), None, p, L.ArmijoWork(x))
    assert flag == t1 == t2 == 0 and x.download()[0] == -0.23
    assert newf == pytest.approx(f(xnew)) and f_diff == pytest.approx(fval - newf)
    assert step_diff == pytest.approx(alpha) and alpha == pytest.approx(0.25)
    flag, t1, t2, newf, f_diff, step_diff, alpha = L.exact_linesearch_(xnew, x, 1, d, f, fval, L.Euclidean(), np.zeros(0), None, p,
                                                                       L.ExactLinesearchWork(x))
    assert flag == t1 == t2 == 0 and x.download()[0] == -0.23
    assert newf == pytest.approx(f(xnew)) and f_diff == pytest.approx(fval - newf)
    assert step_diff == pytest.approx(alpha) and alpha == pytest.approx(0.23, abs=1e-6)


# ------------------------------------------------------------------------------- end to end
def _note(msg):
    """Which branch a tolerant comparison took must be visible in the test log (pytest's warnings summary / -s output)."""
    print("[trajectory parity]", msg)
    warnings.warn("[trajectory parity] " + msg)


def _compare_traces(tr, tr0, rtol=1e-10, pcg_slack=0, failed_retractions_may_differ=False):
    """Trajectory parity: iterates within rtol, and equal counts / flags / step types.  `pcg_slack`
    tolerates a +-k difference in the CUMULATIVE inner pcg! count of a ProjPenalty retraction: its
    stopping test `norm(r) > tol` can flip by one iteration when the residual lands within rounding
    of tol (different fp64 summation order than the oracle; SURVEY §7 "hard parts").

    `failed_retractions_may_differ` (config 4 on the GPU): a linesearch whose trial retractions FAIL
    (Newton iterations that do not converge within the reference's 100-iteration limit, src/retractions.jl:133,
    answered by halving alpha, src/linesearch.jl:57-60) iterates chaotically, so whether such a run happens to
    land inside tol at iteration 80 or never is decided by the last bit (FMA contraction, summation order) --
    for the reference itself as much as for this build.  For an outer iteration whose ORACLE count contains a
    failed retraction (>= 100) the Newton-iteration count may therefore differ; the accepted step must still
    agree (same alpha, iterate within rtol).  If even the accepted alpha differs (a stray convergence that also
    passes Armijo) the trajectories have legitimately forked: returns the fork index instead of asserting."""
    if not failed_retractions_may_differ:
        assert len(tr) == len(tr0)
    worst = max((np.linalg.norm(a['x'] - b['x']) / max(np.linalg.norm(b['x']), 1e-300) for a, b in zip(tr, tr0)), default=0.0)
    if rtol > 1e-10:        # a site looser than north_star's 1e-10 says what it measured (the reason for its tolerance is stated at the call)
        import inspect
        _note(f"{inspect.stack()[1].function}: max relative deviation of an iterate {worst:.2e} (tolerance {rtol:g})")
    for a, b in zip(tr, tr0):
        chaotic = failed_retractions_may_differ and (b.get('retract_iter1') or 0) >= 100
        if chaotic and a.get('retract_iter1') != b.get('retract_iter1'):
            _note(f"outer iteration {a['iter']}: Newton-iteration count of a linesearch with failed retractions differs "
                  f"({a.get('retract_iter1')} vs oracle {b.get('retract_iter1')}), accepted alpha {a.get('alpha')} vs {b.get('alpha')}")
        if chaotic and a.get('alpha') != b.get('alpha'):
            _note(f"TRAJECTORY FORKED at outer iteration {a['iter']} (accepted alpha {a.get('alpha')} vs oracle {b.get('alpha')})")
            return a['iter']
        assert np.linalg.norm(a['x'] - b['x']) <= rtol * np.linalg.norm(b['x']), f"iterate {a['iter']} deviates"
        for k in ('tn_iter', 'steptype', 'mtype', 'retract_iter1', 'alpha', 'ls_flag', 'rank'):
            if chaotic and k == 'retract_iter1':
                continue
            assert a.get(k) == b.get(k), (a['iter'], k, a.get(k), b.get(k))
        if a.get('retract_iter2') is not None:
            assert abs(a['retract_iter2'] - b['retract_iter2']) <= pcg_slack, (a['iter'], a['retract_iter2'], b['retract_iter2'])
    assert len(tr) == len(tr0)
    return None


def test_config1_rosenbrock_through_host_callbacks(dev_ctx):
    """BASELINE configs[0] / README.md:18-37 through the device driver with host callables."""
    f, dv = rosenbrock()
    x, obj, lam, ti = L.optimize(f, np.zeros(2), L.LFPSQPParams(disp=L.DisplayOption.off),
                                 derivatives=L.Derivatives(dv.grad_, dv.hess_lag_vec_), ctx=dev_ctx)
    assert ti.condition == L.TerminationCondition.f_tol and ti.iter == 17 and len(obj) == 18
    assert ti.kkt_diff == pytest.approx(4.332627751789361e-5, rel=1e-9)
    assert ti.f_diff == pytest.approx(1.0898882046786806e-7, rel=1e-8)
    assert ti.step_diff == pytest.approx(0.0007384068067118611, rel=1e-8)


@pytest.mark.parametrize("do_project_retract", [False, True])
def test_config2_single_linear_equality(dev_ctx, do_project_retract):
    """BASELINE configs[1]: f = x'x, c = x_1 - 0.75 (README.md:42-53 pattern)."""
    ctx = dev_ctx
    n = 1_000_000 if not _is_emu(ctx) else 2000          # BASELINE configs[1]'s stated size on the GPU
    prob0, x0 = synth.config2(n)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, 1,
                                     R.LFPSQPParams(do_project_retract=do_project_retract, disp=R.DisplayOption.off), trace=tr0)
    J = np.zeros((n, 1), order='F')
    J[0, 0] = 1.0
    P = L.QuadLinearBallBox(ctx, n, 1, ctx.matrix(n, 1, J), np.array([0.75]))
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=do_project_retract, disp=L.DisplayOption.off), trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    _compare_traces(tr, tr0, pcg_slack=2 if do_project_retract else 0)
    np.testing.assert_allclose(lam, lamr, rtol=1e-9)
    np.testing.assert_allclose(obj, objr, rtol=1e-12)


@pytest.mark.parametrize("do_project_retract", [False, True])
def test_config3_dense_linear_equalities(dev_ctx, do_project_retract):
    """BASELINE configs[2] shape at oracle-sized n: trajectory parity, both retractions."""
    ctx = dev_ctx
    n, m = (20000, 32) if not _is_emu(ctx) else (2500, 6)
    prob0, x0 = synth.config3(n, m)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, m,
                                     R.LFPSQPParams(do_project_retract=do_project_retract, disp=R.DisplayOption.off), trace=tr0)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), prob0.b)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=do_project_retract, disp=L.DisplayOption.off), trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    _compare_traces(tr, tr0, pcg_slack=2 if do_project_retract else 0)
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(lam, lamr, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("case", ["cond1e7", "cond1e9", "duplicates"])
@pytest.mark.parametrize("bounds", [False, True])
def test_ill_conditioned_jacobian_takes_the_references_retraction(dev_ctx, case, bounds):
    """The rank scan of src/optimize.jl:297-302 is absolute (sigma >= eps_rank = 1e-10) on dgesvd's singular values, and
    :396-412 picks the Newton retraction iff rank == m.  An equality block of condition 1e7 / 1e9 (all sigma >= 1e-10) is
    therefore full rank in the reference -- NR, full tangent projection, all multipliers -- and one with exactly duplicated
    rows is rank deficient -- ProjPenalty.  The device driver must take the same branch with the same rank, and the first
    outer iterates must agree with the oracle's dgesvd path to the accuracy the data allow: the tangent projector of an
    ill-conditioned block is only determined to eps * cond (for dgesvd as for any other backward-stable factorisation)."""
    ctx = dev_ctx
    if _is_emu(ctx) and case == "duplicates" and bounds:
        pytest.skip("ProjPenalty's inner pcg! with bounds: a minute of launch emulation; the GPU suite runs it")
    n, m = (1500, 8) if not _is_emu(ctx) else (400, 5)
    rng = np.random.default_rng(17)
    Q1, _ = np.linalg.qr(rng.standard_normal((n, m)))
    Q2, _ = np.linalg.qr(rng.standard_normal((m, m)))
    if case == "duplicates":
        sv, cond = np.logspace(0, -2, m), 1e2
    else:
        cond = 1e7 if case == "cond1e7" else 1e9
        sv = np.logspace(0, -np.log10(cond), m)
    Jct = np.asfortranarray((Q1 * sv) @ Q2.T)
    if case == "duplicates":
        Jct[:, m - 1] = Jct[:, 0]
    xs = synth.hash_vector(2, n)
    prob0 = synth.QuadLinearProblem(Jct, Jct.T @ xs)
    x0 = xs.copy()
    xl = xu = None
    if bounds:
        i = np.arange(n)
        xl = np.where(i % 3 == 1, -2.0, -np.inf)
        xu = np.where(i % 3 == 2, 2.0, np.inf)
    maxiter = 3 if not (_is_emu(ctx) and case == "duplicates") else 1      # (ProjPenalty's inner pcg! is slow on the emulator)
    tr0, tr = [], []
    p0 = R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m, Jct), prob0.b, xl=xl, xu=xu)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter), trace=tr)
    want_rank = m - 1 if case == "duplicates" else m
    if case == "duplicates" and not bounds:
        # Without bounds the REFERENCE cannot run a rank-deficient block at all: projcg! ends with mul!(lambda, U', r) on a
        # length-m lambda and the rank x n view U' (src/projcg.jl:118, src/optimize.jl:369,381) -- a DimensionMismatch in Julia,
        # a shape error in the oracle.  The device driver never forms that unused lambda and carries on with ProjPenalty.
        with pytest.raises(ValueError):
            R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, xl, xu, m, p0, trace=tr0)
        assert [t['rank'] for t in tr] == [want_rank] * len(tr) and tr[0].get('mtype') == 1
        assert np.abs(Jct.T @ x - prob0.b).max() < 1e-5 and all(obj[k + 1] <= obj[k] for k in range(len(obj) - 1))
        return
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, xl, xu, m, p0, trace=tr0)
    assert [t['rank'] for t in tr0] == [want_rank] * len(tr0)                # what the reference's rule gives ...
    assert [t['rank'] for t in tr] == [t['rank'] for t in tr0]               # ... and the device agrees
    assert [t.get('mtype') for t in tr] == [t.get('mtype') for t in tr0]     # NR (0) iff rank == m, else ProjPenalty (1)
    assert tr0[0].get('mtype') == (1 if case == "duplicates" else 0)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    tol = max(1e-10, 1e3 * np.finfo(float).eps * cond)
    for a, b in zip(tr, tr0):
        assert np.linalg.norm(a['x'] - b['x']) <= tol * np.linalg.norm(b['x']), (a['iter'], np.linalg.norm(a['x'] - b['x']))
        assert a.get('steptype') == b.get('steptype') and a.get('alpha') == b.get('alpha')
    np.testing.assert_allclose(obj, objr, rtol=max(1e-12, tol))


@pytest.mark.parametrize("mode", ["default", "one_by_one", "matrix_cores"])
def test_config4_ball_box_newton_retraction(dev_ctx, mode):
    """BASELINE configs[3]: equalities + ball (slack form) + four-way bounds, NR retraction, Armijo, from the start whose first five
    linesearches FAIL repeatedly (1311, 1107, 605, 405, 204 Newton iterations in the oracle; the fifth is decided by whether the trial at
    alpha = 0.5 converges inside its 100th step, tools/c4_fork_probe.py).
    ``default`` (DeviceOptions as shipped: batched trial retractions in the EXACT mode -- bit for bit the one-by-one retractions) and
    ``one_by_one`` (ls_batch = 1) must follow the oracle's trajectory: every accepted step, 1e-10 on every iterate, NO fork.
    ``matrix_cores`` (the opt-in ls_batch_matrix_cores): the trials of a pass are summed in another order and may land elsewhere in the
    chaotic set -- another count, sometimes another accepted alpha: only this setting may take the fork branch below."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    if emu and mode != "default":
        pytest.skip("the emulator case never reaches a failing linesearch: one setting is enough")
    ctx.options.ls_batch = 1 if mode == "one_by_one" else 0
    ctx.options.ls_batch_matrix_cores = mode == "matrix_cores"
    n, m = (4000, 16) if not emu else (200, 4)
    P0 = synth.BallBoxProblem(n, m)
    x0 = P0.x0
    if emu:   # start near the feasible point so the emulator is not asked for ~2000 Newton iterations
        x0 = 0.97 * synth.hash_vector(2, n) + 0.03 * P0.x0
    maxiter = 10000 if not emu else 3
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, P0.m, P0.p,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter),
                                     derivatives=P0.derivatives(), trace=tr0)
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter), trace=tr)
    assert ti.condition.name == tir.condition.name
    # on the emulator (no FMA contraction) every count matches; on the GPU the COUNTS of failed retractions may differ (see helper)
    fork = _compare_traces(tr, tr0, failed_retractions_may_differ=not emu)
    if fork is None:
        strict = all(a.get('retract_iter1') == b.get('retract_iter1') for a, b in zip(tr, tr0))
        _note((f"config 4 from x0, {mode}: no fork -- every accepted step equals the oracle's"
               + ("; the Newton-iteration counts of the failing linesearches too" if strict else "")) if not emu else "config 4 (emulator): strict")
        assert ti.iter == tir.iter
        assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
        np.testing.assert_allclose(lam, lamr, rtol=1e-7, atol=1e-10)
    else:       # forked inside a failed-retraction linesearch: same (unique) optimum of the convex problem, to the KKT tolerance
        assert mode == "matrix_cores", "the default options and the one-by-one search must follow the oracle's trajectory"
        assert abs(obj[-1] - objr[-1]) <= 1e-8 * abs(objr[-1])
        assert np.linalg.norm(x - xr) <= 1e-4 * np.linalg.norm(xr)
    # feasibility of the result (LFPSQP iterates are feasible)
    if not emu:
        assert np.abs(P0.eq.Jct.T @ x - P0.eq.b).max() < 1e-5 and x @ x <= P0.R2 + 1e-5
        assert np.all(x >= P0.xl - 1e-9) and np.all(x <= P0.xu + 1e-9)


def test_config4_strict_trajectory_when_no_retraction_fails(dev_ctx):
    """BASELINE configs[3] (equalities + ball in slack form + four-way bounds, Newton retraction, Armijo) from a start near
    the feasible set, where no trial retraction runs into the reference's 100-iteration limit: there is no chaotic regime,
    so the comparison with the oracle is STRICT on the GPU as well -- equal lengths, counts, step types, accepted alpha
    (src/linesearch.jl:49-60), Newton iterations (src/retractions.jl:133-165), iterates within 1e-10 after every outer
    iteration."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m, maxiter = (4000, 16, 60) if not emu else (200, 4, 3)
    P0 = synth.BallBoxProblem(n, m)
    x0 = 0.9 * synth.hash_vector(2, n) + 0.1 * P0.x0
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, P0.m, P0.p,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter),
                                     derivatives=P0.derivatives(), trace=tr0)
    assert all((t.get('retract_iter1') or 0) < 100 for t in tr0) and len(tr0) == maxiter + 1      # the premise of strictness
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter), trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    assert _compare_traces(tr, tr0) is None
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(obj, objr, rtol=1e-12)
    np.testing.assert_allclose(lam, lamr, rtol=1e-7, atol=1e-10)


def test_config4_default_projection_penalty_with_bounds(dev_ctx):
    """Same problem through the reference's DEFAULT retraction (ProjPenalty + pcg! with the full
    bound Jacobian operator, src/retractions.jl:324)."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (2000, 8) if not emu else (120, 3)
    P0 = synth.BallBoxProblem(n, m)
    x0 = 0.9 * synth.hash_vector(2, n) + 0.1 * P0.x0
    maxiter = 4 if not emu else 2
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, P0.m, P0.p,
                                     R.LFPSQPParams(disp=R.DisplayOption.off, maxiter=maxiter), derivatives=P0.derivatives(), trace=tr0)
    Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(disp=L.DisplayOption.off, maxiter=maxiter), trace=tr)
    assert ti.iter == tir.iter
    _compare_traces(tr, tr0, pcg_slack=2)        # (1e-10; measured on the GPU: 3.9e-12, emulator 2.2e-13)


def test_pcg_with_exact_preconditioners(sin_setup):
    """test_retractions.jl:127-139: with an exact-inverse preconditioner pcg! converges in exactly ONE
    iteration (generic M! path), and proj_precondition! (src/retractions.jl:248-257) is that inverse
    for J'J + mu I when (U, Sigma) factor J'."""
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    ctx, n, m, x0, c_, jac_, Jct, Z, S, Vt, step, rng = sin_setup
    w = L.ProjPenaltyWork(ctx, m, n, False)
    Jh = Jct.download().T                       # m x n
    for mu in (1e-1, 1e-2):
        bh = rng.standard_normal(n)
        Afull = mu * np.eye(n) + Jh.T @ Jh

        def M_host(z, r):
            z.upload(np.linalg.solve(Afull, r.download()))
            return z
        x, r = ctx.vector(n), ctx.vector(n, bh)
        flag, i = L.pcg_(mu, _JacPlain(Jct, w), M_host, x, r, w.p, w.z, None, 1e-6, 100)
        assert flag == 0 and i == 1
        assert np.linalg.norm(r.download()) < 1e-6
        assert np.linalg.norm(Afull @ x.download() - bh) < 1e-6
        # proj_precondition! == the same exact inverse, on the device
        tmp_m = ctx.vector(m)

        def M_dev(z, r):
            return L.proj_precondition_(z, r, mu, Z, S, m, tmp_m)
        x, r = ctx.vector(n), ctx.vector(n, bh)
        flag, i = L.pcg_(mu, _JacPlain(Jct, w), M_dev, x, r, w.p, w.z, None, 1e-6, 100)
        assert flag == 0 and i == 1
        assert np.linalg.norm(Afull @ x.download() - bh) < 1e-6
        z0 = np.zeros(n)
        R.proj_precondition_(z0, bh, mu, Z.download(), S, m, np.zeros(m))
        zz = ctx.vector(n)
        L.proj_precondition_(zz, ctx.vector(n, bh), mu, Z, S, m, tmp_m)
        np.testing.assert_allclose(zz.download(), z0, rtol=1e-10, atol=1e-12)


def test_exact_linesearch_through_the_driver(dev_ctx):
    """param.linesearch = exact (src/optimize.jl:419 -> src/linesearch.jl:107-339) end to end, NR retraction."""
    ctx = dev_ctx
    n, m = (6000, 8) if not _is_emu(ctx) else (1500, 4)
    prob0, x0 = synth.config3(n, m)
    prob0.xc = 0.3                                              # objective ||x - 0.3||^2: the exact search has work to do
    tr0, tr = [], []
    par0 = R.LFPSQPParams(do_project_retract=False, linesearch=R.LinesearchOption.exact, disp=R.DisplayOption.off, maxiter=3)
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, m, par0, trace=tr0)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), prob0.b, xc=0.3)
    par = L.LFPSQPParams(do_project_retract=False, linesearch=L.LinesearchOption.exact, disp=L.DisplayOption.off, maxiter=3)
    x, obj, lam, ti = P.optimize(x0, par, trace=tr)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    for a, b in zip(tr, tr0):
        assert np.linalg.norm(a['x'] - b['x']) <= 1e-9 * np.linalg.norm(b['x'])
        assert a.get('alpha') == pytest.approx(b.get('alpha'), rel=1e-6) or a.get('alpha') is None
    np.testing.assert_allclose(obj, objr, rtol=1e-10)


def test_optimize_surface_with_host_callables_inequalities_and_bounds(dev_ctx):
    """The north-star surface optimize(f, c!, d!, x0, xl, xu, m, p) with arbitrary HOST callables:
    slack transformation (src/optimize.jl:13-71), bounds, a general (non-diagonal) Hessian callable through
    the generic projcg path with the bound operator Q, Newton retraction with a host c!."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (60, 3) if emu else (400, 6)
    rng = np.random.default_rng(17)
    P0 = synth.BallBoxProblem(n, m)
    Bq = rng.standard_normal((n, 4)) * 0.3                       # f = ||x - 0.2||^2 + 0.5 ||B'x||^2 : non-diagonal Hessian
    f = lambda x: float((x - 0.2) @ (x - 0.2) + 0.5 * np.sum((Bq.T @ x) ** 2))

    def grad_(g, x):
        g[:] = 2.0 * (x - 0.2) + Bq @ (Bq.T @ x)

    dv0 = P0.derivatives()

    def hlv_(dest, src, x, lam):
        dest[:] = (2.0 + 2.0 * lam[m]) * src + Bq @ (Bq.T @ src)

    x0 = 0.95 * synth.hash_vector(2, n) + 0.05 * 0.5
    maxiter = 3 if emu else 6
    dv = R.Derivatives(grad_=grad_, hess_lag_vec_=hlv_, jac_c_=dv0.jac_c_, jac_d_=dv0.jac_d_)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(f, P0.c_, P0.d_, x0, P0.xl, P0.xu, m, 1,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter),
                                     derivatives=dv, trace=tr0)
    x, obj, lam, ti = L.optimize(f, P0.c_, P0.d_, x0, P0.xl, P0.xu, m, 1,
                                 L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter),
                                 derivatives=L.Derivatives(grad_=grad_, hess_lag_vec_=hlv_, jac_c_=dv0.jac_c_, jac_d_=dv0.jac_d_),
                                 ctx=ctx, trace=tr)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name and len(x) == n
    _compare_traces(tr, tr0)                      # (1e-10; measured on the GPU: 6.9e-16)
    np.testing.assert_allclose(obj, objr, rtol=1e-10)
    np.testing.assert_allclose(lam, lamr, rtol=1e-6, atol=1e-9)


def test_retract_pp_c_matches_python_mirror(dev_ctx):
    """lfpsqp_retract_pp (C) against the statement-by-statement Python mirror of src/retractions.jl:265-441 on
    the device primitives, with bounds (stacked operator) and the device-resident ball/linear constraints."""
    from .pp_mirror import retract_pp_reference_loop
    ctx = dev_ctx
    n, m = (1500, 5) if not _is_emu(ctx) else (150, 3)
    P0 = synth.BallBoxProblem(n, m)
    N, M = n + 1, m + 1
    Jct = ctx.matrix(N, M).hash_fill(1, 0, n, 1.0, n, m)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    x0a, xl, xu = P.aux_start(0.9 * synth.hash_vector(2, n) + 0.05)
    idata = L.InequalityData(ctx, xl, xu)
    X = L.StackedVector(ctx, N)
    X.upload(x0a, 0)
    L.generate_initial_y_(X, idata)
    xt = X.download2() + 0.01 * np.random.default_rng(3).standard_normal(2 * N)
    res = []
    for fn in (L.retract_, None):
        idc = L.InequalityDecomp(ctx, N, M, Jct)
        pp = L.ProjPenalty(P.jac_, None, np.ones(M), np.eye(M), M, 0.01, 1e-8, 100, 100, L.ProjPenaltyWork(ctx, M, N, True), True, idc, idata)
        xtilde, xnew = L.StackedVector(ctx, N).upload2(xt), L.StackedVector(ctx, N)
        cval = np.zeros(M)
        if fn is None:
            out = retract_pp_reference_loop(cval, xnew, P.cons, xtilde, X, pp)
        else:
            out = fn(cval, xnew, P.cons, xtilde, X, pp)
        res.append((out, xnew.download2(), cval.copy()))
    (o1, x1, c1), (o2, x2, c2) = res
    assert o1[0] == o2[0] == 0 and o1[1] == o2[1] and abs(o1[2] - o2[2]) <= 2
    np.testing.assert_allclose(x1, x2, rtol=1e-10, atol=1e-12)
    assert np.max(np.abs(c1)) < 1e-8


def test_noise_option_runs_and_stays_feasible(dev_ctx):
    """param.beta > 0 (src/optimize.jl:264-273): random perturbation of the steepest-descent step; the RNG stream
    of the reference cannot be matched, so only the invariants are checked (feasible iterates, descent)."""
    ctx = dev_ctx
    n, m = (3000, 5) if not _is_emu(ctx) else (1200, 3)
    prob0, x0 = synth.config3(n, m)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), prob0.b)
    np.random.seed(0)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, beta=1e-3, t_beta=5, maxiter=4))
    assert np.abs(prob0.Jct.T @ x - prob0.b).max() < 1e-5          # iterates are feasible
    assert obj[-1] < obj[0]


@pytest.mark.parametrize("m_lin,has_ball", [(1, False), (6, True), (40, False), (128, True), (300, True)])
def test_newton_retraction_one_stream_step(dev_ctx, monkeypatch, m_lin, has_ball):
    """The one-stream Newton step (basis generator Z = Jct*W known: both products of a step run over Jct) against
    the two-stream step (LFPSQP_ONEPASS=-1) and the oracle -- same iteration counts, iterates equal to rounding;
    covers the shifted last column group (m not a multiple of 4) and the m < 4 fallback."""
    ctx0 = dev_ctx
    emu = _is_emu(ctx0)
    n = 2500 if emu else 300_000
    if emu and m_lin > 64:
        n = 2100 if m_lin < 256 else 800
    m = m_lin + (1 if has_ball else 0)
    N = n + (1 if has_ball else 0)
    rng = np.random.default_rng(5 + m)
    results = {}
    for mode in ("-1", "0"):
        monkeypatch.setenv("LFPSQP_ONEPASS", mode)
        ctx = L.Context(0, ctx0.L)
        Jct = ctx.matrix(N, m).hash_fill(1, 0, n, 1.0, n, m_lin)
        xs_h = np.concatenate([synth.hash_vector(2, n), [0.0]])[:N]
        if has_ball:
            xs_h[n] = xs_h[:n] @ xs_h[:n] - 0.4 * n          # slack makes the ball equality hold at xs
        Jh = Jct.download()
        b = Jh[:, :m_lin].T @ xs_h
        cons = L.DeviceConstraints(Jct, m_lin, b, has_ball, 0.4 * n, n, n if has_ball else -1)
        xs = ctx.vector(N, xs_h)
        cv = np.zeros(m)
        cons.jac_(Jct, cv, xs)                               # ball column of Jct at xs
        assert np.max(np.abs(cv)) < 1e-8
        Z = ctx.matrix(N, m)
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(Jct, Z, W=W)
        assert rank == m
        np.testing.assert_allclose(Jct.download() @ W, Z.download(), atol=1e-12)
        pert = 1e-2 * rng.standard_normal(N) if mode == "-1" else results["pert"]
        results.setdefault("pert", pert)
        xt = ctx.vector(N, xs_h + pert)
        xnew = ctx.vector(N)
        nr = L.NR(L.DeviceBasis(Z, generator=(Jct, W)), S, Vt, 1e-9, 50, L.NRWork(m), False, None)
        cval = np.zeros(m)
        flag, it, _ = L.retract_(cval, xnew, cons, xt, xs, nr)
        results[mode] = (flag, it, xnew.download(), cval.copy())
        if mode == "-1":
            Zh, Jh = Z.download(), Jct.download()

            def c_host(out, x, Jh=Jh):
                out[:m_lin] = Jh[:, :m_lin].T @ x - b
                if has_ball:
                    out[m_lin] = x[:n] @ x[:n] - 0.4 * n - x[n]
            nr0 = R.NR(Zh, S, Vt, 1e-9, 50, R.NRWork(m), False, R.InequalityData())
            xn0, cv0 = np.zeros(N), np.zeros(m)
            f0, i0, _ = R.retract_(cv0, xn0, c_host, xs_h + pert, xs_h, nr0)
            results["oracle"] = (f0, i0, xn0, cv0)
        ctx.close()
    f0, i0, xn0, cv0 = results["oracle"]
    assert f0 == 0
    for mode in ("-1", "0"):
        flag, it, xn, cv = results[mode]
        assert (flag, it) == (f0, i0), mode
        assert np.linalg.norm(xn - xn0) <= 1e-11 * np.linalg.norm(xn0), mode
        assert np.max(np.abs(cv)) < 1e-9
    # the one-stream kernels are a different summation order than the two-stream one: they must not be the same code path
    assert not np.array_equal(results["-1"][2], results["0"][2]) or m < 4


@pytest.mark.parametrize("m", [7, 32])
def test_pcg_one_pass_iteration_matches_two_pass_kernels(dev_ctx, monkeypatch, m):
    """lfpsqp_pcg's default iteration makes one pass over Jct (J p+ = (J r - alpha J z) + beta J p, csrc/pcg.hip);
    LFPSQP_ONEPASS=-1 selects the two-pass kernels.  Same counts / flags, solutions equal to rounding, both equal to
    the oracle's pcg! (src/retractions.jl:179-246)."""
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    n = 2300
    rng = np.random.default_rng(17 + m)
    Jr = rng.standard_normal((m, n))
    bh = rng.standard_normal(n)
    for mu, tol, maxiter in ((1e-1, 1e-6, 100), (1e-4, 1e-9, 100), (1e-2, 1e-12, 6)):
        x0h, r0 = np.zeros(n), bh.copy()
        f0, i0 = R.pcg_(mu, Jr, R.no_precondition, x0h, r0, np.zeros(n), np.zeros(n), np.zeros(m), tol, maxiter)
        res = {}
        for mode in ("-1", "0"):
            monkeypatch.setenv("LFPSQP_ONEPASS", mode)
            ctx = L.Context(0, dev_ctx.L)
            Jd = ctx.matrix(n, m, np.asfortranarray(Jr.T))
            w = L.ProjPenaltyWork(ctx, m, n, False)
            x, r = ctx.vector(n), ctx.vector(n, bh)
            flag, i = L.pcg_(mu, _JacPlain(Jd, w), L.no_precondition, x, r, w.p, w.z, None, tol, maxiter)
            res[mode] = (flag, i, x.download(), r.download())
            ctx.close()
        for mode in ("-1", "0"):
            flag, i, xh, rh = res[mode]
            assert (flag, i) == (f0, i0), (mode, mu)
            np.testing.assert_allclose(xh, x0h, atol=1e-10 * max(1.0, np.abs(x0h).max()))
            np.testing.assert_allclose(rh, r0, atol=1e-9 * max(1.0, np.abs(bh).max()))
        assert not np.array_equal(res["-1"][2], res["0"][2])


@pytest.mark.parametrize("nb,bounds,mcols", [(2, False, 0), (3, True, 0), (4, True, 0), (4, False, 0), (5, False, 0), (8, True, 0), (8, False, 0),
                                             (13, True, 0), (16, True, 0), (16, False, 0),
                                             # 65 .. 132 generator columns: the LDS-DMA form of the matrix-core step (nrb2_kernel); m = 128 with
                                             # the ball column (129 columns: 17 DMA pieces) is config 4's shape, m = 127 + ball fills 128 exactly
                                             (16, True, 128), (7, False, 128), (16, False, 127), (5, True, 69), (12, True, 100),
                                             # 133 .. 528 generator columns: the WIDE matrix-core form (nrb_mfma_wide_kernel: the four waves of a
                                             # workgroup share a 16-row tile and split its columns), up to eight trial points per pass; m + ball =
                                             # 513 columns is config 5's shape with ball and bounds (33 column groups per wave), 141 / 385 exercise
                                             # a ragged last group inside the second / in the first register of a wave
                                             (4, True, 300), (3, False, 260), (8, True, 300), (8, False, 512), (6, True, 512), (8, True, 140),
                                             (5, False, 384), (8, True, 527),
                                             # beyond 528 columns: the VALU wide form of the one-pass kernel, four trial points per pass
                                             (4, True, 600)])
def test_batched_newton_retractions_equal_one_by_one(dev_ctx, nb, bounds, mcols):
    """lfpsqp_retract_nr_batch: nb trial points of one linesearch retracted together (one pass over Jct per Newton step for all of
    them) give, trial by trial, what lfpsqp_retract_nr gives one by one -- including trials that converge at different
    iterations and one that fails (maxiter).  Two trials: the VALU form of the one-pass kernel; 3 ... 16: the step on the matrix
    cores (nrbatch.h: both products of src/retractions.jl:141 / :146 as v_mfma_f64_16x16x4_f64 contractions over the stacked trials)."""
    ctx = dev_ctx
    if _is_emu(ctx) and mcols in (384, 527, 600) and nb != 8:
        pytest.skip("a wide-form shape the emulator covers through its neighbours (300 / 512 / 140 columns); runs on the GPU")
    n, m = (1500, 7) if _is_emu(ctx) else (200_000, 31)
    if mcols:
        n, m = ((700 if mcols >= 500 else 1100) if _is_emu(ctx) else 150_000), mcols      # (the emulated wide matrix-core step: 40 s at 1100 x 512)
    P0 = synth.BallBoxProblem(n, m)
    N, M = n + 1, m + 1
    Jct = ctx.matrix(N, M).hash_fill(1, 0, n, 1.0, n, m)
    xl, xu = (P0.xl, P0.xu) if bounds else (np.full(n, -np.inf), np.full(n, np.inf))
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=xl, xu=xu)
    # one outer iteration's worth of state through the driver's own setup path: stop after the first tangent setup
    x0 = 0.9 * synth.hash_vector(2, n) + 0.05
    captured = {}
    import lfpsqp_jl_amd.linesearch as LS
    orig = LS.armijo_

    def spy(xnew, x, nn, d, g, f, fval, retract_method, cval, c_, param, work):
        captured.update(x=x, d=d, method=retract_method, c_=c_, m=len(cval), work=work, xnew=xnew)
        raise StopIteration
    LS.armijo_ = spy
    import sys
    OPT = sys.modules["lfpsqp_jl_amd.optimize"]            # (the package attribute `optimize` is the function, not the module)
    OPT.armijo_ = spy
    try:
        with pytest.raises(StopIteration):
            P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3))
    finally:
        LS.armijo_ = orig
        OPT.armijo_ = orig
    x, d, method, c_, mm = captured["x"], captured["d"], captured["method"], captured["c_"], captured["m"]
    assert isinstance(method, L.NR)
    method.maxiter = 40                                    # so that the largest step fails while the small ones converge
    alphas = ([64.0, 0.02, 2e-3, 1e-4] + [0.05 * 0.5 ** k for k in range(12)])[:nb]
    assert L.retract_nr_batch_width_(c_, method) == 4      # the default: the exact batch (tests/test_exact_batch.py)
    ctx.set_nr_batch_mode(True)                            # this test: the matrix-core batch (opt-in)
    assert L.retract_nr_batch_width_(c_, method) == (16 if M <= 132 else (8 if M <= 528 else 4))
    xts, xns = captured["work"].batch_vectors(nb)
    for a, xt in zip(alphas, xts):
        L.waxpby(1.0, x, a, d, xt)
    cvs = np.zeros((nb, mm))
    got = L.retract_nr_batch_(cvs, xns, c_, xts, x, method)
    assert got is not None
    one = captured["xnew"]
    flags = []
    for b in range(nb):
        cv = np.zeros(mm)
        fl, it, _ = L.retract_(cv, one, c_, xts[b], x, method)
        flags.append(fl)
        assert (got[b][0], got[b][1]) == (fl, it), (b, got[b], fl, it)
        xa = xns[b].download2() if bounds else xns[b].download()
        xb_ = one.download2() if bounds else one.download()
        if fl == 0:
            # The matrix-core form (nb > 4) sums in another order than the single-trial kernel.  A trial that starts far away (alpha = 64:
            # iterates of size 1e3, constraint values of size 1e9 in its first steps) and needs dozens of Broyden steps amplifies that
            # rounding along its path -- it still lands on c = 0 after the same number of steps, at a point 1e-9 (relative) along the manifold
            # from the other kernel's; trials that converge from nearby agree to 1e-12.
            rel = 1e-12 if (nb <= 2 or it <= 20) else 1e-8
            if mcols >= 512:
                rel = max(rel, 1e-11)      # (n / m = 2 on the emulator: the Newton steps of this nearly square block amplify rounding a little more)
            np.testing.assert_allclose(xa, xb_, rtol=0, atol=rel * max(1.0, np.abs(xb_).max()))
            if rel == 1e-12:
                np.testing.assert_allclose(cvs[b], cv, atol=1e-8)      # c(xnew) ~ 0: differences of rounding size in a sum over n terms
            else:
                assert np.abs(cvs[b]).max() < method.tol and np.abs(cv).max() < method.tol      # both converged; see above
    assert 0 in flags
    if _is_emu(ctx) and mcols != 140:                          # (at 141 columns on 1101 rows every trial happens to need five steps)
        assert len(set(g[1] for g in got)) > 1, got            # the trials really finished at different iterations


@pytest.mark.parametrize("matrix_cores", [False, True])
def test_armijo_with_batched_trial_retractions_is_the_same_search(dev_ctx, matrix_cores):
    """DeviceOptions.ls_batch (ctx.options): after the first failed retraction of an Armijo search the next trial steps are retracted
    together; the search consumes them in the reference's order (src/linesearch.jl:32-89).  In the default EXACT mode the accepted
    step, every count and the iterates are those of the one-by-one search -- strictly (bitwise: tests/test_exact_batch.py).  With the
    matrix-core batch (opt-in) the counts of FAILING searches may differ on the GPU and the run may fork (the helper says so)."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (150, 4) if emu else (4000, 16)
    mr = 30 if emu else 100                               # (shorter failed retractions keep the emulator run short)
    P0 = synth.BallBoxProblem(n, m)                       # from P0.x0 the first linesearches fail repeatedly (config 4's regime)
    res = {}
    ctx.options.ls_batch_matrix_cores = matrix_cores
    for k in (1, 4):
        Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
        P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
        tr = []
        ctx.options.ls_batch = k
        x, obj, lam, ti = P.optimize(P0.x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=2 if emu else 6,
                                                           maxiter_retract=mr), trace=tr)
        res[k] = (tr, x, ti)
    tr1, x1, ti1 = res[1]
    tr4, x4, ti4 = res[4]
    assert ti1.iter == ti4.iter and len(tr1) == len(tr4)
    assert any((t.get('retract_iter1') or 0) >= mr for t in tr1)           # the regime with failed retractions was reached
    fork = _compare_traces(tr4, tr1, rtol=1e-11, failed_retractions_may_differ=matrix_cores and not emu)
    if fork is None:
        assert np.linalg.norm(x4 - x1) <= 1e-10 * np.linalg.norm(x1)
    else:
        assert matrix_cores


@pytest.mark.parametrize("matrix_cores", [False, True])
def test_exact_linesearch_with_batched_shrinking_is_the_same_search(dev_ctx, matrix_cores):
    """DeviceOptions.ls_batch with linesearch = exact: the trial steps of the SHRINKING phase (src/linesearch.jl:176-208, a fixed
    sequence a_c*phi1^k from the same x -- the phase that runs when the first trial retraction fails) are retracted together
    and consumed in the reference's order: accepted step, counts and iterates of the one-by-one search, and of the oracle
    (strictly in the default exact mode; the matrix-core batch may differ in the counts of failing searches on the GPU)."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (150, 4) if emu else (4000, 16)
    mr = 30 if emu else 100
    maxiter = 2 if emu else 4
    P0 = synth.BallBoxProblem(n, m)                       # from P0.x0 the first linesearches fail repeatedly (config 4's regime)
    res = {}
    ctx.options.ls_batch_matrix_cores = matrix_cores
    for k in (1, 4):
        Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
        P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
        tr = []
        ctx.options.ls_batch = k
        x, obj, lam, ti = P.optimize(P0.x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter, maxiter_retract=mr,
                                                           linesearch=L.LinesearchOption.exact), trace=tr)
        res[k] = (tr, x, ti)
    tr1, x1, ti1 = res[1]
    tr4, x4, ti4 = res[4]
    assert ti1.iter == ti4.iter and len(tr1) == len(tr4)
    assert any((t.get('retract_iter1') or 0) >= mr for t in tr1)           # the regime with failed retractions was reached
    fork = _compare_traces(tr4, tr1, rtol=1e-11, failed_retractions_may_differ=matrix_cores and not emu)
    if fork is None:
        assert np.linalg.norm(x4 - x1) <= 1e-10 * np.linalg.norm(x1)
    else:
        assert matrix_cores
    if emu:
        tr0 = []
        R.optimize(P0.f, P0.c_, P0.d_, P0.x0, P0.xl, P0.xu, P0.m, P0.p,
                   R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter, maxiter_retract=mr,
                                  linesearch=R.LinesearchOption.exact), derivatives=P0.derivatives(), trace=tr0)
        assert _compare_traces(tr4, tr0) is None      # (1e-10; measured: 8.8e-14)


def _sep_host(kind, a, c):
    """phi, phi', phi'' of lfpsqp_separable on the host (the oracle side of SeparableLinearBallBox)"""
    def phi(x):
        t = x - c
        if kind == 0: return a * t * t
        if kind == 1: return a * t ** 4 + t * t
        return a * (np.sqrt(1.0 + t * t) - 1.0)

    def d1(x):
        t = x - c
        if kind == 0: return 2.0 * a * t
        if kind == 1: return 4.0 * a * t ** 3 + 2.0 * t
        return a * t / np.sqrt(1.0 + t * t)

    def d2(x):
        t = x - c
        if kind == 0: return 2.0 * a * np.ones_like(t)
        if kind == 1: return 12.0 * a * t * t + 2.0
        return a / np.sqrt(1.0 + t * t) ** 3
    return phi, d1, d2


@pytest.mark.parametrize("kind,with_ball_box", [(1, False), (2, True), (1, True)])
def test_separable_objective_problem_class(dev_ctx, kind, with_ball_box):
    """SeparableLinearBallBox (a second device-resident problem class): non-quadratic separable objectives with
    per-variable parameters under dense equalities (+ ball in slack form + box).  f, grad! and the diagonal Lagrangian Hessian are
    device kernels; the trajectory -- truncated-Newton iteration counts included, the Hessian now changes with x -- is the
    oracle's with the same functions evaluated in numpy."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (240, 4) if emu else (6000, 16)
    maxiter = 4 if emu else 12
    a = 0.5 + synth.hash_vector(21, n) ** 2
    c = 0.3 * synth.hash_vector(22, n)
    phi, d1, d2 = _sep_host(kind, a, c)
    P0 = synth.BallBoxProblem(n, m)
    x0 = 0.9 * synth.hash_vector(2, n) + 0.1 * P0.x0 if with_ball_box else synth.hash_vector(2, n)
    f = lambda x: float(np.sum(phi(x[:n])))

    def grad_(g, x):
        g[:n] = d1(x[:n])

    tr0, tr = [], []
    if with_ball_box:
        dv0 = P0.derivatives()

        def hlv_(dest, src, x, lam):
            dest[:] = (d2(x) + 2.0 * lam[m]) * src
        xr, objr, lamr, tir = R.optimize(f, P0.c_, P0.d_, x0, P0.xl, P0.xu, m, 1,
                                         R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter, tn_kappa=1e-6),
                                         derivatives=R.Derivatives(grad_=grad_, hess_lag_vec_=hlv_, jac_c_=dv0.jac_c_, jac_d_=dv0.jac_d_), trace=tr0)
        Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
        P = L.SeparableLinearBallBox(ctx, n, m, Jct, P0.eq.b, kind, a, c, R2=P0.R2, xl=P0.xl, xu=P0.xu)
    else:
        def hlv_(dest, src, x, lam):
            dest[:] = d2(x) * src
        xr, objr, lamr, tir = R.optimize(f, grad_, P0.eq.c_, P0.eq.jac_, hlv_, x0, None, None, m,
                                         R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=maxiter, tn_kappa=1e-6), trace=tr0)
        P = L.SeparableLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), P0.eq.b, kind, a, c)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=maxiter, tn_kappa=1e-6), trace=tr)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    assert any((t.get('tn_iter') or 0) > 1 for t in tr0)                       # the Newton systems are not solved in one iteration here
    assert _compare_traces(tr, tr0) is None               # (1e-10; measured on the GPU: 4.1e-16)
    np.testing.assert_allclose(obj, objr, rtol=1e-11)
    np.testing.assert_allclose(lam, lamr, rtol=1e-6, atol=1e-9)
