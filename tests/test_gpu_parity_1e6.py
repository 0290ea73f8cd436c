"""The n = 1e6 point of the parity protocol (trajectories at n in {1e3, 1e4, 1e6}; callback period 1,
src/optimize.jl:432-434): `optimize` on the GPU against the numpy oracle's run of the same problem -- x after every outer iteration within
1e-10 relative, equal iteration counts, flags, step types, accepted steps, retraction iterations.  Config 3 at the headline m = 128 with
both retractions; config 4 (ball in slack form + four-way bounds, Newton retraction) in the strict regime (no failed trial retraction) at
m = 16 and m = 128.  The oracle needs 20 ... 50 s per case on the box's host cores, which is why these cases exist on the GPU only."""
import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import synth

from .test_capi_retractions import _compare_traces

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_lib):
    c = L.Context(0, gpu_lib)
    yield c
    c.close()


@pytest.mark.parametrize("do_project_retract", [False, True])
def test_config3_trajectory_at_1e6_by_128(ctx, do_project_retract):
    """BASELINE configs[2]'s constraint block at n = 1e6, m = 128: Newton retraction (src/retractions.jl:75-177) and the reference's default
    ProjPenalty + pcg! (:179-246, :265-441)."""
    n, m = 1_000_000, 128
    prob0, x0 = synth.config3(n, m)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, m,
                                     R.LFPSQPParams(do_project_retract=do_project_retract, disp=R.DisplayOption.off), trace=tr0)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), prob0.b)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=do_project_retract, disp=L.DisplayOption.off), trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    _compare_traces(tr, tr0, rtol=1e-10, pcg_slack=2 if do_project_retract else 0)
    dev = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    print(f"[parity n=1e6 m=128 {'PP' if do_project_retract else 'NR'}] {ti.iter} outer iteration(s), |x - x_oracle| / |x_oracle| = {dev:.2e}")
    assert dev <= 1e-10
    np.testing.assert_allclose(obj, objr, rtol=1e-12)
    np.testing.assert_allclose(lam, lamr, rtol=1e-8, atol=1e-12)


def test_config3_exact_linesearch_at_1e6_by_128(ctx):
    """exact_linesearch! (src/linesearch.jl:107-339: bracketing by golden-section growth, then golden-section search, every trial point
    retracted) at the protocol's size: config 3's block at n = 1e6, m = 128 with `linesearch = exact`, Newton retraction, against the oracle --
    the accepted step of every search, the cumulative Newton-iteration counts and the iterates within 1e-10."""
    n, m = 1_000_000, 128
    prob0, x0 = synth.config3(n, m)
    x0 = x0 + 0.3 * synth.hash_vector(9, n)                       # (off the constraint manifold AND not a multiple of ones: the search has work to do)
    tr0, tr = [], []
    p0 = R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, linesearch=R.LinesearchOption.exact, maxiter=3)
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, m, p0, trace=tr0)
    P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m).hash_fill(1), prob0.b)
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, linesearch=L.LinesearchOption.exact, maxiter=3),
                                 trace=tr)
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    assert _compare_traces(tr, tr0, rtol=1e-10) is None
    dev = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    print(f"[parity n=1e6 m=128 config 3, exact linesearch] {ti.iter} outer iteration(s) ({ti.condition.name}), accepted steps {[t.get('alpha') for t in tr[:-1]]}, "
          f"Newton iterations {[t.get('retract_iter1') for t in tr[:-1]]}, |x - x_oracle| / |x_oracle| = {dev:.2e}")
    assert dev <= 1e-10
    np.testing.assert_allclose(obj, objr, rtol=1e-12)


@pytest.mark.parametrize("m", [16, 128])
def test_config4_strict_trajectory_at_1e6(ctx, m):
    """BASELINE configs[3]'s shape at n = 1e6, m = 16 and at the protocol's m = 128, from a start near the feasible set (no trial retraction
    reaches the reference's 100-iteration limit -- asserted on the oracle's trace -- so there is no chaotic regime and the comparison is
    strict): eight (m = 128: three) outer iterations, every count and accepted step equal, iterates within 1e-10 after every one of them.  (From this start
    the run is in its negative-curvature phase -- projcg! leaves through src/projcg.jl:77-82 and the step is the unit vector of :79 --
    and would take some 1e5 such iterations to leave it: eight of them is what the host's dgesvd affords, 8 s each at m = 128.)"""
    n, maxiter = 1_000_000, (8 if m == 16 else 3)      # (m = 128 runs to convergence in the next test)
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
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    assert _compare_traces(tr, tr0, rtol=1e-10) is None
    worst = max(np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x']) for a, b in zip(tr, tr0))
    print(f"[parity n=1e6 m={m} config 4] {ti.iter} outer iterations, worst iterate deviation {worst:.2e}")
    np.testing.assert_allclose(obj, objr, rtol=1e-11)
    np.testing.assert_allclose(lam, lamr, rtol=1e-7, atol=1e-10)


def test_config4_complete_trajectory_to_convergence_at_1e6_by_128(ctx):
    """BASELINE configs[3] at the protocol's n = 1e6, m = 128, ALL outer iterations to `kkt_tol`, strictly.  From the config's own start
    the first searches are chaotic (trial retractions that run into the 100-iteration limit); once the run is past them it is not.  So: the
    device first runs the whole problem from P0.x0; the iterate from which on no linesearch contains a failed retraction (at most the last
    five outer iterations: one host dgesvd of a 2e6 x 129 matrix each) becomes the start of BOTH runs -- the oracle's and a fresh device
    run -- and those are compared strictly: equal lengths, counts, step types, accepted alpha, Newton iterations, termination, iterates
    within 1e-10 after every outer iteration (src/optimize.jl:432-434)."""
    n, m = 1_000_000, 128
    P0 = synth.BallBoxProblem(n, m)

    def device_run(x0):
        Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
        P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
        tr = []
        out = P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off), trace=tr)
        Jct.free()
        return out, tr

    (xa, obja, lama, tia), tra = device_run(P0.x0)
    assert tia.condition.name == "kkt_tol"
    its = [t.get('retract_iter1') or 0 for t in tra]
    k = len(tra) - 1                                      # (the last entry is the converged iterate: no linesearch)
    while k > 0 and its[k - 1] < 100 and len(tra) - 1 - (k - 1) <= 5:
        k -= 1
    assert len(tra) - 1 - k >= 2, (its, k)                # at least two outer iterations to go
    x0 = tra[k]['x'][:n].copy()
    tr0 = []
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, P0.m, P0.p,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off), derivatives=P0.derivatives(), trace=tr0)
    assert all((t.get('retract_iter1') or 0) < 100 for t in tr0), [t.get('retract_iter1') for t in tr0]      # the premise of strictness
    (x, obj, lam, ti), tr = device_run(x0)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name == "kkt_tol"
    assert _compare_traces(tr, tr0, rtol=1e-10) is None
    worst = max(np.linalg.norm(a['x'] - b['x']) / np.linalg.norm(b['x']) for a, b in zip(tr, tr0))
    print(f"[parity n=1e6 m=128 config 4, to convergence] start = outer iterate {k} of {len(tra) - 1} of the run from the config's x0; {ti.iter} outer "
          f"iterations to {ti.condition.name}, Newton iterations {[t.get('retract_iter1') for t in tr[:-1]]}, truncated-Newton iterations "
          f"{[t.get('tn_iter') for t in tr[:-1]]}, worst iterate deviation {worst:.2e}, |x - x_oracle| / |x_oracle| = "
          f"{np.linalg.norm(x - xr) / np.linalg.norm(xr):.2e}, f = {obj[-1]:.10e} / {objr[-1]:.10e}")
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(obj, objr, rtol=1e-11)
    np.testing.assert_allclose(lam, lamr, rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("do_project_retract", [False, True])
def test_nonlinear_class_with_streamed_gradients_at_1e6_by_16(ctx, do_project_retract):
    """The nonlinear constraint class c(x) = A' phi(x) + qw x'x - b (mixed kinds and the common quadratic term) at n = 1e6, m = 16 with its
    gradients STREAMED through a matrix view (jac! writes two n-vectors, src/autodiff_generators.jl:60-66 writes n x m doubles), from a start on
    the manifold: `optimize` against the oracle's run with host callables of the same functions -- counts, step types, retraction iterations
    equal, iterates within 1e-10 after every outer iteration; Newton-Raphson and the reference's default ProjPenalty retraction."""
    from .test_elementwise import _trace_compare, ew_callables
    n, m, maxiter = 1_000_000, 16, 5
    rng = np.random.default_rng(61)
    Ah = np.asfortranarray(rng.standard_normal((n, m)) / np.sqrt(n))
    kind = rng.integers(0, 3, n).astype(np.float64)
    qw = 0.01 * rng.standard_normal(m)
    bh = 0.1 * rng.standard_normal(m)
    x0 = 0.2 * rng.standard_normal(n)
    target = 0.5 * rng.standard_normal(n)
    cv0 = np.zeros(m)
    ew_callables(Ah, kind, qw, bh)[0](cv0, x0)
    bh = bh + cv0                                                                     # x0 on the manifold
    c_, jac_, hdiag = ew_callables(Ah, kind, qw, bh)
    cons = L.ElementwiseConstraints(ctx, ctx.matrix(n, m, Ah), bh, kind=kind, qw=qw)
    assert cons.streamed
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target)
    tr, tr0 = [], []
    xd, obj, lam, ti = prob.optimize(x0, L.LFPSQPParams(do_project_retract=do_project_retract, maxiter=maxiter, disp=L.DisplayOption.off), trace=tr)

    f = lambda xx: float(np.sum((xx[:n] - target) ** 2))

    def grad_(g, xx):
        g[:n] = 2.0 * (xx[:n] - target)

    def hlv_(dest, src, xx, lam_):
        dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]

    xr, objr, lamr, tir = R.optimize_core(f, grad_, c_, jac_, hlv_, x0, None, None, m,
                                          R.LFPSQPParams(do_project_retract=do_project_retract, maxiter=maxiter, disp=R.DisplayOption.off), trace=tr0)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    _trace_compare(tr, tr0)
    dev = np.linalg.norm(xd - xr) / np.linalg.norm(xr)
    print(f"[parity n=1e6 m=16 nonlinear class, streamed gradients, {'PP' if do_project_retract else 'NR'}] {ti.iter} outer iterations, "
          f"|x - x_oracle| / |x_oracle| = {dev:.2e}")
    assert dev <= 1e-10
    np.testing.assert_allclose(obj, objr, rtol=1e-10)
