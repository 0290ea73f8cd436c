"""Trajectory parity at BASELINE's FULL size ("at full size for as many outer iterations as host RAM allows"; the reference's
trajectory hook is the per-iteration callback, src/optimize.jl:432-434).

  * projcg! (src/projcg.jl:40-121) at n = 1e7, m = 128: 30 iterations on the GPU against oracle/projcg_port.c on the box's host cores --
    the same orthonormal basis on both sides, equal iteration count, iterates within 1e-10 relative;
  * config 3 (random dense linear equalities, n = 1e7, m = 128): `optimize` on the GPU against the numpy oracle's run of the same problem --
    x after every outer iteration within 1e-10, equal counts / step types / accepted steps.  The oracle factorises a 1e7 x 128 matrix with
    LAPACK dgesvd twice (as the reference does) and holds four 10 GB matrices: it needs ~100 GB of host memory and a few minutes of host
    time, so the case is skipped -- with the reason printed -- on a box that does not have them.
Measured deviations are printed (run with -s) and recorded in profiles/."""
import os
import time

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import port, synth

from .test_capi_retractions import _compare_traces

pytestmark = pytest.mark.gpu

N, M = 10_000_000, 128


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


@pytest.fixture(scope="module")
def ctx(gpu_lib):
    c = L.Context(0, gpu_lib)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ksvd_memo():
    """The oracle factorises the SAME 1e7 x 128 matrix four times in the two config-3 cases below (linear constraints: Jct is constant, the
    reference calls ksvd! at every outer iteration and the cases differ only in the retraction) -- 40 s of host dgesvd each.  LAPACK is
    deterministic for identical input, so the checker keeps the factors of a matrix it has seen (keyed by a hash of its bytes) and copies
    them out the next time: same oracle results, a quarter of its time.  Test infrastructure only."""
    import xxhash
    orig, cache = R.ksvd_, {}

    def ksvd(A, U, S, VT):
        if A.size < 10 ** 8 or not A.flags.f_contiguous:
            return orig(A, U, S, VT)
        key = (A.shape, xxhash.xxh3_64_hexdigest(memoryview(A.T)))
        if key not in cache:
            cache.clear()                                  # (one 10 GB entry at a time)
            orig(A, U, S, VT)
            cache[key] = (U.copy(order='F'), S.copy(), VT.copy(order='F'))
            return
        u, s, vt = cache[key]
        U[:, :] = u
        S[:] = s
        VT[:, :] = vt
        A[:, :] = np.nan                                   # "destroyed", as the real call leaves it
    R.ksvd_ = ksvd
    yield cache
    R.ksvd_ = orig
    cache.clear()


def test_projcg_against_the_c_oracle_port_at_full_size(ctx):
    """n = 1e7, m = 128, 30 iterations (tol = 0: both sides run to the iteration limit), kappa(A) = 9."""
    need = 8.0 * N * (M + 12) / 2 ** 30 + 4
    if _mem_available_gb() < need:
        pytest.skip(f"host memory: {_mem_available_gb():.0f} GB available, the C port needs {need:.0f} GB at n = 1e7, m = 128")
    Z = ctx.matrix(N, M).hash_fill(1)
    L.orthonormalize_(Z)
    Uh = Z.download()                                   # the SAME orthonormal basis for both sides (10.2 GB)
    a = port.hash_vector(3, N, 0, 4.0, 5.0)
    b = port.hash_vector(4, N)
    port.lib().port_set_num_threads(port.usable_cpus())
    t0 = time.perf_counter()
    x0, l0, it0, nr0 = port.projcg(a, Uh, b, None, 0.0, 30)
    t_cpu = time.perf_counter() - t0
    x, lam = ctx.vector(N), ctx.vector(M)
    ad, bd = ctx.vector(N).hash_fill(3, 0, 4.0, 5.0), ctx.vector(N).hash_fill(4)
    ctx.sync()
    t0 = time.perf_counter()
    it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ad), L.DeviceBasis(Z), bd, None, tol=0.0, maxit=30)
    t_gpu = time.perf_counter() - t0
    xd = x.download()
    dev = np.linalg.norm(xd - x0) / np.linalg.norm(x0)
    dl = np.abs(lam.download() - l0).max()
    print(f"[parity n=1e7 m=128 projcg! vs C port] iterations {it}/{it0}, nr {nr:.6e}/{nr0:.6e}, |x - x_port| / |x_port| = {dev:.2e}, "
          f"max |lambda - lambda_port| = {dl:.2e}; C port {t_cpu:.1f} s on {port.usable_cpus()} host cores, GPU {t_gpu * 1e3:.0f} ms")
    assert it == it0 == 30 and nr == pytest.approx(nr0, rel=1e-8)
    assert dev <= 1e-10
    assert dl <= 1e-9
    Z.free()


@pytest.mark.parametrize("do_project_retract", [False, True])
def test_config3_trajectory_at_full_size(ctx, ksvd_memo, do_project_retract):
    """BASELINE configs[2] itself: n = 1e7, m = 128, f = x'x, x0 = ones; one outer iteration to kkt_tol -- with the Newton retraction
    (src/retractions.jl:75-177) and with the reference's default, ProjPenalty + pcg! (:179-441; the cumulative inner count may differ by two,
    see _compare_traces)."""
    if os.environ.get("LFPSQP_SKIP_FULL_ORACLE"):
        pytest.skip("LFPSQP_SKIP_FULL_ORACLE is set")
    avail = _mem_available_gb()
    if avail < 100.0:
        pytest.skip(f"host memory: {avail:.0f} GB available, the numpy oracle of config 3 at n = 1e7 needs ~100 GB "
                    "(Jc, Jct, U and dgesvd's copy of a 10.2 GB matrix next to the problem's own)")
    port.lib().port_set_num_threads(port.usable_cpus())
    t0 = time.perf_counter()
    Jh = port.hash_matrix(1, N, M)                       # == synth.hash_matrix(1, N, M), bit for bit (tests/test_golden.py), in seconds
    xstar = synth.hash_vector(2, N)
    prob0 = synth.QuadLinearProblem(Jh, port.gemv_t(Jh, xstar))
    x0 = np.ones(N)
    tr0, tr = [], []
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, M,
                                     R.LFPSQPParams(do_project_retract=do_project_retract, disp=R.DisplayOption.off), trace=tr0)
    t_cpu = time.perf_counter() - t0
    P = L.QuadLinearBallBox(ctx, N, M, ctx.matrix(N, M).hash_fill(1), prob0.b)
    t0 = time.perf_counter()
    x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(do_project_retract=do_project_retract, disp=L.DisplayOption.off), trace=tr)
    t_gpu = time.perf_counter() - t0
    assert ti.condition.name == tir.condition.name and ti.iter == tir.iter
    _compare_traces(tr, tr0, rtol=1e-10, pcg_slack=2 if do_project_retract else 0)
    dev = np.linalg.norm(x - xr) / np.linalg.norm(xr)
    print(f"[parity n=1e7 m=128 config 3 {'PP' if do_project_retract else 'NR'}] {ti.iter} outer iteration(s) ({ti.condition.name}), "
          f"|x - x_oracle| / |x_oracle| = {dev:.2e}, objective {obj[-1]:.12e} / {objr[-1]:.12e}; oracle {t_cpu:.0f} s on {port.usable_cpus()} host cores, "
          f"GPU {t_gpu:.2f} s (with the trace's downloads)")
    assert dev <= 1e-10
    np.testing.assert_allclose(obj, objr, rtol=1e-12)
    np.testing.assert_allclose(lam, lamr, rtol=1e-8, atol=1e-12)


def test_config4_first_outer_iteration_at_full_size(ctx):
    """BASELINE configs[3] at its stated size: n = 1e7, m = 128 equalities + the ball in slack form + four-way bounds, Newton retraction --
    one outer iteration (both runs are stopped by the trajectory callback behind it) from a start near the feasible set (the strict regime: no trial retraction reaches the 100-iteration limit, asserted
    on the oracle's trace) against the numpy oracle: the tangent setup with bounds (dgesvd of the 2e7 x 129 projected matrix on the host,
    the weighted Gram factorisation here), the projected step and multipliers (src/optimize.jl:312-343, src/inequality_helper.jl:286-308),
    projcg! with the augmented Hessian (:144-158; it leaves through negative curvature, src/projcg.jl:77-82), the Newton retraction with
    y_retract! (src/retractions.jl:133-165), Armijo -- counts equal, x after the iteration within 1e-10.  The oracle holds Jc, Jct, the
    projected copy, U and dgesvd's work copy of a 20.6 GB matrix: skipped, with the reason, below 200 GB of host memory."""
    if os.environ.get("LFPSQP_SKIP_FULL_ORACLE"):
        pytest.skip("LFPSQP_SKIP_FULL_ORACLE is set")
    avail = _mem_available_gb()
    if avail < 200.0:
        pytest.skip(f"host memory: {avail:.0f} GB available, the numpy oracle of config 4 at n = 1e7, m = 128 needs ~190 GB "
                    "(five copies of a 2e7 x 129 matrix next to the problem's own 10 GB)")
    port.lib().port_set_num_threads(port.usable_cpus())
    t0 = time.perf_counter()
    Jh = port.hash_matrix(1, N, M)
    P0 = synth.BallBoxProblem(N, M, Jct=Jh, b=port.gemv_t(Jh, synth.hash_vector(2, N)))
    x0 = 0.9 * synth.hash_vector(2, N) + 0.1 * P0.x0
    tr0, tr = [], []

    class _AfterFirstIteration(Exception):
        pass
    got0, got = {}, {}

    def cb0(i, x):             # the reference's trajectory hook (src/optimize.jl:432-434): x after the outer iteration -- and stop there: the
        got0['x'] = np.array(x, copy=True)      # second tangent setup (another 90 s of host SVD) would only feed the termination tests
        raise _AfterFirstIteration
    with pytest.raises(_AfterFirstIteration):
        R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, P0.m, P0.p,
                   R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=5, callback=cb0, callback_period=1),
                   derivatives=P0.derivatives(), trace=tr0)
    t_cpu = time.perf_counter() - t0
    assert len(tr0) == 1 and (tr0[0].get('retract_iter1') or 0) < 100                   # the premise of strictness
    b = P0.eq.b.copy()
    xl, xu, R2 = P0.xl, P0.xu, P0.R2
    del P0, Jh                                                                          # (10 GB of host memory back before the device run's downloads)
    Jct = ctx.matrix(N + 1, M + 1).hash_fill(1, 0, N, 1.0, N, M)
    P = L.QuadLinearBallBox(ctx, N, M, Jct, b, R2=R2, xl=xl, xu=xu)
    t0 = time.perf_counter()

    def cb(i, xdev):
        got['x'] = xdev.download2()
        raise _AfterFirstIteration
    with pytest.raises(_AfterFirstIteration):
        P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=5, callback=cb, callback_period=1), trace=tr)
    t_gpu = time.perf_counter() - t0
    assert _compare_traces(tr, tr0, rtol=1e-10) is None                                  # the start, and every count / step type / accepted step of the iteration
    xr, x = got0['x'], got['x']                                                          # [x; y] after the iteration
    assert x.shape == xr.shape
    worst = max(np.linalg.norm(tr[0]['x'] - tr0[0]['x']) / np.linalg.norm(tr0[0]['x']), np.linalg.norm(x - xr) / np.linalg.norm(xr))
    print(f"[parity n=1e7 m=128 config 4] one outer iteration: Newton iterations {tr[0].get('retract_iter1')}/{tr0[0].get('retract_iter1')}, "
          f"truncated-Newton iterations {tr[0].get('tn_iter')}/{tr0[0].get('tn_iter')}, alpha {tr[0].get('alpha')}/{tr0[0].get('alpha')}, worst iterate "
          f"deviation {worst:.2e}, multipliers max |lambda - lambda_oracle| {np.abs(tr[0]['lam_kkt'] - tr0[0]['lam_kkt']).max():.2e}; "
          f"oracle {t_cpu:.0f} s on {port.usable_cpus()} host cores, GPU {t_gpu:.2f} s (with the trace's downloads)")
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(tr[0]['lam_kkt'], tr0[0]['lam_kkt'], rtol=1e-7, atol=1e-10)
    assert tr[0]['fval'] == pytest.approx(tr0[0]['fval'], rel=1e-13)
    Jct.free()


def test_tridiagonal_hessian_on_one_pass_at_full_size(ctx):
    """projcg! with a tridiagonal A at n = 1e7, m = 128 (lfpsqp_projcg_tridiag): the one-pass iteration against the callback path with the same
    operator (lfpsqp_projcg_op: A d from lfpsqp_tridiag_mul, two passes over U) -- equal counts, iterates within 1e-10 -- and, independent of
    both loops, the properties of the solution: U'x = 0 and |P(A x - b)| equal to the reported residual norm (src/projcg.jl:97-103)."""
    Z = ctx.matrix(N, M).hash_fill(1)
    L.orthonormalize_(Z)
    U = L.DeviceBasis(Z)
    a = ctx.vector(N).hash_fill(3, 0, 4.5, 6.5)             # 2 .. 11
    off = ctx.vector(N).hash_fill(15, 0, 0.8, 0.0)          # both signs, diagonally dominant
    b = ctx.vector(N).hash_fill(4)
    T = L.TridiagonalOperator(0.0, a, off)
    work = L.ProjCGWork(ctx, N, M)
    x1, l1, x2, l2 = ctx.vector(N), ctx.vector(M), ctx.vector(N), ctx.vector(M)
    it1, nr1 = L.projcg_(x1, l1, T, U, b, None, tol=1e-6, maxit=200, work=work)
    T.fused = False
    it2, nr2 = L.projcg_(x2, l2, T, U, b, None, tol=1e-6, maxit=200, work=work)
    assert it1 == it2 and 10 < it1 < 200 and nr1 == pytest.approx(nr2, rel=1e-6)
    d = ctx.vector(N)
    L.waxpby(1.0, x1, -1.0, x2, d)
    dx = L.nrm2(d) / L.nrm2(x2)
    dl = np.abs(l1.download() - l2.download()).max()
    # the solution's own properties, from the primitives
    t = ctx.vector(M)
    L.gemv_t(Z, x1, t)
    in_range = np.abs(t.download()).max() / L.nrm2(x1)
    r = ctx.vector(N)
    T.mul_(r, x1)
    L.axpby(-1.0, b, 1.0, r)                                 # r = A x - b
    L.gemv_t(Z, r, t)
    L.gemv_n(Z, t, r, -1.0, 1.0)                             # g = r - U U'r
    pres = L.nrm2(r)
    print(f"[tridiagonal, full size] {it1} iterations, |x1 - x2| / |x2| = {dx:.2e}, |lambda1 - lambda2| = {dl:.2e}, |U'x| / |x| = {in_range:.2e}, "
          f"|P(Ax - b)| = {pres:.6e} against the reported {nr1:.6e}")
    assert dx <= 1e-10 and dl <= 1e-9
    assert in_range <= 1e-12
    assert pres == pytest.approx(nr1, rel=1e-2)          # (the loop's residual is a recurrence: after a reduction by 1e-9 it has drifted by rounding)
    for v in (Z, a, off, b, x1, x2, d, r):
        v.free()
