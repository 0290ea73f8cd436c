"""The EXACT batch of Newton retractions (lfpsqp_ctx_set_nr_batch_mode, LFPSQP_NR_BATCH_EXACT -- the default): the trial points of a
failing linesearch (src/linesearch.jl:57-60) share their passes over Jct, and every trial gets BIT FOR BIT what lfpsqp_retract_nr gives
it alone (src/retractions.jl:75-177): same flag, same Newton-iteration count, same iterate, same constraint values.  The batched search
is then the one-by-one search also where it is chaotic (a trial that fails in its 100th step or converges just inside it decides the
accepted step of BASELINE config 4's fifth search)."""
import sys

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import synth


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def capture_first_linesearch(ctx, n, m, bounds, x0=None, maxiter_retract=None):
    """Run `optimize` of the ball + box + linear-equality class up to its first linesearch and hand back that search's state
    (x, d, the NR method with the factors of the tangent setup, the constraints, ArmijoWork)."""
    P0 = synth.BallBoxProblem(n, m)
    N, M = n + 1, m + 1
    Jct = ctx.matrix(N, M).hash_fill(1, 0, n, 1.0, n, m)
    xl, xu = (P0.xl, P0.xu) if bounds else (np.full(n, -np.inf), np.full(n, np.inf))
    P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=xl, xu=xu)
    if x0 is None:
        x0 = 0.9 * synth.hash_vector(2, n) + 0.05
    captured = {}
    import lfpsqp_jl_amd.linesearch as LS
    orig = LS.armijo_

    def spy(xnew, x, nn, d, g, f, fval, retract_method, cval, c_, param, work):
        captured.update(x=x, d=d, method=retract_method, c_=c_, m=len(cval), work=work, xnew=xnew, P=P, Jct=Jct)
        raise StopIteration
    LS.armijo_ = spy
    OPT = sys.modules["lfpsqp_jl_amd.optimize"]            # (the package attribute `optimize` is the function, not the module)
    OPT.armijo_ = spy
    try:
        with pytest.raises(StopIteration):
            kw = {} if maxiter_retract is None else dict(maxiter_retract=maxiter_retract)
            P.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3, **kw))
    finally:
        LS.armijo_ = orig
        OPT.armijo_ = orig
    assert isinstance(captured["method"], L.NR)
    return captured


@pytest.mark.parametrize("nb,bounds,mcols", [(2, False, 0), (4, True, 0), (3, True, 0), (4, False, 0),
                                             (4, True, 36), (3, False, 129), (4, True, 129),   # 129 = config 4's width (33 column groups, exact form)
                                             (4, True, 300), (2, False, 300)])                 # the wide form of the one-pass kernel
def test_exact_batch_is_bit_identical(dev_ctx, nb, bounds, mcols):
    _bit_identical(dev_ctx, nb, bounds, mcols)


@pytest.mark.parametrize("which", ["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("nb,bounds,mcols,cap", [(4, True, 0, 1), (3, False, 129, 1), (4, True, 300, 2), (4, True, 129, 1)])
def test_exact_batch_with_several_spans_per_workgroup(request, monkeypatch, which, nb, bounds, mcols, cap):
    """The batched launch keeps fewer workgroups resident than the single-trial step (four trials' accumulators per lane): each then works
    through SEVERAL of the single-trial launch's spans and emits a partial row per span.  Forced here with the test hook LFPSQP_NRB_WG_CAP
    (read when a context is created), whatever the occupancy of the two kernels on the device at hand."""
    lib = request.getfixturevalue("emu_lib" if which == "emu" else "gpu_lib")
    monkeypatch.setenv("LFPSQP_NRB_WG_CAP", str(cap))
    ctx = L.Context(0, lib)
    monkeypatch.delenv("LFPSQP_NRB_WG_CAP")
    try:
        _bit_identical(ctx, nb, bounds, mcols)
    finally:
        ctx.close()


def _bit_identical(ctx, nb, bounds, mcols):
    emu = _is_emu(ctx)
    n, m = (1500, 7) if emu else (200_000, 31)
    if mcols:
        n, m = ((700 if mcols >= 256 else 1100) if emu else 150_000), mcols - 1
    cap = capture_first_linesearch(ctx, n, m, bounds)
    x, d, method, c_, mm = cap["x"], cap["d"], cap["method"], cap["c_"], cap["m"]
    ctx.set_nr_batch_mode(False)
    assert L.retract_nr_batch_width_(c_, method) == 4
    alphas = [64.0, 0.02, 2e-3, 1e-4][:nb]
    xts, xns = cap["work"].batch_vectors(nb)
    for a, xt in zip(alphas, xts):
        L.waxpby(1.0, x, a, d, xt)
    one = cap["xnew"]
    seen_flags, maxiter = set(), 40
    for _ in range(2):
        method.maxiter = maxiter
        cvs = np.zeros((nb, mm))
        got = L.retract_nr_batch_(cvs, xns, c_, xts, x, method)
        assert got is not None
        its = []
        for b in range(nb):
            cv = np.zeros(mm)
            fl, it, _ = L.retract_(cv, one, c_, xts[b], x, method)
            seen_flags.add(fl)
            its.append(it)
            assert (got[b][0], got[b][1]) == (fl, it), (b, got[b], fl, it)
            xa = xns[b].download2() if bounds else xns[b].download()
            xb_ = one.download2() if bounds else one.download()
            # bit for bit -- the iterate a FAILED trial stops at too (NaN patterns and all)
            assert np.array_equal(xa, xb_, equal_nan=True), (b, fl, np.nanmax(np.abs(xa - xb_)))
            assert np.array_equal(cvs[b], cv, equal_nan=True), (b, fl, np.nanmax(np.abs(cvs[b] - cv)))
        if 1 in seen_flags:
            break
        maxiter = max(its) - 1          # second round: the slowest trial now runs into the iteration limit while the others converge
        if maxiter < 1:
            break
    assert 0 in seen_flags, seen_flags
    if nb >= 3:
        assert 1 in seen_flags, seen_flags


def test_armijo_with_exact_batch_is_the_one_by_one_search(dev_ctx):
    """Default DeviceOptions (ls_batch automatic, exact batch) against ls_batch = 1 in config 4's failing regime: the traces are IDENTICAL --
    accepted steps, flags, every Newton-iteration count -- and the iterates agree bit for bit."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (150, 4) if emu else (4000, 16)
    mr = 30 if emu else 100
    P0 = synth.BallBoxProblem(n, m)
    res = {}
    for k in (1, 0):
        Jct = ctx.matrix(n + 1, m + 1).hash_fill(1, 0, n, 1.0, n, m)
        P = L.QuadLinearBallBox(ctx, n, m, Jct, P0.eq.b, R2=P0.R2, xl=P0.xl, xu=P0.xu)
        tr = []
        ctx.options.ls_batch = k
        assert ctx.options.ls_batch_matrix_cores is False
        x, obj, lam, ti = P.optimize(P0.x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3 if emu else 10,
                                                           maxiter_retract=mr), trace=tr)
        res[k] = (tr, x, ti)
    ctx.options.ls_batch = 0
    tr1, x1, ti1 = res[1]
    tr0, x0_, ti0 = res[0]
    assert ti1.iter == ti0.iter and len(tr1) == len(tr0)
    assert any((t.get('retract_iter1') or 0) >= mr for t in tr1)           # the regime with failed retractions was reached
    for a, b in zip(tr1, tr0):
        for key in ('alpha', 'ls_flag', 'retract_iter1', 'retract_iter2', 'steptype', 'tn_iter', 'mtype', 'rank'):
            assert a.get(key) == b.get(key), (key, a.get(key), b.get(key), a['iter'])
        assert np.array_equal(a['x'], b['x']), (a['iter'], np.abs(a['x'] - b['x']).max())
    assert np.array_equal(x1, x0_)
