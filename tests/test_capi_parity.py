"""Parity of the C-ABI library against the oracle.

Every case runs twice: on the CPU HIP emulator build of the SAME sources
(tests/emu, `-m "not gpu"`: host logic, indexing, reduction order, exit semantics) and
on the real MI355X library (`-m gpu`, the parity tests proper).  Tolerances: fp64
reductions in a different order than OpenBLAS => a few ulp on each primitive, 1e-10
relative on iterates (BASELINE.json north_star), iteration counts equal."""
import math

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R
from oracle import port, synth

from .helpers import DiagOpRef


def test_primitives(dev_ctx):
    ctx = dev_ctx
    rng = np.random.default_rng(0)
    for n, m in ((1, 1), (1023, 3), (1024, 4), (1025, 5), (5000, 37), (2049, 130)):
        Mh = np.asfortranarray(rng.standard_normal((n, m)))
        vh = rng.standard_normal(n)
        th = rng.standard_normal(m)
        M, v, t = ctx.matrix(n, m, Mh), ctx.vector(n, vh), ctx.vector(m, th)
        out = ctx.vector(m)
        L.gemv_t(M, v, out)
        np.testing.assert_allclose(out.download(), Mh.T @ vh, rtol=0, atol=1e-13 * np.sqrt(n) * 10)
        np.testing.assert_array_equal(M.download(), Mh)
        y = ctx.vector(n, vh)
        L.gemv_n(M, t, y, 2.0, 3.0)
        np.testing.assert_allclose(y.download(), 2 * (Mh @ th) + 3 * vh, atol=1e-13 * m)
        L.gemv_n(M, t, y, 1.0, 0.0)
        np.testing.assert_allclose(y.download(), Mh @ th, atol=1e-13 * m)
        # leading-`rank` columns only (kgemv!, src/la_helper.jl:36-44)
        r = max(m - 2, 0)
        out.fill(7.0)
        L.gemv_t(M, v, out, ncols=r)
        got = out.download()
        np.testing.assert_allclose(got[:r], Mh[:, :r].T @ vh, atol=1e-12 * np.sqrt(n))
        assert np.all(got[r:] == 7.0)
        assert L.dot(v, v) == pytest.approx(vh @ vh, rel=1e-14)
        assert L.nrm2(v) == pytest.approx(np.linalg.norm(vh), rel=1e-14)
        assert L.amax(v) == np.abs(vh).max()
        w = ctx.vector(n, rng.standard_normal(n))
        wh = w.download()
        L.axpby(0.5, v, -2.0, w)
        np.testing.assert_allclose(w.download(), 0.5 * vh - 2 * wh, atol=1e-15 * 8)
        L.vmul(v, w, w)
        np.testing.assert_allclose(w.download(), vh * (0.5 * vh - 2 * wh), rtol=1e-15, atol=1e-300)
        for obj in (M, v, t, out, y, w):
            obj.free()


def test_empty_and_ragged(dev_ctx):
    ctx = dev_ctx
    v = ctx.vector(0)
    assert L.dot(v, v) == 0.0 and L.amax(v) == 0.0
    M = ctx.matrix(5, 0)
    y = ctx.vector(5, np.arange(5.0))
    L.gemv_n(M, ctx.vector(1), y, 1.0, 2.0, ncols=0)      # n x 0 operator: y = beta*y
    np.testing.assert_array_equal(y.download(), 2 * np.arange(5.0))


def test_hash_generators_bit_exact(dev_ctx):
    ctx = dev_ctx
    n, m = 3001, 7
    np.testing.assert_array_equal(ctx.matrix(n, m).hash_fill(1).download(), synth.hash_matrix(1, n, m))
    # sharded generation: rows 1024.. of a taller matrix
    np.testing.assert_array_equal(ctx.matrix(n - 2048, m).hash_fill(1, 2048, n).download(), synth.hash_matrix(1, n, m)[2048:])
    np.testing.assert_array_equal(ctx.vector(n).hash_fill(3, 5, 4.0, 5.0).download(), 4.0 * synth.hash_vector(3, n, 5) + 5.0)
    np.testing.assert_array_equal(port.hash_matrix(1, n, m), synth.hash_matrix(1, n, m))


def _cg_problem(n, m, seed=0):
    Uh, _ = np.linalg.qr(synth.hash_matrix(1 + seed, n, m)) if m > 0 else (np.zeros((n, 0)), None)
    a = 4.0 * synth.hash_vector(3 + seed, n) + 5.0       # A = diag(a), a in [1, 9)
    b = synth.hash_vector(4 + seed, n)
    return np.asfortranarray(Uh), a, b


@pytest.mark.parametrize("n,m", [(1000, 10), (3000, 1), (2500, 33), (1024, 0)])
def test_projcg_matches_oracle(dev_ctx, n, m):
    """test_cg.jl:21-30 on the device + trajectory parity with the oracle."""
    ctx = dev_ctx
    Uh, a, bh = _cg_problem(n, m)
    rng = np.random.default_rng(3)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = L.DiagOperator(0.0, ctx.vector(n, a))
    b = ctx.vector(n, bh)
    work = L.ProjCGWork(ctx, n, m)
    for ch in (None, rng.standard_normal(m)):
        for tol in (1e-6, 1e-10, 1e-14):
            x0, l0 = np.zeros(n), np.zeros(m)
            i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, np.zeros(m) if ch is None else ch, tol=tol)
            x, lam = ctx.vector(n), ctx.vector(max(m, 1))
            c = None if ch is None else ctx.vector(max(m, 1), ch if m else np.zeros(1))
            i1, nr1 = L.projcg_(x, lam, A, U, b, c, tol=tol, work=work)
            xd, ld = x.download(), lam.download()[:m]
            assert i1 == i0
            assert nr1 < tol and nr1 == pytest.approx(nr0, rel=1e-6)
            assert np.linalg.norm(xd - x0) <= 1e-10 * np.linalg.norm(x0)
            if m:
                assert np.abs(ld - l0).max() < 1e-11
                cc = np.zeros(m) if ch is None else ch
                assert np.linalg.norm(Uh.T @ xd - cc) < 1e-13          # test_cg.jl:27
                K = np.concatenate([a * xd + Uh @ ld - bh, Uh.T @ xd - cc])
                assert np.linalg.norm(K) < max(tol, 1e-12)               # test_cg.jl:28


@pytest.mark.parametrize("n,m", [(2100, 4), (2500, 33), (2100, 128), (4100, 129), (1300, 300), (1100, 513)])
def test_projcg_fused_iteration_matches_two_pass_kernels(dev_ctx, monkeypatch, n, m):
    """The default projcg iteration makes ONE pass over U (U'rp assembled from U'g and U'(A g), csrc/projcg.hip);
    LFPSQP_ONEPASS=-1 selects the two-pass kernels.  Same counts, iterates equal to rounding, both equal to the oracle;
    covers exact / shifted last column groups, the widest narrow tile and the wide form (m > 256: the four waves of a
    workgroup split the columns)."""
    Uh, a, bh = _cg_problem(n, m)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, np.zeros(m), tol=1e-12)
    res = {}
    for mode in ("-1", "0"):
        monkeypatch.setenv("LFPSQP_ONEPASS", mode)
        ctx = L.Context(0, dev_ctx.L)
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, a)), L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None,
                           tol=1e-12)
        res[mode] = (it, nr, x.download(), lam.download()[:m])
        ctx.close()
    for mode in ("-1", "0"):
        it, nr, xd, ld = res[mode]
        assert it == i0 and nr == pytest.approx(nr0, rel=1e-5)
        assert np.linalg.norm(xd - x0) <= 1e-11 * np.linalg.norm(x0)
        assert np.abs(ld - l0).max() < 1e-11
        assert np.linalg.norm(Uh.T @ xd) < 1e-13
    assert not np.array_equal(res["-1"][2], res["0"][2])          # really two different code paths


def test_projcg_fused_iteration_long_run_on_an_ill_conditioned_operator(dev_ctx, monkeypatch):
    """A few hundred iterations with kappa(A) = 1e3: the pieces the fused iteration assembles (U'rp from U'g, U'(A g) and the
    t3 recurrence; d'Ad from three sums) must not drift -- the solution stays in the null space of U' to rounding and meets
    the KKT system as well as the two-pass kernels' does.  (Iterates of the two paths differ here like any two summation
    orders do: CG on such an operator amplifies rounding.)"""
    n, m = 1100, 8
    Uh, _, bh = _cg_problem(n, m)
    a = np.exp(np.log(1e3) * (synth.hash_vector(7, n) + 1.0) / 2.0)            # spectrum spread over [1, 1e3]
    res = {}
    for mode in ("-1", "0"):
        monkeypatch.setenv("LFPSQP_ONEPASS", mode)
        ctx = L.Context(0, dev_ctx.L)
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, a)), L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None,
                           tol=1e-6, maxit=3000)
        xd, ld = x.download(), lam.download()[:m]
        res[mode] = (it, nr, np.linalg.norm(Uh.T @ xd), np.linalg.norm(a * xd + Uh @ ld - bh))
        ctx.close()
    for mode in ("-1", "0"):
        it, nr, feas, kkt = res[mode]
        assert 60 < it < n and nr < 1e-6           # (the loop bound is min(maxit, n + m), src/projcg.jl:71)
        assert feas < 1e-11 and kkt < 1e-4
    assert abs(res["0"][0] - res["-1"][0]) <= max(3, res["-1"][0] // 50)


def test_projcg_negative_curvature(dev_ctx):
    """test_cg.jl:39-55: indefinite A => (i, Inf), lambda = NaN, x a unit direction with x'Ax <= 0."""
    ctx = dev_ctx
    n, m = 2000, 10
    Uh, a, bh = _cg_problem(n, m)
    a = a.copy()
    a[::3] *= -1.0
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, np.zeros(m), tol=1e-20)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, a)), L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None,
                        tol=1e-20)
    xd = x.download()
    assert math.isinf(nr1) and math.isinf(nr0) and i1 == i0
    assert np.all(np.isnan(lam.download()))
    assert np.linalg.norm(Uh.T @ xd) < 1e-13
    assert xd @ (a * xd) <= 0.0
    assert np.linalg.norm(xd - x0) < 1e-10


def test_projcg_iteration_limit_and_identity_operator(dev_ctx):
    """maxit semantics (src/projcg.jl:71) and the a0*I operator (hess_lag_vec! = 2v of configs 2-5)."""
    ctx = dev_ctx
    n, m = 1500, 6
    Uh, a, bh = _cg_problem(n, m)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    b = ctx.vector(n, bh)
    x, lam = ctx.vector(n), ctx.vector(m)
    for maxit in (0, 1, 3):
        x0, l0 = np.zeros(n), np.zeros(m)
        i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, np.zeros(m), tol=1e-300, maxit=maxit)
        i1, nr1 = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, a)), U, b, None, tol=1e-300, maxit=maxit)
        assert i1 == i0 == maxit
        assert (math.isinf(nr1) and math.isinf(nr0)) or nr1 == pytest.approx(nr0, rel=1e-9)
        assert np.linalg.norm(x.download() - x0) <= 1e-12 * max(np.linalg.norm(x0), 1.0)
    # A = 2I converges in one iteration (SURVEY "hard parts")
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(2.0 * np.ones(n)), Uh, bh, np.zeros(m), tol=1e-8)
    i1, nr1 = L.projcg_(x, lam, L.DiagOperator(2.0), U, b, None, tol=1e-8)
    assert i1 == i0 == 1
    assert np.linalg.norm(x.download() - x0) <= 1e-13 * np.linalg.norm(x0)


@pytest.mark.parametrize("n,m,stack", [(2100, 16, False), (700, 300, False)])
def test_projcg_resume_continues_the_same_solve(dev_ctx, n, m, stack):
    """LFPSQP_PROJCG_RESUME (bench.py's timed region): W iterations, then K more, give bit for bit the iterate, count and
    residual of one call with the limit W + K -- also when the resumed part converges -- and the flag is refused when there
    is nothing to resume."""
    ctx = dev_ctx
    Uh, a, bh = _cg_problem(n, m)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = L.DiagOperator(0.0, ctx.vector(n, a))
    b = ctx.vector(n, bh)
    work, work2 = L.ProjCGWork(ctx, n, m), L.ProjCGWork(ctx, n, m)
    x1, x2 = ctx.vector(n), ctx.vector(n)
    for W, K, tol in ((3, 4, 1e-300), (1, 1, 1e-300), (2, 500, 1e-9)):
        i_ref, nr_ref = L.projcg_(x1, None, A, U, b, None, tol=tol, maxit=W + K, work=work, want_lambda=False)
        iw, _ = L.projcg_(x2, None, A, U, b, None, tol=tol, maxit=W, work=work2, want_lambda=False)
        assert iw == W
        i2, nr2 = L.projcg_(x2, None, A, U, b, None, tol=tol, maxit=K, work=work2, want_lambda=False, resume=True)
        assert (i2, nr2) == (i_ref, nr_ref)
        assert np.array_equal(x1.download(), x2.download())
    with pytest.raises(L.LfpsqpError):        # the converged solve above left nothing to resume
        L.projcg_(x2, None, A, U, b, None, tol=1e-9, maxit=2, work=work2, want_lambda=False, resume=True)
    L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=2, work=work2, want_lambda=False)
    ctx.sync(); x2.download()                           # calls that queue no device work keep the state ...
    L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=1, work=work2, want_lambda=False, resume=True)
    with pytest.raises(L.LfpsqpError):        # ... other work vectors do not match it
        L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=1, work=work, want_lambda=False, resume=True)
    # ANY library call that queued kernels in between voids it (it may have rewritten the CG scalars or the vectors: round-2 advisor finding),
    # and so does another operator or right-hand side
    L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=2, work=work2, want_lambda=False)
    L.nrm2(x2)
    with pytest.raises(L.LfpsqpError):
        L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=1, work=work2, want_lambda=False, resume=True)
    L.projcg_(x2, None, A, U, b, None, tol=1e-300, maxit=2, work=work2, want_lambda=False)
    with pytest.raises(L.LfpsqpError):
        L.projcg_(x2, None, L.DiagOperator(1.0, A.dg), U, b, None, tol=1e-300, maxit=1, work=work2, want_lambda=False, resume=True)


@pytest.mark.parametrize("n,m", [(2100, 16), (800, 260)])
def test_projcg_residual_buffer_schemes_agree(dev_ctx, n, m):
    """lfpsqp_ctx_set_residual_buffers: the alternating scheme moves where the residual lives between iterations, never
    the arithmetic -- iterates, counts and residual norms are bit for bit those of the in-place scheme, with and without
    a resumed second call; switching the scheme drops the state a resume would continue from."""
    ctx = dev_ctx
    Uh, a, bh = _cg_problem(n, m)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = L.DiagOperator(0.0, ctx.vector(n, a))
    b = ctx.vector(n, bh)
    work = L.ProjCGWork(ctx, n, m)
    out = {}
    try:
        for mode in (0, 1):
            ctx.set_residual_buffers(mode)
            x, lam = ctx.vector(n), ctx.vector(m)
            res = [L.projcg_(x, lam, A, U, b, None, tol=1e-300, maxit=7, work=work)]
            res.append(L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=3, work=work, want_lambda=False))
            res.append(L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=4, work=work, want_lambda=False, resume=True))
            res.append(L.projcg_(x, lam, A, U, b, None, tol=1e-9, maxit=500, work=work))
            out[mode] = (res, x.download(), lam.download())
        assert out[0][0] == out[1][0]
        assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
        x = ctx.vector(n)
        L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=2, work=work, want_lambda=False)
        ctx.set_residual_buffers(0)
        with pytest.raises(L.LfpsqpError):
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=1, work=work, want_lambda=False, resume=True)
        with pytest.raises(L.LfpsqpError):
            ctx.set_residual_buffers(2)
    finally:
        ctx.set_residual_buffers(0)


def test_projcg_generic_operator_path(dev_ctx):
    """Duck-typed A (the LinearMap case, src/optimize.jl:228-230) goes through the unfused loop."""
    ctx = dev_ctx
    n, m = 1200, 5
    Uh, a, bh = _cg_problem(n, m)

    class UserOp:
        def __init__(self, dg):
            self.inner = L.DiagOperator(0.0, dg)

        def mul_(self, dest, v, al=None, be=None):
            return self.inner.mul_(dest, v, al, be)

        def adjoint(self):
            return self

    x0, l0 = np.zeros(n), np.zeros(m)
    ch = np.linspace(-1, 1, m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, ch, tol=1e-10)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, UserOp(ctx.vector(n, a)), L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), ctx.vector(m, ch),
                        tol=1e-10)
    assert i1 == i0
    assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
    assert np.abs(lam.download() - l0).max() < 1e-11


class _TriDevice:
    """A = tridiag(e, a, e) on device vectors out of the library's own asynchronous primitives (vmul, ranged copies, axpby):
    what a host's LinearMap looks like when its body runs on the device.  Works on plain n-vectors, and on the x-half of a
    stacked vector with a diagonal (here: 2) on the y-half -- the shape of augmented_hess_lag_vec! (inequality_helper.jl:144-158)."""

    def __init__(self, ctx, a, e, stacked_N=None):
        n = len(a)
        self.n, self.N = n, stacked_N
        self.a = ctx.vector(n, a)
        self.e_up = ctx.vector(n, np.concatenate([e, [0.0]]))          # e_up[i] = e[i]   multiplies v[i+1]
        self.e_dn = ctx.vector(n, np.concatenate([[0.0], e]))          # e_dn[i] = e[i-1] multiplies v[i-1]
        self.v, self.sh, self.t, self.out = (ctx.vector(n) for _ in range(4))
        self.calls = 0

    def _tri(self, dest, v):
        n = self.n
        L.vmul(self.a, v, dest)
        self.sh.fill(0.0)
        self.sh.copy_range_from(v, n - 1, 0, 1)                          # sh[i] = v[i+1]
        L.vmul(self.e_up, self.sh, self.t)
        L.axpby(1.0, self.t, 1.0, dest)
        self.sh.fill(0.0)
        self.sh.copy_range_from(v, n - 1, 1, 0)                          # sh[i] = v[i-1]
        L.vmul(self.e_dn, self.sh, self.t)
        L.axpby(1.0, self.t, 1.0, dest)

    def mul_(self, dest, v, al=None, be=None):
        assert al is None
        self.calls += 1
        if self.N is None:
            self._tri(dest, v)
        else:
            hs = dest.hs
            self.v.copy_range_from(v, self.n, 0, 0)
            self._tri(self.out, self.v)
            L.waxpby(2.0, v, 0.0, v, dest)                               # y-half: 2 * src_y (x-half overwritten next)
            dest.copy_range_from(self.out, self.n, 0, 0)
        return dest

    def adjoint(self):
        return self


class _TriRef:
    def __init__(self, a, e, stacked=False):
        self.a, self.e, self.stacked = a, e, stacked

    def _tri(self, v):
        out = self.a * v
        out[:-1] += self.e * v[1:]
        out[1:] += self.e * v[:-1]
        return out

    def mul_(self, dest, v, al=None, be=None):
        n = len(self.a)
        t = np.concatenate([self._tri(v[:n]), 2.0 * v[n:]]) if self.stacked else self._tri(v)
        dest[:] = t if al is None else al * t + be * dest
        return dest

    def adjoint(self):
        return self


@pytest.mark.parametrize("n,m", [(1500, 6), (2500, 130)])
def test_projcg_op_with_a_tridiagonal_operator(dev_ctx, n, m):
    """lfpsqp_projcg_op: the reference's A is a LinearMap closure (src/optimize.jl:228-230; applied at src/projcg.jl:57,74,116).
    A tridiagonal A whose body consists of queued device primitives runs on the C loop -- counts, iterate and multipliers of
    the oracle, the callback invoked once per iteration plus start and lambda."""
    ctx = dev_ctx
    Uh, a, bh = _cg_problem(n, m)
    e = 0.8 * synth.hash_vector(15, n - 1)
    A = _TriDevice(ctx, a, e)
    for ch, tol in ((None, 1e-10), (np.linspace(-1, 1, m), 1e-12)):
        x0, l0 = np.zeros(n), np.zeros(m)
        i0, nr0 = R.projcg_(x0, l0, _TriRef(a, e), Uh, bh, np.zeros(m) if ch is None else ch, tol=tol)
        x, lam = ctx.vector(n), ctx.vector(m)
        A.calls = 0
        i1, nr1 = L.projcg_(x, lam, A, L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None if ch is None else ctx.vector(m, ch), tol=tol)
        assert i1 == i0 and nr1 == pytest.approx(nr0, rel=1e-5)
        assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
        assert np.abs(lam.download() - l0).max() < 1e-10
        assert i1 + 2 <= A.calls <= i1 + 5                     # (the host runs up to two iterations ahead of the exit it sees)
        # the same operator as a lfpsqp_tridiag_op: ONE pass over the basis per iteration (lfpsqp_projcg_tridiag; tests/test_projcg_tridiag.py)
        if m >= 4:
            T = L.TridiagonalOperator(0.0, ctx.vector(n, a), ctx.vector(n, np.concatenate([e, [0.0]])))
            x3, lam3 = ctx.vector(n), ctx.vector(m)
            i3, nr3 = L.projcg_(x3, lam3, T, L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None if ch is None else ctx.vector(m, ch), tol=tol)
            assert i3 == i0 and nr3 == pytest.approx(nr0, rel=1e-5)
            assert np.linalg.norm(x3.download() - x0) <= 1e-10 * np.linalg.norm(x0)
            assert np.abs(lam3.download() - l0).max() < 1e-10
    # negative curvature (src/projcg.jl:77-82) through the same loop
    A2 = _TriDevice(ctx, -a, e)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, _TriRef(-a, e), Uh, bh, np.zeros(m), tol=1e-10)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, A2, L.DeviceBasis(ctx.matrix(n, m, Uh)), ctx.vector(n, bh), None, tol=1e-10)
    assert (i1, nr1) == (i0, nr0) and math.isinf(nr1)
    assert np.linalg.norm(x.download() - x0) <= 1e-10 and np.all(np.isnan(lam.download()))


def test_projcg_op_with_bounds_stacked_basis(dev_ctx):
    """The same with the bound-projected basis Q (InequalityDecompProject, src/inequality_helper.jl:161-212) and an operator of
    augmented_hess_lag_vec!'s shape ([H src_x + ...; diag src_y], :144-158): stacked vectors on the device, 2N-vectors in
    the oracle."""
    from lfpsqp_jl_amd.inequality import InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_, inequality_gradient_
    ctx = dev_ctx
    n, m = 900, 5
    Jh = synth.hash_matrix(1, n, m)
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    xh = 0.6 * synth.hash_vector(2, n)
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    e = 0.8 * synth.hash_vector(15, n - 1)
    bh = synth.hash_vector(4, 2 * n)
    # oracle side
    id0 = R.InequalityData(xl, xu)
    xa0 = np.zeros(2 * n); xa0[:n] = xh
    R.generate_initial_y_(xa0, id0)
    dec0 = R.InequalityDecomp(np.empty((2 * n, m), order='F'), np.empty(m), np.empty((m, m), order='F'), np.empty(n), np.empty(n), np.empty(n),
                              np.asfortranarray(Jh), m)
    R.inequality_gradient_(dec0, xa0, id0)
    PJ = np.vstack([(1 - dec0.Dx ** 2)[:, None] * Jh, (-dec0.Dy * dec0.Dx)[:, None] * Jh])
    U0, S0, Vt0 = np.linalg.svd(PJ, full_matrices=False)
    dec0.U[:, :] = U0
    dec0.rank = m
    Q0 = R.InequalityDecompProject(dec0)
    x0, l0 = np.zeros(2 * n), np.zeros(n + m)
    i0, nr0 = R.projcg_(x0, l0, _TriRef(a, e, stacked=True), Q0, bh, np.zeros(n + m), tol=1e-10)
    # device side
    idata = InequalityData(ctx, xl, xu)
    xa = StackedVector(ctx, n)
    xa.upload(xh, 0)
    generate_initial_y_(xa, idata)
    Jct = ctx.matrix(n, m, Jh)
    dec = InequalityDecomp(ctx, n, m, Jct)
    inequality_gradient_(dec, xa, idata)
    S, Vt, rank = L.ksvd_(Jct, dec.Z, w2=dec.sx)
    dec.rank = rank
    Q = InequalityDecompProject(dec)
    b = StackedVector(ctx, n).upload2(bh)
    x = StackedVector(ctx, n)
    it, nr = L.projcg_(x, None, _TriDevice(ctx, a, e, stacked_N=n), Q, b, None, tol=1e-10, want_lambda=False)
    assert rank == m and it == i0 and nr == pytest.approx(nr0, rel=1e-5)
    assert np.linalg.norm(x.download2() - x0) <= 1e-10 * np.linalg.norm(x0)


def test_c_port_agrees_with_numpy_oracle():
    n, m = 4000, 9
    Uh, a, bh = _cg_problem(n, m)
    x, lam, it, nr = port.projcg(a, Uh, bh, None, 1e-10, 500)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Uh, bh, np.zeros(m), tol=1e-10, maxit=500)
    assert it == i0 and nr == pytest.approx(nr0, rel=1e-8)
    assert np.linalg.norm(x - x0) <= 1e-12 * np.linalg.norm(x0)


# ----------------------------------------------------------------------------- tangent setup
@pytest.mark.parametrize("n,m", [(700, 5), (2049, 16), (1500, 130), (900, 260), (600, 300)])   # 1, 1 (+2 border), 2 (+4 border), 3 panels
def test_gram_and_rmul(dev_ctx, n, m):
    ctx = dev_ctx
    rng = np.random.default_rng(11)
    Mh = np.asfortranarray(rng.standard_normal((n, m)))
    wh = rng.random(n) + 0.1
    M, w = ctx.matrix(n, m, Mh), ctx.vector(n, wh)
    np.testing.assert_allclose(L.gram(M), Mh.T @ Mh, atol=1e-12 * n)
    np.testing.assert_allclose(L.gram(M, w2=w), Mh.T @ (wh[:, None] * Mh), atol=1e-12 * n)
    np.testing.assert_allclose(L.gram(M, ncols=m - 1), (Mh.T @ Mh)[:m - 1, :m - 1], atol=1e-12 * n)
    W = rng.standard_normal((m, max(m - 2, 1)))
    O = ctx.matrix(n, m)
    L.rmul(M, W, O)
    np.testing.assert_allclose(O.download()[:, :W.shape[1]], Mh @ W, atol=1e-12 * m)


@pytest.mark.parametrize("n,m", [(1000, 10), (3000, 40), (1030, 129)])
def test_factorize_replaces_ksvd(dev_ctx, n, m):
    """ksvd! (dgesvd) parity through the quantities the reference actually uses: singular values,
    the projector U U', U S Vt = Jct, and lambda = V S^-1 U'd (src/optimize.jl:335-342)."""
    ctx = dev_ctx
    Jh = synth.hash_matrix(1, n, m)
    U0 = np.empty((n, m), order='F'); S0 = np.empty(m); Vt0 = np.empty((m, m), order='F')
    R.ksvd_(Jh.copy(order='F'), U0, S0, Vt0)
    J, Z = ctx.matrix(n, m, Jh), ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(J, Z)
    Zh = Z.download()
    assert rank == m
    np.testing.assert_array_equal(J.download(), Jh)                 # input is not destroyed
    np.testing.assert_allclose(S, S0, rtol=1e-12)
    np.testing.assert_allclose(Zh.T @ Zh, np.eye(m), atol=5e-14)
    np.testing.assert_allclose((Zh * S) @ Vt, Jh, atol=1e-12)
    d = synth.hash_vector(9, n)
    np.testing.assert_allclose(Zh @ (Zh.T @ d), U0 @ (U0.T @ d), atol=1e-12)
    lam0 = Vt0.T @ ((U0.T @ d) / S0)
    lam1 = Vt.T @ ((Zh.T @ d) / S)
    np.testing.assert_allclose(lam1, lam0, rtol=1e-9, atol=1e-12)


def test_factorize_weighted_is_the_bound_projected_factor(dev_ctx):
    """With bounds the reference factors PJct = [(1-Dx^2).*Jct; -Dy.*Dx.*Jct] (src/optimize.jl:288-291);
    on the device that is the weighted factorisation with w2 = Dy^2 and U = [Dy^2.*Z; -Dx.*Dy.*Z]."""
    ctx = dev_ctx
    n, m = 1200, 7
    rng = np.random.default_rng(5)
    Jh = synth.hash_matrix(1, n, m)
    th = rng.uniform(0, 2 * np.pi, n)
    Dx, Dy = np.cos(th), np.sin(th)
    PJ = np.vstack([(1 - Dx * Dx)[:, None] * Jh, (-Dy * Dx)[:, None] * Jh])
    U0, S0, Vt0 = np.linalg.svd(PJ, full_matrices=False)
    J, Z = ctx.matrix(n, m, Jh), ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(J, Z, w2=ctx.vector(n, Dy * Dy))
    Zh = Z.download()
    Ufull = np.vstack([(Dy * Dy)[:, None] * Zh, (-Dx * Dy)[:, None] * Zh])
    assert rank == m
    np.testing.assert_allclose(S, S0, rtol=1e-12)
    np.testing.assert_allclose(Ufull.T @ Ufull, np.eye(m), atol=5e-14)
    np.testing.assert_allclose(Ufull @ Ufull.T @ np.ones(2 * n), U0 @ (U0.T @ np.ones(2 * n)), atol=1e-11)
    np.testing.assert_allclose((Ufull * S) @ Vt, PJ, atol=1e-12)


def test_factorize_rank_deficient(dev_ctx):
    """rank < m (src/optimize.jl:297-302): duplicated constraints give zero singular values."""
    ctx = dev_ctx
    n, m = 1500, 8
    Jh = synth.hash_matrix(1, n, m)
    Jh[:, 5] = Jh[:, 1] + Jh[:, 2]
    Jh[:, 7] = 2 * Jh[:, 0]
    J, Z = ctx.matrix(n, m, Jh), ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(J, Z)
    Zh = Z.download()
    assert rank == 6 and np.all(S[6:] < 1e-5)
    np.testing.assert_allclose(Zh[:, :6].T @ Zh[:, :6], np.eye(6), atol=5e-14)
    assert np.all(Zh[:, 6:] == 0.0)
    np.testing.assert_allclose((Zh * S) @ Vt, Jh, atol=1e-11)


def test_max_reductions_propagate_nan(dev_ctx):
    """norm(v, Inf) in Julia is NaN if any entry is NaN; the retraction loops rely on that
    ("NaN < tol" is false => keep iterating / report failure, src/retractions.jl:135)."""
    ctx = dev_ctx
    for n, pos in ((5, 3), (3000, 1777), (3000, 2999)):
        h = np.ones(n)
        h[pos] = np.nan
        assert math.isnan(L.amax(ctx.vector(n, h)))
    assert L.amax(ctx.vector(10, -np.arange(10.0))) == 9.0


def test_tuning_variants_agree(dev_ctx):
    """lfpsqp_ctx_set_tuning: (ks, nt) variants of the streaming kernels give the same results
    (bit-identical across nt; rounding-level across ks, whose summation order differs)."""
    ctx = dev_ctx
    n, m = 5000, 9
    Uh, a, bh = _cg_problem(n, m)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = L.DiagOperator(0.0, ctx.vector(n, a))
    b = ctx.vector(n, bh)
    res = {}
    for ks in (2, 4):
        for nt in (False, True):
            ctx.set_tuning(ks, nt)
            x, lam = ctx.vector(n), ctx.vector(m)
            it, nr = L.projcg_(x, lam, A, U, b, None, tol=1e-10)
            res[(ks, nt)] = (it, x.download(), lam.download())
    ctx.set_tuning(0, True)
    for ks in (2, 4):
        assert res[(ks, False)][0] == res[(ks, True)][0]
        np.testing.assert_array_equal(res[(ks, False)][1], res[(ks, True)][1])
    assert res[(2, True)][0] == res[(4, True)][0]
    np.testing.assert_allclose(res[(2, True)][1], res[(4, True)][1], rtol=0, atol=1e-13)
    with pytest.raises(L.LfpsqpError):
        ctx.set_tuning(3, True)


@pytest.mark.parametrize("rows,cols,want_v", [(40, 12, True), (64, 64, True), (130, 130, True), (130, 130, False), (300, 200, True), (520, 500, True), (512, 512, False)])
def test_small_svd_one_sided_jacobi(dev_ctx, monkeypatch, rows, cols, want_v):
    """lfpsqp_small_svd: the replicated small step of the tangent setup (host Jacobi below 64 columns, the device block
    Jacobi of csrc/jacobi.hip from there on, with and without accumulated right vectors) against LAPACK on a graded matrix:
    singular values to HIGH RELATIVE accuracy (the matrix is well-conditioned up to column scaling -- the property the
    refinement rounds of lfpsqp_factorize rely on), orthonormal factors, A = U S V'."""
    ctx = dev_ctx
    if _is_emu_ctx(ctx) and cols > 300:
        pytest.skip("kept small on the emulator")
    monkeypatch.setenv("LFPSQP_EMU_DEVICE_JACOBI", "1")          # (emulator build: the device kernels for more than 256 rows run on request only)
    rng = np.random.default_rng(rows + cols)
    Q, _ = np.linalg.qr(rng.standard_normal((rows, cols)))
    Wm, _ = np.linalg.qr(rng.standard_normal((cols, cols)))
    D = np.logspace(0, -12, cols)
    A = (Q @ (np.eye(cols) + 0.3 * Wm)) * D                       # moderately conditioned times a strong column scaling
    U, S, V = L.small_svd_(ctx, A, want_v)
    import mpmath
    S0 = np.linalg.svd(A / D, compute_uv=False)                    # (reference values through the scaled problem below)
    assert np.all(np.diff(S) <= 0)
    np.testing.assert_allclose(U.T @ U, np.eye(cols), atol=5e-13)
    if want_v:
        np.testing.assert_allclose(V.T @ V, np.eye(cols), atol=1e-13)
        np.testing.assert_allclose((U * S) @ V.T, A, atol=1e-14 * np.abs(A).max())
        # relative accuracy of the SMALL singular values: each sigma_j with its vectors satisfies A v_j = sigma_j u_j column by column
        R_ = A @ V - U * S
        assert np.all(np.linalg.norm(R_, axis=0) <= 1e-12 * S + 1e-300)
    # singular values against extended precision (LAPACK's dgesvd only promises eps * sigma_1 absolute)
    mpmath.mp.dps = 40
    Sx = sorted([float(x) for x in mpmath.svd_r(mpmath.matrix(A.tolist()), compute_uv=False)], reverse=True) if cols <= 64 else None
    if Sx is not None:
        np.testing.assert_allclose(S, Sx, rtol=1e-11)


def _is_emu_ctx(ctx):
    return "emulator" in ctx.device_name


def _with_singular_values(n, sv, seed=21):
    rng = np.random.default_rng(seed)
    m = len(sv)
    Q1, _ = np.linalg.qr(rng.standard_normal((n, m)))
    Q2, _ = np.linalg.qr(rng.standard_normal((m, m)))
    return np.asfortranarray((Q1 * np.asarray(sv)) @ Q2.T), Q1


@pytest.mark.parametrize("case", ["cond1e5", "cond1e7", "cond1e9", "cluster", "across_eps_rank", "big_sigma1", "duplicates", "zero_column", "m200"])
def test_factorize_honours_the_reference_rank_rule(dev_ctx, case):
    """The reference factors with backward-stable dgesvd and counts sigma_j >= eps_rank = 1e-10 ABSOLUTELY
    (src/la_helper.jl:8-34, src/optimize.jl:297-302) -- a Jacobian of condition 1e9 whose singular values all exceed 1e-10 is
    full rank there (Newton retraction, full tangent projection).  lfpsqp_factorize must make the same decision as the
    oracle's dgesvd and return what dgesvd returns to ITS accuracy: singular values to eps*sigma_1 absolute, the rank, a basis
    of range(A V_r) orthonormal to the rounding floor of the problem (eps * sigma_1 / sigma_j), the projector, A = Z S Vt."""
    ctx = dev_ctx
    if case == "m200" and _is_emu_ctx(ctx):
        pytest.skip("kept small on the emulator (the 32-column block kernel is covered by test_small_svd_one_sided_jacobi)")
    n = 3000
    dup = zero = False
    if case == "cond1e5": sv = np.logspace(0, -5, 12)
    elif case == "cond1e7": sv = np.logspace(0, -7, 12)
    elif case == "cond1e9": sv = np.logspace(0, -9, 12)
    elif case == "cluster": sv = np.array([1.0, 1.0, 0.5, 1e-8, 1e-9, 1.0001e-9, 3e-10, 2e-10])
    elif case == "across_eps_rank": sv = np.array([2.0, 1.0, 1e-3, 1e-6, 1e-9, 1.5e-10, 0.5e-10, 1e-12])
    elif case == "big_sigma1": sv = np.array([1e3, 10.0, 1.0, 1e-3, 1e-6, 1e-9])
    elif case == "duplicates": sv, dup = np.logspace(0, -3, 12), True
    elif case == "zero_column": sv, zero = np.logspace(0, -2, 9), True
    else: sv = np.logspace(0, -8, 200)
    Jh, _ = _with_singular_values(n, sv)
    if dup:
        Jh[:, -1] = Jh[:, 0]                     # exactly rank deficient
        Jh[:, -2] = Jh[:, 1]
    if zero:
        Jh[:, 3] = 0.0
    m = Jh.shape[1]
    U0 = np.empty((n, m), order='F'); S0 = np.empty(m); Vt0 = np.empty((m, m), order='F')
    R.ksvd_(Jh.copy(order='F'), U0, S0, Vt0)
    rank0 = int(np.sum(S0 >= 1e-10))
    J, Z = ctx.matrix(n, m, Jh), ctx.matrix(n, m)
    W = np.zeros((m, m), order='F')
    S, Vt, rank = L.ksvd_(J, Z, W=W)
    Zh = Z.download()
    r = rank
    assert rank == rank0, (rank, rank0, S, S0)                           # THE decision optimize takes (src/optimize.jl:396-412)
    np.testing.assert_allclose(S, S0, rtol=0, atol=(100 + 4 * m) * np.finfo(float).eps * S0[0])   # dgesvd's own accuracy
    np.testing.assert_allclose(S[:r], S0[:r], rtol=1e-6)                                      # (eps * cond relative at worst)
    assert np.all(np.diff(S[:r]) <= 0) and np.all(Zh[:, r:] == 0.0) and np.all(Vt[r:, :] == 0.0)
    # orthonormal to the floor the products A*w_j allow: eps * sigma_1 / min(sigma_i, sigma_j)
    floor = 200 * np.sqrt(m) * np.finfo(float).eps * S0[0] / np.minimum.outer(S[:r], S[:r])
    assert np.all(np.abs(Zh[:, :r].T @ Zh[:, :r] - np.eye(r)) <= np.maximum(5e-13, floor))
    np.testing.assert_allclose(Vt[:r] @ Vt[:r].T, np.eye(r), atol=1e-13)
    gen_err = np.abs(Zh[:, :r] - Jh @ W[:, :r]).max(axis=0)                                    # generator: Z = Jct W, to the
    assert np.all(gen_err <= 100 * np.sqrt(m) * np.finfo(float).eps * S0[0] / S[:r])            # rounding of that product
    np.testing.assert_allclose((Zh[:, :r] * S[:r]) @ Vt[:r], (U0[:, :r] * S0[:r]) @ Vt0[:r], atol=1e-12 * S0[0])
    # projector onto the kept subspace vs dgesvd's; both carry the eps*cond uncertainty of the smallest kept direction
    d = synth.hash_vector(9, n)
    unc = 1e3 * np.finfo(float).eps * S0[0] / S0[r - 1]
    assert np.linalg.norm(Zh[:, :r] @ (Zh[:, :r].T @ d) - U0[:, :r] @ (U0[:, :r].T @ d)) <= max(1e-12, unc) * np.linalg.norm(d)
    np.testing.assert_array_equal(J.download(), Jh)


def test_factorize_weighted_ill_conditioned(dev_ctx):
    """The same with row weights (bounds present: w2 = Dy.^2, the reference factors the 2N x M matrix PJct)."""
    ctx = dev_ctx
    n, m = 2500, 10
    rng = np.random.default_rng(8)
    Jh, _ = _with_singular_values(n, np.logspace(0, -8, m), seed=3)
    th = rng.uniform(0, 2 * np.pi, n)
    Dx, Dy = np.cos(th), np.sin(th)
    PJ = np.vstack([(1 - Dx * Dx)[:, None] * Jh, (-Dy * Dx)[:, None] * Jh])
    S0 = np.linalg.svd(PJ, compute_uv=False)
    J, Z = ctx.matrix(n, m, Jh), ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(J, Z, w2=ctx.vector(n, Dy * Dy))
    Zh = Z.download()
    assert rank == int(np.sum(S0 >= 1e-10)) == m
    np.testing.assert_allclose(S, S0, rtol=1e-6, atol=50 * np.finfo(float).eps * S0[0])
    Ufull = np.vstack([(Dy * Dy)[:, None] * Zh, (-Dx * Dy)[:, None] * Zh])
    np.testing.assert_allclose((Ufull * S) @ Vt, PJ, atol=1e-12)
    assert np.abs(Ufull.T @ Ufull - np.eye(m)).max() < 1e-6


@pytest.mark.parametrize("n,m", [(700, 300), (1500, 513)])
def test_wide_matrices_cross_the_column_chunk(dev_ctx, n, m):
    """m larger than the 256-column LDS chunk of the GEMV kernels and than one 128-column Gram panel
    (BASELINE configs[4] has m = 512)."""
    ctx = dev_ctx
    if "emulator" in ctx.device_name and m > 400:
        pytest.skip("kept small on the emulator")
    rng = np.random.default_rng(4)
    Mh = np.asfortranarray(rng.standard_normal((n, m)))
    vh, th = rng.standard_normal(n), rng.standard_normal(m)
    M, v, t, out, y = ctx.matrix(n, m, Mh), ctx.vector(n, vh), ctx.vector(m, th), ctx.vector(m), ctx.vector(n, vh)
    L.gemv_t(M, v, out)
    np.testing.assert_allclose(out.download(), Mh.T @ vh, atol=1e-11)
    L.gemv_n(M, t, y, -1.0, 1.0)
    np.testing.assert_allclose(y.download(), vh - Mh @ th, atol=1e-11)
    np.testing.assert_allclose(L.gram(M), Mh.T @ Mh, atol=1e-10)
    Z = ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(M, Z)
    Zh = Z.download()
    assert rank == m
    np.testing.assert_allclose(S, np.linalg.svd(Mh, compute_uv=False), rtol=1e-10)
    np.testing.assert_allclose(Zh.T @ Zh, np.eye(m), atol=1e-12)


def test_argument_errors_are_status_codes_not_crashes(dev_ctx):
    """Error conventions of the boundary: invalid arguments come back as negative status codes with a
    message (raised as LfpsqpError by the host layer), never as crashes; numerical outcomes are never errors."""
    ctx = dev_ctx
    a, b = ctx.vector(10), ctx.vector(11)
    with pytest.raises(L.LfpsqpError, match="invalid argument"):
        L.dot(a, b)                                             # length mismatch
    M = ctx.matrix(10, 3)
    with pytest.raises(L.LfpsqpError):
        L.gemv_t(M, b, ctx.vector(3))                           # v has the wrong length
    with pytest.raises(L.LfpsqpError):
        L.gemv_t(M, a, ctx.vector(3), ncols=4)                  # more columns than the matrix has
    with pytest.raises(L.LfpsqpError):
        a.upload(np.zeros(11))                                  # upload past the end
    with pytest.raises(L.LfpsqpError):
        L.ksvd_(M, M)                                           # factorize needs Z != Jct
    with pytest.raises(ValueError, match="different lengths"):
        L.InequalityData(ctx, np.zeros(3), np.zeros(4))         # src/inequality_helper.jl:42-44
    # a NaN right-hand side is a numerical outcome, not an error: projcg returns, like the reference would
    Uh = np.asfortranarray(np.linalg.qr(synth.hash_matrix(1, 600, 3))[0])
    bh = synth.hash_vector(4, 600)
    bh[5] = np.nan
    x, lam = ctx.vector(600), ctx.vector(3)
    it, nr = L.projcg_(x, lam, L.DiagOperator(2.0), L.DeviceBasis(ctx.matrix(600, 3, Uh)), ctx.vector(600, bh), None, tol=1e-8, maxit=5)
    assert it == 5 and math.isnan(nr)


def test_a_failed_allocation_does_not_poison_the_next_launch(dev_ctx):
    """lfpsqp_vec_alloc / lfpsqp_mat_alloc that run out of device memory return LFPSQP_ERR_HIP -- and clear HIP's sticky last error, so a caller
    that carries on (a linesearch that batches as many trial points as fit, lfpsqp.jl_amd/linesearch.py::_grow_batch) does not meet the stale
    out-of-memory error at its next launch check."""
    ctx = dev_ctx
    with pytest.raises(L.LfpsqpError):
        ctx.vector(2 ** 46)                               # 512 TiB
    with pytest.raises(L.LfpsqpError):
        ctx.matrix(2 ** 40, 64)
    xh = synth.hash_vector(3, 5000)
    x = ctx.vector(5000, xh)
    assert L.nrm2(x) == pytest.approx(np.linalg.norm(xh), rel=1e-14)      # a launch with a check behind it
    y = ctx.vector(5000)
    L.waxpby(2.0, x, 0.0, x, y)
    np.testing.assert_array_equal(y.download(), 2.0 * xh)
