"""projcg! with a TRIDIAGONAL Hessian on the one-pass iteration (lfpsqp_projcg_tridiag).

The reference applies its Hessian as a LinearMap (src/optimize.jl:228-230; src/projcg.jl:57,74,116); a tridiagonal one -- a chain or
finite-difference term of the objective -- is not row-local, so lfpsqp_projcg_op pays two passes over the basis per iteration.  The
dedicated entry keeps ONE pass (second product carries A rr, the post-op subtracts (U'A U) t; include/lfpsqp_hip.h).  Checked here: the
operator itself, the reduced operator's effect through counts / iterates / multipliers against the oracle's projcg! with the same A as a
matrix-free map, c != 0, the negative-curvature exit, factored and materialised bases, narrow and wide tiles, couplings of both signs and
a matrix that is not diagonally dominant (negative Gram weights), sizes at multiples of the tile and padding granularities, and the
callback path with the same operator."""
import math

import numpy as np
import pytest

import lfpsqp_jl_amd as L
from oracle import synth


class _TriRef:
    def __init__(self, a, e):
        self.a, self.e = a, e

    def _tri(self, v):
        out = self.a * v
        out[:-1] += self.e * v[1:]
        out[1:] += self.e * v[:-1]
        return out

    def mul_(self, dest, v, al=None, be=None):
        t = self._tri(v)
        dest[:] = t if al is None else al * t + be * dest
        return dest

    def adjoint(self):
        return self


def _operator(ctx, a, e, a0=0.0):
    n = len(a)
    return L.TridiagonalOperator(a0, ctx.vector(n, a - a0), ctx.vector(n, np.concatenate([e, [123.0]])))      # (the last entry must be ignored)


@pytest.mark.parametrize("n", [1, 2, 3, 511, 512, 2048, 2049, 4097])
def test_tridiagonal_product(dev_ctx, n):
    ctx = dev_ctx
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    e = 0.8 * synth.hash_vector(15, max(n - 1, 1))[:n - 1]
    vh = synth.hash_vector(7, n)
    A = _operator(ctx, a, e, a0=0.25)
    out = ctx.vector(n)
    A.mul_(out, ctx.vector(n, vh))
    ref = _TriRef(a, e)._tri(vh.copy())
    assert np.abs(out.download() - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())
    out.upload(np.ones(n))
    A.mul_(out, ctx.vector(n, vh), 2.0, -1.0)                       # mul!(dest, A, v, alpha, beta)
    assert np.abs(out.download() - (2.0 * ref - 1.0)).max() <= 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n,m,factored,dominant", [(1500, 6, False, True), (1700, 130, False, True), (2048, 128, False, False), (4096, 33, True, True),
                                                    (2400, 128, True, False), (800, 300, False, True)])
def test_projcg_with_a_tridiagonal_operator_on_one_pass(dev_ctx, n, m, factored, dominant):
    from oracle import lfpsqp_ref as R
    ctx = dev_ctx
    a = 4.0 * synth.hash_vector(3, n) + 5.0                          # 1 .. 9
    e = (0.8 if dominant else 3.0) * synth.hash_vector(15, n - 1)    # both signs; not dominant: |e_i| + |e_{i-1}| > a_i on many rows
    if not dominant:
        a = a + 4.5                                                  # ... but still positive definite (Gershgorin is not sharp here: checked below)
        lo = np.linalg.eigvalsh(np.diag(a) + np.diag(e, 1) + np.diag(e, -1))[0] if n <= 2500 else 1.0
        assert lo > 0.05 and np.any(a - np.abs(np.concatenate([e, [0]])) - np.abs(np.concatenate([[0], e])) < 0)
    bh = synth.hash_vector(4, n)
    if factored:                                   # the basis kept as U = J W (lfpsqp_basis.Z == NULL)
        Jh = synth.hash_matrix(5, n, m)
        J = ctx.matrix(n, m, np.asfortranarray(Jh))
        W = np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(J, None, W=W)
        U = L.DeviceBasis(None, rank, generator=(J, W))
        Uh = np.asfortranarray(Jh @ W)
    else:
        Uh, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
        Uh = np.asfortranarray(Uh)
        U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = _operator(ctx, a, e)
    Aref = _TriRef(a, e)
    b = ctx.vector(n, bh)
    work = L.ProjCGWork(ctx, n, m)
    for ch, tol in ((None, 1e-10), (np.linspace(-1, 1, m), 1e-12)):
        x0, l0 = np.zeros(n), np.zeros(m)
        i0, nr0 = R.projcg_(x0, l0, Aref, Uh, bh, np.zeros(m) if ch is None else ch, tol=tol)
        x, lam = ctx.vector(n), ctx.vector(m)
        i1, nr1 = L.projcg_(x, lam, A, U, b, None if ch is None else ctx.vector(m, ch), tol=tol, work=work)
        assert i1 == i0 and i1 > 3 and nr1 == pytest.approx(nr0, rel=1e-5)
        assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
        assert np.abs(lam.download() - l0).max() < 1e-10
        if factored:
            continue                               # (the callback path needs a materialised basis)
        A.fused = False                            # the callback path (lfpsqp_projcg_op, two passes per iteration) with the same operator
        x2, lam2 = ctx.vector(n), ctx.vector(m)
        i2, nr2 = L.projcg_(x2, lam2, A, U, b, None if ch is None else ctx.vector(m, ch), tol=tol)
        A.fused = True
        assert i2 == i1
        assert np.linalg.norm(x.download() - x2.download()) <= 1e-10 * np.linalg.norm(x0)
    # an iteration limit (src/projcg.jl:71) leaves the same iterate
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, Aref, Uh, bh, np.zeros(m), tol=1e-30, maxit=5)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, A, U, b, None, tol=1e-30, maxit=5, work=work)
    assert (i1, i0) == (5, 5) and nr1 == pytest.approx(nr0, rel=1e-9)
    assert np.linalg.norm(x.download() - x0) <= 1e-12 * np.linalg.norm(x0)
    # negative curvature (src/projcg.jl:77-82)
    A2 = _operator(ctx, -a, e)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, _TriRef(-a, e), Uh, bh, np.zeros(m), tol=1e-10)
    x, lam = ctx.vector(n), ctx.vector(m)
    i1, nr1 = L.projcg_(x, lam, A2, U, b, None, tol=1e-10, work=work)
    assert (i1, nr1) == (i0, nr0) and math.isinf(nr1)
    assert np.linalg.norm(x.download() - x0) <= 1e-10 and np.all(np.isnan(lam.download()))


def test_tridiagonal_operator_is_refused_where_the_one_pass_form_does_not_exist(dev_ctx):
    """Two columns (no one-pass tile), and a matrix view as the basis: LFPSQP_ERR_UNSUPPORTED from the C entry; projcg_ then takes the callback
    path and still solves the problem."""
    import ctypes as C
    from lfpsqp_jl_amd import _capi
    from oracle import lfpsqp_ref as R
    ctx = dev_ctx
    n, m = 900, 2
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    e = 0.8 * synth.hash_vector(15, n - 1)
    Uh, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
    Uh = np.asfortranarray(Uh)
    U = L.DeviceBasis(ctx.matrix(n, m, Uh))
    A = _operator(ctx, a, e)
    bh = synth.hash_vector(4, n)
    b, x, lam, Av = ctx.vector(n, bh), ctx.vector(n), ctx.vector(m), ctx.vector(n)
    work = L.ProjCGWork(ctx, n, m)
    it, nr = _capi.c_i64(), C.c_double()
    a_c, u_c, w_c = A._c(), U._c(), work._c()
    rc = ctx.L.lfpsqp_projcg_tridiag(ctx.h, x.h, lam.h, C.byref(a_c), Av.h, C.byref(u_c), b.h, None, 1e-10, 100, n, 1, C.byref(w_c), C.byref(it), C.byref(nr))
    assert rc == -5
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, _TriRef(a, e), Uh, bh, np.zeros(m), tol=1e-10)
    i1, nr1 = L.projcg_(x, lam, A, U, b, None, tol=1e-10, work=work)
    assert i1 == i0 and np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)


def test_chain_objective_problem_class_follows_the_oracle(dev_ctx):
    """ChainSeparableLinear: f = sum phi(x_i) + kappa/2 sum (x_{i+1} - x_i)^2 under dense equalities.  `optimize` runs its truncated-Newton
    solves with the tridiagonal Lagrangian Hessian on the one-pass iteration; the trajectory -- counts of every solve included -- is the
    oracle's with the same functions in numpy and hess_lag_vec! as a matrix-free tridiagonal product (src/optimize.jl:228-230)."""
    from oracle import lfpsqp_ref as R
    from .test_capi_retractions import _compare_traces, _is_emu, _sep_host
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m = (260, 4) if emu else (6000, 16)
    maxiter = 4 if emu else 10
    kind, kappa = 1, 1.7
    a = 0.5 + synth.hash_vector(21, n) ** 2
    c = 0.3 * synth.hash_vector(22, n)
    phi, d1, d2 = _sep_host(kind, a, c)
    P0 = synth.BallBoxProblem(n, m)
    x0 = synth.hash_vector(2, n)

    def lap(v):
        out = np.zeros_like(v)
        dv = v[1:] - v[:-1]
        out[:-1] -= dv
        out[1:] += dv
        return kappa * out

    f = lambda x: float(np.sum(phi(x[:n])) + 0.5 * np.dot(x[:n], lap(x[:n])))

    def grad_(g, x):
        g[:n] = d1(x[:n]) + lap(x[:n])

    def hlv_(dest, src, x, lam):
        dest[:] = d2(x) * src + lap(src)

    tr0, tr = [], []
    par = dict(do_project_retract=False, maxiter=maxiter, tn_kappa=1e-6)
    xr, objr, lamr, tir = R.optimize(f, grad_, P0.eq.c_, P0.eq.jac_, hlv_, x0, None, None, m, R.LFPSQPParams(disp=R.DisplayOption.off, **par), trace=tr0)
    P = L.ChainSeparableLinear(ctx, n, m, ctx.matrix(n, m).hash_fill(1), P0.eq.b, kind, a, c, kappa=kappa)
    import sys
    OPT = sys.modules["lfpsqp_jl_amd.optimize"]            # (the package attribute `optimize` is the function, not the module)
    seen, orig = [], OPT.projcg_

    def spy(*args, **kw):
        seen.append((type(args[2]).__name__, bool(kw.get("start_given")), bool(kw.get("start_projected"))))
        return orig(*args, **kw)
    OPT.projcg_ = spy
    try:
        x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(disp=L.DisplayOption.off, **par), trace=tr)
    finally:
        OPT.projcg_ = orig
    # every truncated-Newton solve ran with the tridiagonal operator, started from the tangent step's state, never from its folded projection
    assert seen and all(s == ("TridiagonalOperator", True, False) for s in seen)
    assert ti.iter == tir.iter and ti.condition.name == tir.condition.name
    assert any((t.get('tn_iter') or 0) > 3 for t in tr0)                       # the Newton systems take several iterations
    # DeviceOptions.tridiagonal_one_pass = False: the same operator through the callback path, the same trajectory
    ctx.options.tridiagonal_one_pass = False
    try:
        tr2 = []
        x2, obj2, lam2, ti2 = P.optimize(x0, L.LFPSQPParams(disp=L.DisplayOption.off, **par), trace=tr2)
    finally:
        ctx.options.tridiagonal_one_pass = True
    assert ti2.iter == ti.iter and _compare_traces(tr2, tr0) is None and np.linalg.norm(x2 - x) <= 1e-10 * np.linalg.norm(x)
    assert _compare_traces(tr, tr0) is None
    np.testing.assert_allclose(obj, objr, rtol=1e-11)
    np.testing.assert_allclose(lam, lamr, rtol=1e-6, atol=1e-9)
    assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)


def test_tridiagonal_operator_refuses_row_shards(emu_lib):
    """A context with a communicator holds one SHARD of the rows: the couplings across the shard boundaries are not its to apply.  Both the
    one-pass solver and the operator on its own answer LFPSQP_ERR_UNSUPPORTED (no silently dropped couplings); projcg_ raises."""
    import ctypes as C
    from lfpsqp_jl_amd import _capi
    ctx = L.Context(0, emu_lib)
    try:
        ctx.comm_init_callback(0, 1, lambda ptr, count, op, stream: 0)          # (one rank, but a communicator: the shard case)
        n, m = 700, 8
        a = 4.0 * synth.hash_vector(3, n) + 5.0
        e = 0.8 * synth.hash_vector(15, n - 1)
        Uh, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
        U = L.DeviceBasis(ctx.matrix(n, m, np.asfortranarray(Uh)))
        A = _operator(ctx, a, e)
        b, x, lam, Av = ctx.vector(n, synth.hash_vector(4, n)), ctx.vector(n), ctx.vector(m), ctx.vector(n)
        work = L.ProjCGWork(ctx, n, m)
        it, nr = _capi.c_i64(), C.c_double()
        a_c, u_c, w_c = A._c(), U._c(), work._c()
        rc = ctx.L.lfpsqp_projcg_tridiag(ctx.h, x.h, lam.h, C.byref(a_c), Av.h, C.byref(u_c), b.h, None, 1e-10, 100, n, 1, C.byref(w_c), C.byref(it), C.byref(nr))
        assert rc == -5
        assert ctx.L.lfpsqp_tridiag_mul(ctx.h, C.byref(a_c), b.h, Av.h) == -5
        with pytest.raises(L.LfpsqpError):
            L.projcg_(x, lam, A, U, b, None, tol=1e-10, work=work)
    finally:
        ctx.close()
