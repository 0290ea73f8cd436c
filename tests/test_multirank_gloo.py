"""world_size-2 run of the sharded hot path on CPU: two processes, the emulator build of the C-ABI
library, m-vector / CG-scalar all-reduces through torch.distributed (gloo) via the library's
callback transport.  Checks the row-sharded results against the single-process oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import lfpsqp_ref as R
from oracle import synth

from .helpers import DiagOpRef

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def two_ranks(emu_lib, tmp_path_factory):
    d = tmp_path_factory.mktemp("mp")
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), str(r), "2", str(port), str(d / f"r{r}.npz")],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(d / f"r{r}.npz") for r in range(2)]


def test_shards_partition_the_rows(two_ranks):
    a, b = two_ranks
    assert a["r0"] == 0 and a["r1"] == b["r0"] and b["r1"] == 3200 and a["r1"] % 2048 == 0


def test_sharded_factorize_and_projcg_match_single_process_oracle(two_ranks):
    a, b = two_ranks
    n, m = 3200, 6
    Jh = synth.hash_matrix(1, n, m)
    Z = np.vstack([a["Z"], b["Z"]])
    np.testing.assert_array_equal(a["S"], b["S"])                        # replicated factors agree bit for bit
    np.testing.assert_array_equal(a["lam"], b["lam"])
    np.testing.assert_allclose(a["S"], np.linalg.svd(Jh, compute_uv=False), rtol=1e-12)
    np.testing.assert_allclose(Z.T @ Z, np.eye(m), atol=1e-13)
    np.testing.assert_allclose((Z * a["S"]) @ a["Vt"], Jh, atol=1e-12)
    av = 4.0 * synth.hash_vector(3, n) + 5.0
    bv = synth.hash_vector(4, n)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, DiagOpRef(av), np.asfortranarray(Z), bv, np.zeros(m), tol=1e-10, maxit=500)
    x = np.concatenate([a["x"], b["x"]])
    assert int(a["it"]) == int(b["it"]) == i0
    assert np.linalg.norm(x - x0) <= 1e-10 * np.linalg.norm(x0)
    np.testing.assert_allclose(a["lam"], l0, atol=1e-11)


@pytest.mark.parametrize("tag,dpr", [("nr", False), ("pp", True)])
def test_sharded_config3_driver(two_ranks, tag, dpr):
    a, b = two_ranks
    n, m = 2800, 5
    prob0, x0 = synth.config3(n, m)
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, m,
                                     R.LFPSQPParams(do_project_retract=dpr, disp=R.DisplayOption.off))
    x = np.concatenate([a[f"c3{tag}_x"], b[f"c3{tag}_x"]])
    assert int(a[f"c3{tag}_iter"]) == int(b[f"c3{tag}_iter"]) == tir.iter
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(a[f"c3{tag}_lam"], lamr, rtol=1e-8, atol=1e-12)
    np.testing.assert_array_equal(a[f"c3{tag}_lam"], b[f"c3{tag}_lam"])
    np.testing.assert_allclose(a[f"c3{tag}_obj"], objr, rtol=1e-12)


def test_sharded_config4_slack_on_last_rank(two_ranks):
    a, b = two_ranks
    n, m = 2600, 4
    P0 = synth.BallBoxProblem(n, m)
    x0 = 0.97 * synth.hash_vector(2, n) + 0.03 * 0.5
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, x0, P0.xl, P0.xu, m, 1,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=3),
                                     derivatives=P0.derivatives())
    x = np.concatenate([a["c4_x"], b["c4_x"]])
    assert int(a["c4_iter"]) == int(b["c4_iter"]) == tir.iter
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(a["c4_obj"], objr, rtol=1e-11)
    np.testing.assert_allclose(a["c4_lam"], lamr, rtol=1e-7, atol=1e-10)


def test_sharded_config4_batched_failed_retractions(two_ranks):
    """From x0 = 0.5 the first Armijo searches of config 4 see failed Newton retractions, so the trial steps are retracted in
    batches (lfpsqp_retract_nr_batch): two ranks must agree with each other and with the single-process oracle on
    every count, accepted step and iterate."""
    a, b = two_ranks
    n, m = 2600, 4
    P0 = synth.BallBoxProblem(n, m)
    tr0 = []
    xr, objr, lamr, tir = R.optimize(P0.f, P0.c_, P0.d_, P0.x0, P0.xl, P0.xu, m, 1,
                                     R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=2, maxiter_retract=25),
                                     derivatives=P0.derivatives(), trace=tr0)
    r1 = np.array([t.get("retract_iter1") or 0 for t in tr0])
    al = np.array([t.get("alpha") or 0.0 for t in tr0])
    assert r1.max() >= 25                                        # failed retractions did occur
    for w in (a, b):
        assert int(w["c4b_iter"]) == tir.iter
        np.testing.assert_array_equal(w["c4b_r1"], r1)
        np.testing.assert_array_equal(w["c4b_alpha"], al)
        assert bool(w["c4b_exact_batch_is_one_by_one"]) and int(w["c4b_failed_retractions"]) >= 1      # the sharded exact batch: bit for bit the one-by-one search
    x = np.concatenate([a["c4b_x"], b["c4b_x"]])
    assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
    np.testing.assert_allclose(a["c4b_obj"], objr, rtol=1e-9)


def test_rank_with_zero_rows(two_ranks):
    """n = 1500 < 2048 (the shard granule): rank 1 owns no rows, yet every collective must still be entered."""
    a, b = two_ranks
    n, m = 1500, 4
    assert (int(a["e0"]), int(a["e1"]), int(b["e0"]), int(b["e1"])) == (0, 1500, 1500, 1500)
    Jh = synth.hash_matrix(1, n, m)
    np.testing.assert_allclose(a["e_S"], np.linalg.svd(Jh, compute_uv=False), rtol=1e-12)
    U, _ = np.linalg.qr(Jh)
    av, bv = 4.0 * synth.hash_vector(3, n) + 5.0, synth.hash_vector(4, n)
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, _ = R.projcg_(x0, l0, DiagOpRef(av), np.asfortranarray(U), bv, np.zeros(m), tol=1e-10, maxit=300)
    assert int(a["e_it"]) == int(b["e_it"]) == i0 and len(b["e_x"]) == 0
    assert np.linalg.norm(a["e_x"] - x0) <= 1e-10 * np.linalg.norm(x0)
    np.testing.assert_array_equal(a["e_lam"], b["e_lam"])
    # the sparse Gram matrix with an empty shard on rank 1
    ke = 2
    rows_e = np.repeat(np.arange(n), ke)
    cols_e = (((np.arange(n) * m) // n)[:, None] + np.arange(ke)[None, :]) % m
    vals_e = (np.random.default_rng(4).standard_normal((n, ke)) + 2.0 * (np.arange(ke) == 0)).ravel()
    Ae = np.zeros((n, m)); np.add.at(Ae, (rows_e, cols_e.ravel()), vals_e)
    we = 0.4 * synth.hash_vector(7, n) + 0.6
    np.testing.assert_array_equal(a["e_gram"], b["e_gram"])
    np.testing.assert_allclose(a["e_gram"], Ae.T @ (we[:, None] * Ae), rtol=0, atol=1e-13 * np.abs(Ae.T @ Ae).max())
    np.testing.assert_allclose(a["e_gram_plain"], Ae.T @ Ae, rtol=0, atol=1e-13 * np.abs(Ae.T @ Ae).max())


def test_sharded_ill_conditioned_factorize_and_operator_callback(two_ranks):
    """Round-2 paths on two ranks: the refinement rounds of lfpsqp_factorize on a block of condition 1e8 (rank and singular
    values of dgesvd, replicated bit for bit; the device Jacobi of the small factor runs on every rank), and lfpsqp_projcg_op
    with a (rank-local) tridiagonal operator against the single-process oracle."""
    a, b = two_ranks
    ni, mi = 3000, 6
    rng = np.random.default_rng(31)
    Q1, _ = np.linalg.qr(rng.standard_normal((ni, mi)))
    Q2, _ = np.linalg.qr(rng.standard_normal((mi, mi)))
    Jill = (Q1 * np.logspace(0, -8, mi)) @ Q2.T
    S0 = np.linalg.svd(Jill, compute_uv=False)
    np.testing.assert_array_equal(a["ill_S"], b["ill_S"])
    assert int(a["ill_rank"]) == int(b["ill_rank"]) == int(np.sum(S0 >= 1e-10)) == mi
    np.testing.assert_allclose(a["ill_S"], S0, rtol=1e-6, atol=100 * np.finfo(float).eps)
    Z = np.vstack([a["ill_Z"], b["ill_Z"]])
    i0 = int(b["i0"])
    av = 4.0 * synth.hash_vector(3, ni) + 5.0
    e = 0.8 * synth.hash_vector(15, ni)[:ni - 1].copy()
    e[i0 - 1] = 0.0                                           # the coupling across the shard boundary is dropped on both ranks
    bv = synth.hash_vector(4, ni)

    class Tri:
        def mul_(self, dest, v, al=None, be=None):
            t = av * v
            t[:-1] += e * v[1:]
            t[1:] += e * v[:-1]
            dest[:] = t if al is None else al * t + be * dest
            return dest

        def adjoint(self):
            return self
    x0, l0 = np.zeros(ni), np.zeros(mi)
    it0, _ = R.projcg_(x0, l0, Tri(), np.asfortranarray(Z), bv, np.zeros(mi), tol=1e-10, maxit=400)
    x = np.concatenate([a["op_x"], b["op_x"]])
    assert int(a["op_it"]) == int(b["op_it"]) == it0
    assert np.linalg.norm(x - x0) <= 1e-9 * np.linalg.norm(x0)
    np.testing.assert_array_equal(a["op_lam"], b["op_lam"])


def test_sharded_sparse_equalities_through_the_default_retraction(two_ranks):
    """Sparse constraint gradients sharded by rows (each rank holds the nonzeros of its rows): c!, jac! and ProjPenalty's pcg! on the
    sparse products, their m-vectors all-reduced -- against the single-process oracle on the assembled dense matrix."""
    a, b = two_ranks
    nsp, msp, ksp = 3000, 8, 3
    rows = np.repeat(np.arange(nsp), ksp)
    cols = (((np.arange(nsp) * msp) // nsp)[:, None] + np.arange(ksp)[None, :]) % msp
    vals = (np.random.default_rng(9).standard_normal((nsp, ksp)) + 2.0 * (np.arange(ksp) == 0)).ravel()
    A = np.zeros((nsp, msp))
    np.add.at(A, (rows, cols.ravel()), vals)
    xs = synth.hash_vector(2, nsp)
    np.testing.assert_allclose(a["sp_b"], A.T @ xs, atol=1e-11)
    np.testing.assert_array_equal(a["sp_b"], b["sp_b"])
    prob0 = synth.QuadLinearProblem(np.asfortranarray(A), A.T @ xs)
    x0 = xs + 0.05 * synth.hash_vector(6, nsp)
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, msp,
                                     R.LFPSQPParams(disp=R.DisplayOption.off, maxiter=3))
    x = np.concatenate([a["sp_x"], b["sp_x"]])
    assert int(a["sp_iter"]) == int(b["sp_iter"]) == tir.iter
    assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)
    np.testing.assert_allclose(a["sp_obj"], objr, rtol=1e-10)
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, msp,
                                     R.LFPSQPParams(disp=R.DisplayOption.off, maxiter=3, do_project_retract=False))
    x = np.concatenate([a["spn_x"], b["spn_x"]])
    assert int(a["spn_iter"]) == int(b["spn_iter"]) == tir.iter
    assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr)
    # lfpsqp_spmat_gram on the shards: replicated bit for bit, equal to the Gram matrix of the assembled block
    w = 0.4 * synth.hash_vector(7, nsp) + 0.6
    np.testing.assert_array_equal(a["sp_gram"], b["sp_gram"])
    np.testing.assert_array_equal(a["sp_gram_plain"], b["sp_gram_plain"])
    np.testing.assert_allclose(a["sp_gram"], A.T @ (w[:, None] * A), rtol=0, atol=1e-13 * np.abs(A.T @ A).max())
    np.testing.assert_allclose(a["sp_gram_plain"], A.T @ A, rtol=0, atol=1e-13 * np.abs(A.T @ A).max())
    np.testing.assert_allclose(a["spn_obj"], objr, rtol=1e-10)


def test_sharded_nonlinear_constraint_class(two_ranks):
    """Round 3: the device-resident nonlinear class (lfpsqp_elementwise) row-sharded over two ranks -- dense A with mixed kinds and the
    common quadratic term (whose sum of squares is a collective), and the reference's sin system on the nonzeros of each shard -- through
    `optimize` with the Newton retraction: counts, objective and iterates of the single-process oracle with host callables."""
    from .test_elementwise import ew_callables, ew_test_data
    a, b = two_ranks
    n, m = 2600, 6
    A, kind, qw, bb, target, x0 = ew_test_data(n, m)
    c_, jac_, hdiag = ew_callables(A, kind, qw, bb)
    f = lambda xx: float(np.sum((xx[:n] - target) ** 2))

    def grad_(g, xx):
        g[:n] = 2.0 * (xx[:n] - target)

    def hlv_(dest, src, xx, lam_):
        dest[:n] = (2.0 + hdiag(xx, lam_)) * src[:n]
    tr0 = []
    xr, objr, lamr, tir = R.optimize_core(f, grad_, c_, jac_, hlv_, x0, None, None, m,
                                          R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=5), trace=tr0)
    x = np.concatenate([a["ew_x"], b["ew_x"]])
    assert int(a["ew_iter"]) == int(b["ew_iter"]) == tir.iter
    np.testing.assert_array_equal(a["ew_r1"], np.array([t.get("retract_iter1") or 0 for t in tr0]))
    np.testing.assert_array_equal(a["ew_lam"], b["ew_lam"])
    assert np.linalg.norm(x - xr) <= 1e-10 * max(1.0, np.linalg.norm(xr))
    np.testing.assert_allclose(a["ew_obj"], objr, rtol=1e-11)
    # the sin system (test/test_retractions.jl:34-54), sparse, sharded
    from .test_oracle_reference_properties import sin_system
    ns, ms = 2600, 40
    _, cs_, js_ = sin_system(ns, ms)
    As = np.zeros((ns, ms)); i_ = np.arange(ms); As[2 * i_ + 1, i_], As[2 * i_, i_] = 1.0, -1.0
    ks = np.zeros(ns); ks[0:2 * ms:2] = 1
    _, _, hd_s = ew_callables(As, ks, None, np.zeros(ms))
    targ = 0.5 * synth.hash_vector(21, ns)
    fs = lambda xx: float(np.sum((xx[:ns] - targ) ** 2))

    def gs_(g, xx):
        g[:ns] = 2.0 * (xx[:ns] - targ)

    def hs_(dest, src, xx, lam_):
        dest[:ns] = (2.0 + hd_s(xx, lam_)) * src[:ns]
    xr, objr, lamr, tir = R.optimize_core(fs, gs_, cs_, js_, hs_, np.zeros(ns), None, None, ms,
                                          R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=5))
    xs = np.concatenate([a["sin_x"], b["sin_x"]])
    assert int(a["sin_iter"]) == int(b["sin_iter"]) == tir.iter
    assert np.linalg.norm(xs - xr) <= 1e-10 * max(1.0, np.linalg.norm(xr))
    np.testing.assert_allclose(a["sin_obj"], objr, rtol=1e-11)
