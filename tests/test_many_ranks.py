"""More than two ranks (reduction points src/projcg.jl:75,84,96,98,103): FOUR and EIGHT emulator processes, row-sharded, over
the library's one-shot peer-to-peer all-reduce (mailboxes in POSIX shared memory) and over the gloo callback transport.  The shards are
UNEVEN (n = 5 tiles + 700 rows: at 4 ranks 2/2/1/1 tiles with a ragged last one, at 8 ranks six ranks with one tile and two with NO rows --
among them the last rank, which still owns config 4's slack variable), every collective must line up on every rank, the replicated results
must agree bit for bit across the ranks of a run, and the iterates must agree with the single-process oracle to 1e-10."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import lfpsqp_ref as R
from oracle import synth

from .helpers import DiagOpRef

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M = 5 * 2048 + 700, 5


def _run(d, world, transport):
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mr_worker.py"), str(r), str(world), str(d), transport, str(N), str(M)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=1500)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(d, f"out_{transport}_{r}.npz")) for r in range(world)]


@pytest.fixture(scope="module")
def oracle_runs():
    out = {}
    Jh = synth.hash_matrix(1, N, M)
    out["S"] = np.linalg.svd(Jh, compute_uv=False)
    prob0, x0 = synth.config3(N, M)
    for tag, dpr in (("nr", False), ("pp", True)):
        out["c3" + tag] = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, None, None, M,
                                     R.LFPSQPParams(do_project_retract=dpr, disp=R.DisplayOption.off))
    P0 = synth.BallBoxProblem(N, M)
    x04 = 0.97 * synth.hash_vector(2, N) + 0.03 * 0.5
    tr = []
    out["c4"] = R.optimize(P0.f, P0.c_, P0.d_, x04, P0.xl, P0.xu, M, 1, R.LFPSQPParams(do_project_retract=False, disp=R.DisplayOption.off, maxiter=3),
                           derivatives=P0.derivatives(), trace=tr)
    out["c4_trace"] = tr
    return out


@pytest.fixture(scope="module", params=[4, 8])
def runs(request, emu_lib, tmp_path_factory):
    world = request.param
    d = tmp_path_factory.mktemp(f"mr{world}")
    # (the gloo callback transport differs from the peer-to-peer one in the all-reduce alone: it runs at four ranks, the library's own at four and eight)
    return world, _run(d, world, "p2p"), (_run(d, world, "gloo") if world == 4 else None)


def test_uneven_and_empty_shards(runs):
    world, p2p, _ = runs
    sizes = [int(w["r1"]) - int(w["r0"]) for w in p2p]
    assert sum(sizes) == N and int(p2p[0]["r0"]) == 0 and all(int(p2p[r]["r1"]) == int(p2p[r + 1]["r0"]) for r in range(world - 1))
    if world == 4:
        assert sizes == [4096, 4096, 2048, 700]
    else:
        assert sizes == [2048] * 5 + [700, 0, 0]                   # two ranks own no rows, the last of them config 4's slack variable


def test_replicated_results_agree_bit_for_bit_across_ranks(runs):
    world, p2p, gloo = runs
    for run in (r_ for r_ in (p2p, gloo) if r_ is not None):
        for key in ("sum9000", "dot", "amax", "S", "Vt", "it", "nr", "lam", "c3nr_lam", "c3nr_obj", "c3pp_lam", "c3pp_obj", "c4_lam", "c4_obj",
                    "c4_r1", "c4_alpha"):
            for r in range(1, world):
                np.testing.assert_array_equal(run[0][key], run[r][key], err_msg=f"{key}, rank {r}")
    assert len({int(w["p2p_collectives"]) for w in p2p}) == 1         # every rank issued the same number of collectives


def test_p2p_sums_in_rank_order(runs):
    """The one-shot all-reduce adds the ranks' payloads in the order 0, 1, 2, ...: reproduce the 9000-double test sum exactly."""
    world, p2p, gloo = runs
    acc = synth.hash_vector(7, 9000)
    for r in range(1, world):
        acc = acc + synth.hash_vector(7 + r, 9000)
    np.testing.assert_array_equal(p2p[0]["sum9000"], acc)
    if gloo is not None:
        np.testing.assert_allclose(gloo[0]["sum9000"], acc, rtol=0, atol=1e-14)  # (gloo's ring associates differently)


def test_sharded_tangent_setup_and_projcg_match_the_oracle(runs, oracle_runs):
    world, p2p, gloo = runs
    av = 4.0 * synth.hash_vector(3, N) + 5.0
    bv = synth.hash_vector(4, N)
    for run in (r_ for r_ in (p2p, gloo) if r_ is not None):
        Z = np.vstack([w["Z"] for w in run])
        np.testing.assert_allclose(run[0]["S"], oracle_runs["S"], rtol=1e-12)
        np.testing.assert_allclose(Z.T @ Z, np.eye(M), atol=1e-13)
        x0, l0 = np.zeros(N), np.zeros(M)
        i0, _ = R.projcg_(x0, l0, DiagOpRef(av), np.asfortranarray(Z), bv, np.zeros(M), tol=1e-10, maxit=400)
        x = np.concatenate([w["x"] for w in run])
        assert int(run[0]["it"]) == i0
        assert np.linalg.norm(x - x0) <= 1e-10 * np.linalg.norm(x0)
        np.testing.assert_allclose(run[0]["lam"], l0, atol=1e-11)


@pytest.mark.parametrize("tag", ["nr", "pp"])
def test_sharded_config3_matches_the_oracle(runs, oracle_runs, tag):
    world, p2p, gloo = runs
    xr, objr, lamr, tir = oracle_runs["c3" + tag]
    for run in (r_ for r_ in (p2p, gloo) if r_ is not None):
        x = np.concatenate([w[f"c3{tag}_x"] for w in run])
        assert int(run[0][f"c3{tag}_iter"]) == tir.iter
        assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
        np.testing.assert_allclose(run[0][f"c3{tag}_obj"], objr, rtol=1e-12)
        np.testing.assert_allclose(run[0][f"c3{tag}_lam"], lamr, rtol=1e-8, atol=1e-12)


def test_sharded_config4_matches_the_oracle(runs, oracle_runs):
    world, p2p, gloo = runs
    xr, objr, lamr, tir = oracle_runs["c4"]
    tr = oracle_runs["c4_trace"]
    for run in (r_ for r_ in (p2p, gloo) if r_ is not None):
        x = np.concatenate([w["c4_x"] for w in run])
        assert x.size == N                                           # (returned truncated, src/optimize.jl:68; the slack variable rode on the last rank)
        assert int(run[0]["c4_iter"]) == tir.iter
        np.testing.assert_array_equal(run[0]["c4_r1"], np.array([t.get("retract_iter1") or 0 for t in tr]))
        np.testing.assert_array_equal(run[0]["c4_alpha"], np.array([t.get("alpha") or 0.0 for t in tr]))
        assert np.linalg.norm(x - xr) <= 1e-10 * np.linalg.norm(xr)
        np.testing.assert_allclose(run[0]["c4_obj"], objr, rtol=1e-11)
        np.testing.assert_allclose(run[0]["c4_lam"], lamr, rtol=1e-7, atol=1e-10)
