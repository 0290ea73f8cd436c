"""The one-shot peer-to-peer all-reduce (lfpsqp_comm_init_p2p, SURVEY §5 "distributed communication backend"; reduction points
src/projcg.jl:75,84,96,98,103): two processes whose mailboxes are mapped into each other -- on the CPU two emulator processes over POSIX
shared memory, on the GPU two processes sharing the one device through hipIpc -- run plain collectives, the sharded tangent setup and the
sharded projected CG, and must agree BIT FOR BIT with the same run over the gloo (host-staged) transport and across ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so")


def _bench(cmd, env, timeout=900, port_flag=None):
    """One bench.py launch as a subprocess; ONE more attempt if it fails (the rendezvous of freshly started ranks stalled once in round 2 and a
    launch failed once in round 6, neither reproducible: under `pytest -x` a transient must not end the suite).  The first attempt's output is
    part of the report if the second fails too.  `port_flag`: index of the --master-port value in cmd (a fresh port for the second attempt)."""
    import socket
    first = None
    for attempt in range(2):
        if attempt and port_flag is not None:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                cmd = list(cmd)
                cmd[port_flag] = str(s.getsockname()[1])
        try:
            res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
        except subprocess.TimeoutExpired as e:
            res = None
            why = ("timeout", str(e)[-500:])
        if res is not None and res.returncode == 0:
            if first is not None:
                print(f"[bench launch] first attempt failed, second passed; first: {first}")
            return res
        if res is not None:
            why = (res.returncode, res.stdout[-800:], res.stderr[-2000:])
        if first is None:
            first = why
    raise AssertionError(("bench.py failed twice", first, why))


def _run(d, transport, lib, n, m, timeout):
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), "2", str(d), transport, lib, str(n), str(m)],
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(d, f"out_{transport}_{r}.npz")) for r in range(2)]


def _check(p2p, gloo, n, m):
    for key in ("sum9000", "dot", "amax", "S", "it", "nr", "lam"):
        np.testing.assert_array_equal(p2p[0][key], p2p[1][key], err_msg=key)          # replicated results agree across the ranks
        np.testing.assert_array_equal(p2p[0][key], gloo[0][key], err_msg=key)         # ... and with the other transport, bit for bit
    for r in range(2):
        np.testing.assert_array_equal(p2p[r]["x"], gloo[r]["x"])
    x = np.concatenate([p2p[0]["x"], p2p[1]["x"]])
    assert x.size == n and int(p2p[0]["it"]) > 3 and float(p2p[0]["nr"]) < 1e-10
    print(f"[p2p] {float(p2p[0]['us_per_allreduce']):.1f} us per 261-double all-reduce (gloo / host-staged: {float(gloo[0]['us_per_allreduce']):.1f}); "
          f"sharded projcg {int(p2p[0]['it'])} iterations in {float(p2p[0]['projcg_s']) * 1e3:.1f} ms ({float(gloo[0]['projcg_s']) * 1e3:.1f})")


def test_two_emulator_processes_over_shared_memory(emu_lib, tmp_path):
    n, m = 6000, 12
    p2p = _run(tmp_path, "p2p", EMU, n, m, 600)
    gloo = _run(tmp_path, "gloo", EMU, n, m, 600)
    _check(p2p, gloo, n, m)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(2_000_000, 128)])
def test_two_processes_sharing_the_gpu(gpu_lib, tmp_path, n, m):
    p2p = _run(tmp_path, "p2p", "default", n, m, 300)
    gloo = _run(tmp_path, "gloo", "default", n, m, 300)
    _check(p2p, gloo, n, m)


@pytest.mark.gpu
def test_bench_two_ranks_over_p2p_at_the_headline_shape():
    """bench.py --gpus 2 --comm p2p exactly as the driver launches it (torch.distributed.run, two ranks -- here sharing the one GPU), n = 1e7,
    m = 128 row-sharded: the same solve as one rank (equal count, ||x|| to 1e-10), bit-identical to the host-staged transport."""
    import json
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--prewarm-seconds", "0.3"]
    out = {}
    for comm in ("p2p", "host-gloo"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        two = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", comm, "--device", "0", *common], env, port_flag=9)
        out[comm] = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][0])
    one = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common], env)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    p, g = out["p2p"], out["host-gloo"]
    assert p["n_gpus"] == 2 and "p2p" in p["config"]["comm"]
    assert p["check"]["iters"] == g["check"]["iters"] == d1["check"]["iters"] == 13
    assert p["check"]["x_norm"] == g["check"]["x_norm"] and p["check"]["nr"] == g["check"]["nr"]          # fixed rank order: bit for bit
    assert abs(d1["check"]["x_norm"] - p["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]
    print(f"[bench --gpus 2 on ONE GPU] p2p {p['value']:.1f} it/s, host-staged gloo {g['value']:.1f} it/s, one rank {d1['value']:.1f} it/s")


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2 --comm p2p --device 0` -- NO torch.distributed.run around it: bench.py starts its ranks itself (a child
    process, before anything touches the GPU) and prints the one line; --comm auto picks the same transport on a box whose ranks share the
    GPU (RCCL is probed too, refuses the duplicate device, and the record says so).  The mailbox must be fine-grained memory."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--prewarm-seconds", "0.3", "--rows", "2e6"]
    out = {}
    for comm in ("p2p", "auto"):
        two = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", comm, "--device", "0", *common], env)
        lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        out[comm] = json.loads(lines[0])
        d = out[comm]
        assert d["n_gpus"] == 2 and d["check"]["iters"] == 13 and "p2p" in d["config"]["comm"]
        pr = d["config"]["comm_probe"]
        assert pr["chosen"] == "p2p" and pr["p2p"]["mailbox_memory"] == "fine-grained", pr
        print(f"[bench --gpus 2, no launcher, --comm {comm}] {d['value']:.1f} it/s; {d['config']['comm']}; probe {pr}")
    assert out["p2p"]["check"] == out["auto"]["check"]


@pytest.mark.gpu
def test_bench_eight_ranks_without_a_launcher():
    """`python3 bench.py --gpus 8 --comm p2p --device 0 --rows 2e6 --cols 128` -- the command the driver runs on an 8-GPU node, here with the
    eight ranks sharing the one GPU (reduction points src/projcg.jl:75,84,96,98,103, one all-reduce of 2m + 5 doubles per iteration over seven
    peers' mailboxes): the solve is the one-rank solve (equal count, ||x|| and nr to 1e-10).  Through --comm auto, bring_up_comm probes BOTH
    library transports: RCCL refuses the duplicate device on every rank (ncclCommInitRank: invalid usage), the record says so, and the run
    carries on over the peer-to-peer transport -- bit-identical to the run that named it."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--rows", "2e6", "--cols", "128", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--prewarm-seconds", "0.2"]
    out = {}
    for comm in ("p2p", "auto"):
        res = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--comm", comm, "--device", "0", *common], env)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = out[comm] = json.loads(lines[0])
        pr = d["config"]["comm_probe"]
        print(f"[bench --gpus 8 on ONE GPU, no launcher, --comm {comm}] {d['value']:.1f} it/s; {d['config']['comm']}; comm_probe {pr}")
        assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "row-sharded x8" and d["check"]["iters"] == 8
        assert pr["chosen"] == "p2p" and pr["ranks_share_a_gpu"] and pr["p2p"]["ok"] and pr["p2p"]["mailbox_memory"] == "fine-grained"
        if comm == "auto":
            assert pr["rccl"]["ok"] is False and "ncclCommInitRank" in pr["rccl"]["refused"], pr      # probed, refused, said so
    assert out["p2p"]["check"] == out["auto"]["check"]                       # fixed rank order: bit for bit
    one = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common], env)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    c1, c8 = d1["check"], out["p2p"]["check"]
    assert c1["iters"] == c8["iters"] == 8
    assert abs(c1["x_norm"] - c8["x_norm"]) <= 1e-10 * c1["x_norm"] and abs(c1["nr"] - c8["nr"]) <= 1e-10 * c1["nr"]

