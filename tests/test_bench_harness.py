"""bench.py's harness logic (argument handling, JSON contract, roofline / cpu_baseline
objects) exercised at toy size on the emulator build; the numbers mean nothing here."""
import json

import bench


def test_bench_json_contract(emu_lib, capsys):
    out = bench.main(["--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "4096", "--cols", "8", "--cpu-sample-n", "4096", "--prewarm-seconds", "0.05", "--side-scale", "0.04"], lib=emu_lib)
    line = capsys.readouterr().out.strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert out["value"] > 0
    assert d["check"]["iters"] == 4 and "resumed" in d["config"]["timed_region"] and d["single_call"]["value"] > 0
    assert d["roofline"]["traffic"] is None            # no PMC passes of this build
    ops = d["extras"]["operators"]                   # the one-pass iteration with non-diagonal Hessians, timed beside the headline
    assert "error" not in ops and ops["tridiagonal_one_pass_ms"] > 0 and ops["diagonal_plus_rank4_one_pass_ms"] > 0


def test_bench_two_ranks_as_the_driver_launches_it(emu_lib):
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`: row-sharded workload, one
    all-reduce per iteration (gloo here, RCCL on the GPUs), barrier-bracketed timing, rank 0 prints the one line;
    the solve is the same as the single-rank one."""
    import os
    import socket
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--steps", "4", "--warmup", "1", "--rows", "9000", "--cols", "8", "--no-cpu-baseline", "--prewarm-seconds", "0.2", "--lib", EMU_LIB]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host-gloo", "--device", "0", *common]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["config"]["parallelism"] == "row-sharded x2"
    assert "extras" not in d2 and d2["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extras", *common], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["check"]["iters"] == d2["check"]["iters"] == 5        # 1 warmup + 4 timed iterations of one solve
    assert abs(d1["check"]["x_norm"] - d2["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]


def test_bench_starts_its_own_ranks_when_called_without_a_launcher(emu_lib):
    """`python bench.py --gpus 4` with no torch.distributed.run around it (the form the driver uses for --gpus 1): the parent, which has
    touched no GPU, starts the ranks as one child, relays rank 0's line and the exit code.  Four ranks over the library's one-shot
    peer-to-peer all-reduce (named explicitly, then picked by --comm auto: the emulator build has no RCCL behind it, so that probe is skipped and recorded);
    the solve is the single-rank one."""
    import os
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so")
    common = ["--steps", "4", "--warmup", "1", "--rows", "19000", "--cols", "8", "--no-cpu-baseline", "--prewarm-seconds", "0.1", "--lib", EMU_LIB]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    got = {}
    for comm in ("p2p", "auto"):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--comm", comm, "--device", "0", *common], cwd=ROOT,
                             capture_output=True, text=True, timeout=900, env=env)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 4 and d["config"]["parallelism"] == "row-sharded x4" and d["config"]["rows_per_gpu"] == 6144
        assert "p2p" in d["config"]["comm"] and "fine-grained" in d["config"]["comm"]
        pr = d["config"]["comm_probe"]
        assert pr["chosen"] == "p2p" and pr["p2p"]["ok"] and pr["p2p"]["mailbox_memory"] == "fine-grained" and pr["p2p"]["us_per_allreduce"] > 0
        if comm == "auto":
            assert pr["ranks_share_a_gpu"] and pr["rccl"]["ok"] is False
        got[comm] = d
    assert got["p2p"]["check"] == got["auto"]["check"]                       # fixed rank order: bit for bit
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extras", *common], cwd=ROOT,
                         capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["check"]["iters"] == got["p2p"]["check"]["iters"] == 5
    assert abs(d1["check"]["x_norm"] - got["p2p"]["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]
    # a launcher whose WORLD_SIZE contradicts --gpus is an error, not a silent 1-rank run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *common], cwd=ROOT, capture_output=True, text=True,
                         timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr
