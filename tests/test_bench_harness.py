"""bench.py's harness logic (argument handling, JSON contract, roofline / cpu_baseline
objects) exercised at toy size on the emulator build; the numbers mean nothing here."""
import json

import bench


def test_bench_json_contract(emu_lib, capsys):
    out = bench.main(["--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "4096", "--cols", "8", "--cpu-sample-n", "4096", "--prewarm-seconds", "0.05", "--work-candidates", "2", "--basis-candidates", "2"], lib=emu_lib)
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
    common = ["--steps", "4", "--warmup", "1", "--rows", "9000", "--cols", "8", "--no-cpu-baseline", "--prewarm-seconds", "0.2", "--work-candidates", "1", "--basis-candidates", "2", "--lib", EMU_LIB]
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
