"""The communicator code paths that can be exercised on a 1-GPU box: a 1-rank RCCL communicator
(dlopen, ncclGetUniqueId, ncclCommInitRank, stream-ordered ncclAllReduce calls from inside the
solvers) and the torch.distributed(nccl) callback transport.  Run in subprocesses so each gets a
clean process group."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
mode = sys.argv[1]
if mode == "torch":
    import torch, torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
if mode == "rccl_after_torch":      # what bench.py does for N > 1: torch (gloo control plane) is loaded first
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1)
import lfpsqp_jl_amd as L
from oracle import lfpsqp_ref as R, synth
from tests.helpers import DiagOpRef
ctx = L.Context(0)
if mode in ("rccl", "rccl_after_torch"):
    ctx.comm_init_rccl(0, 1, ctx.comm_unique_id())
else:
    from lfpsqp_jl_amd.distributed import torch_allreduce_callback
    ctx.comm_init_callback(0, 1, torch_allreduce_callback(0))
# force the multi-rank code path (separate reduce / all-reduce / post kernels) with a 1-rank communicator
import ctypes
n, m = 30000, 12
J = ctx.matrix(n, m).hash_fill(1); Z = ctx.matrix(n, m)
S, Vt, rank = L.ksvd_(J, Z)
a = 4.0 * synth.hash_vector(3, n) + 5.0; bh = synth.hash_vector(4, n)
x, lam = ctx.vector(n), ctx.vector(m)
it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n, a)), L.DeviceBasis(Z), ctx.vector(n, bh), None, tol=1e-10, maxit=300)
x0, l0 = np.zeros(n), np.zeros(m)
i0, nr0 = R.projcg_(x0, l0, DiagOpRef(a), Z.download(), bh, np.zeros(m), tol=1e-10, maxit=300)
assert it == i0, (it, i0)
assert np.linalg.norm(x.download() - x0) <= 1e-10 * np.linalg.norm(x0)
assert abs(L.amax(ctx.vector(n, bh)) - np.abs(bh).max()) == 0
print("OK", mode, it)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rccl", "rccl_after_torch", "torch"])
def test_one_rank_communicator(mode, tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "comm1.py"
    script.write_text(_SCRIPT.format(root=ROOT, port=port))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), mode], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "OK " + mode in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu(tmp_path):
    """The whole N > 1 flow of bench.py (torch.distributed.run launch, row sharding, generator offsets,
    all-reduce placement, max-over-ranks timing, single JSON line) with 2 ranks on ONE GPU through the
    host-staged gloo transport; the 2-rank result must reproduce the 1-rank iterate norm."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    common = ["--steps", "6", "--warmup", "1", "--rows", "600000", "--cols", "24", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    def launch(port):
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host-gloo",
                               "--device", "0", "--watchdog-seconds", "150"] + common, cwd=ROOT, capture_output=True, text=True, timeout=240,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        two = launch(port)
    except subprocess.TimeoutExpired:
        two = None
    if two is None or two.returncode != 0:
        # the rendezvous of the two ranks stalled ONCE in round 2 (never reproduced, FINDINGS.md 7): one more attempt on a fresh port, and
        # the first attempt's output in the report if that fails too
        first = "timeout" if two is None else (two.stdout[-800:], two.stderr[-1500:])
        s2 = socket.socket(); s2.bind(("127.0.0.1", 0)); port2 = s2.getsockname()[1]; s2.close()
        two = launch(port2)
        assert two.returncode == 0, (first, two.stdout[-1500:], two.stderr[-3000:])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints exactly one JSON line
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["steps"] == 6
    assert d2["config"]["rows_per_gpu"] < 600000
    assert abs(d2["check"]["x_norm"] - d1["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]
    assert abs(d2["check"]["nr"] - d1["check"]["nr"]) <= 1e-8 * d1["check"]["nr"]


def _device_count():
    """HIP devices this process may use, without initialising the GPU in the pytest process (torch.cuda.device_count() reads the topology)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["p2p", "rccl"])
def test_bench_two_ranks_on_two_different_gpus(comm):
    """The first box with more than one GPU runs the sharded path for real: bench.py --gpus 2 as the driver launches it (torch.distributed.run,
    one rank per device, LOCAL_RANK -> device), once over the library's own one-shot all-reduce (hipIpc-mapped mailboxes ACROSS devices) and once
    over RCCL with two ranks -- neither has executed with more than one device in any earlier round.  The two-rank solve must be the one-rank
    solve: same iteration count, ||x|| to 1e-10.  On a one-GPU box the test skips and says how many devices it saw."""
    import json
    import socket
    ndev = _device_count()
    if ndev < 2:
        pytest.skip(f"needs two GPUs: this box shows {ndev} HIP device(s) (hipGetDeviceCount); the two-rank paths that share ONE device are "
                    "covered by test_bench_two_ranks_share_one_gpu / test_bench_eight_ranks_without_a_launcher")
    common = ["--steps", "6", "--warmup", "2", "--rows", "4000000", "--cols", "128", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from .test_p2p_transport import _bench                      # (one more attempt with a fresh port if the launch of the ranks fails)
    one = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *common], env)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", comm, "--watchdog-seconds", "300",
                  *common], env, port_flag=9)
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["rows_per_gpu"] * 2 >= 4000000
    assert len(set(d2["config"]["device_uuids"])) == 2, d2["config"]["device_uuids"]      # two DIFFERENT devices
    assert d2["config"]["comm"].startswith(comm), d2["config"]["comm"]                     # the transport asked for by name (it fails loudly otherwise)
    assert d1["check"]["iters"] == d2["check"]["iters"]
    assert abs(d1["check"]["x_norm"] - d2["check"]["x_norm"]) <= 1e-10 * d1["check"]["x_norm"]
    assert abs(d1["check"]["nr"] - d2["check"]["nr"]) <= 1e-8 * d1["check"]["nr"]
    print(f"[two GPUs, {comm}] {d2['value']:.1f} it/s on two devices against {d1['value']:.1f} on one; x_norm equal to "
          f"{abs(d1['check']['x_norm'] - d2['check']['x_norm']) / d1['check']['x_norm']:.1e}")
