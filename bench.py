#!/usr/bin/env python3
"""Benchmark of the hot path: projected-CG iterations/s at n=1e7, m=128, fp64
(BASELINE.json metric), with the achieved HBM GB/s of the J / J' matvec kernels.

    python bench.py --gpus N --steps K --warmup W

A "step" is one projected-CG iteration (src/projcg.jl:71-112 of the reference) on the
synthetic sustained-iteration workload of SURVEY §8(d): U = n x m basis, A = diag(1 + 9 (u + 1) / 2) = diag(5.5 + 4.5 u)
(seed 3), b = u (seed 4), c = 0, tol = 1e-300 so exactly K iterations run inside one
`lfpsqp_projcg` call; everything is resident in HBM before the timed region.  For N > 1
the SAME global problem is row-sharded over the N GPUs (one process per GPU, m-vector and
CG scalars all-reduced over RCCL) => "scaling": "strong".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


SIDE_SCALE = 1.0          # --side-scale (tests): lengths of the side measurements

def main(argv=None, lib=None):
    """`lib` is injected only by tests/test_bench_harness.py (emulator build, tiny sizes)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", dest="n", type=float, default=1e7, help="global n (default: the metric's 1e7)")
    ap.add_argument("--cols", dest="m", type=int, default=128, help="m (default 128)")
    ap.add_argument("--basis", choices=["orthonormal", "scaled-hash"], default="orthonormal")
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="untimed device warm-up before the W warmup steps: the same iteration loop for this long, so that a GPU that "
                         "was idle (low-power state) is timed in its steady state; 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the tangent-setup / Newton-retraction timings (not part of `value`)")
    ap.add_argument("--cpu-sample-n", type=float, default=2e6)
    ap.add_argument("--side-scale", type=float, default=1.0,
                    help="(tests) scale of the SIDE measurements' lengths -- iterations of the Newton / pcg! timings in `extras`, CPU seconds and the host "
                         "triad's size in `cpu_baseline`; never touches the timed region of `value`")
    ap.add_argument("--comm", choices=["auto", "rccl", "p2p", "torch", "host-gloo"], default="auto",
                    help="all-reduce transport for N > 1.  auto (default): bring up the library's one-shot peer-to-peer all-reduce (mailboxes mapped "
                         "through hipIpc: one exchange instead of a ring) AND the library-native RCCL communicator, check each with a known sum, time "
                         "200 all-reduces of the iteration's payload on each, keep the faster one that works (all ranks agree through the gloo control "
                         "plane; both latencies are reported in config.comm_probe); rccl / p2p: that transport or fail loudly -- no fallback; torch: "
                         "torch.distributed nccl callback; host-gloo: functional test only (host-staged; ranks may share one GPU)")
    ap.add_argument("--p2p-allow-coarse", action="store_true",
                    help="allow ordinary (coarse-grained) device memory for the p2p mailboxes when fine-grained memory cannot be exported "
                         "(lfpsqp_comm_p2p_allow_coarse); the kind that was allocated is reported in config.comm either way")
    ap.add_argument("--device", type=int, default=None, help="HIP device index (default: LOCAL_RANK)")
    ap.add_argument("--lib", default=None, help="(development) alternative build of liblfpsqp_hip.so to load")
    ap.add_argument("--placement", choices=["library", "first"], default="library",
                    help="where the basis and the solver's n-vectors are allocated decides between two speeds of the fused kernel (FINDINGS.md 6). "
                         "library (default): the library's own policy -- lfpsqp_basis_work_alloc_placed, what optimize() uses; "
                         "first: plain first allocations (policy off)")
    ap.add_argument("--placement-tries", type=int, default=3, help="candidate allocations per placed buffer (lfpsqp_ctx_set_placement)")
    ap.add_argument("--watchdog-seconds", type=float, default=1500.0,
                    help="dump every thread's Python stack to stderr and exit non-zero if the run takes longer (0 = off): a stalled "
                         "rendezvous or collective then fails with a diagnosis instead of hanging the caller")
    args = ap.parse_args(argv)
    global SIDE_SCALE
    SIDE_SCALE = args.side_scale
    n, m, K, W = int(args.n), args.m, args.steps, args.warmup

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): this process has not touched the GPU (nothing but argparse ran), so it
        # starts the ranks itself -- one `torch.distributed.run` child, one process per GPU -- relays rank 0's line and exits
        # with the child's code.  A child process, never an exec: the ranks initialise the GPU, the parent never does.
        return self_launch(args, argv if argv is not None else sys.argv[1:])
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one process per GPU: "
                 f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., or call "
                 f"`python bench.py --gpus {args.gpus}` without a launcher and let it start its ranks)")
    watchdog = False
    if args.watchdog_seconds > 0:
        import faulthandler
        try:
            faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True, file=sys.__stderr__)
            watchdog = True
        except (ValueError, OSError, AttributeError):       # no real stderr (in-process test harness): run without
            pass
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: keep gloo's pair connections on the loopback interface instead of whatever the host name resolves to
            # (it may not resolve at all on these boxes, and a resolver time-out per connection stalls the rendezvous)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")        # RCCL's bootstrap sockets likewise (the data path is xGMI / P2P)
        # control plane only, on CPU tensors (gloo): barriers, the RCCL id, the max over ranks.  The data path is the library's own
        # RCCL communicator; the torch-nccl callback (--comm torch, or the fallback) gets a separate nccl group when it is needed.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import lfpsqp_jl_amd as L

    if args.lib:
        lib = L.load_library(args.lib)
    dev = local_rank if args.device is None else args.device
    comm_used, comm_probe = "none", None
    if world > 1:
        ctx, comm_used, comm_probe = bring_up_comm(args, L, dist, dev, rank, world, lib)
    else:
        ctx = L.Context(dev, lib)
    dev_uuid = ctx.device_uuid() if rank == 0 else None       # of the device this rank computes on
    dev_uuids = [ctx.device_uuid()]
    if world > 1:                                             # every rank's device (control plane: gloo): N ranks on N DIFFERENT devices, or sharing one
        gathered = [None] * world
        dist.all_gather_object(gathered, dev_uuids[0])
        dev_uuids = gathered
    r0, r1 = ctx.shard_range(n, rank, world)
    n_loc = r1 - r0

    # ---- workload, generated on the device -------------------------------
    def make_basis(placed=False, Zc=None):
        Zc = Zc if Zc is not None else ctx.matrix(n_loc, m, placed=placed)
        if args.basis == "orthonormal" and hasattr(L, "orthonormalize_"):
            Zc.hash_fill(1, r0, n, 1.0)
            L.orthonormalize_(Zc, n_global=n)
            return Zc, "orthonormalised hash matrix (CholeskyQR2 on device)"
        Zc.hash_fill(1, r0, n, 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n))))
        return Zc, "scaled hash matrix (columns orthonormal to O(sqrt(m/n)))"

    b = ctx.vector(n_loc).hash_fill(4, r0)
    # Placement.  Where the basis and the work vectors land in memory decides between two speeds of the fused kernel, 10-15 % apart (FINDINGS.md 6;
    # a property of the PAIR of allocations).  Default: the library's own policy, the one optimize() uses -- the basis from
    # lfpsqp_mat_alloc_placed, ProjCGWork and the operator diagonal from lfpsqp_vecs_alloc_placed (candidate allocations, the fused kernel
    # timed on each, the fastest kept; every rank makes the same calls, the choice is local).
    emulated = "emulator" in ctx.device_name

    placement = {"policy": args.placement}
    first_probe = None          # (trial time of the FIRST allocations, trial time of the kept ones): what the policy changed
    tries = 1 if args.placement == "first" else max(1, args.placement_tries)
    ctx.set_placement(tries)
    work = L.ProjCGWork(ctx, n_loc, m, against=("new", n_loc, m), extra=1)       # the basis and its work vectors, allocated together
    pt = ctx.placement_info()
    Z, basis_desc = make_basis(Zc=work.basis)
    A = L.DiagOperator(0.0, work.placed_extra[0].hash_fill(3, r0, 4.5, 5.5))
    x = ctx.vector(n_loc)
    flat = sorted(pt[2])
    first_probe = (pt[2][0], pt[2][pt[1]]) if len(pt[2]) > 1 else None
    placement.update({"tries_per_buffer": tries, "pairs_tried": pt[0], "kept_pair": pt[1], "probe_F_ms": [round(t, 4) for t in pt[2]],
                      "probe_min_median_max_F_ms": ([round(flat[0], 4), round(flat[len(flat) // 2], 4), round(flat[-1], 4)] if flat else None),
                      "note": "lfpsqp_basis_work_alloc_placed: every (basis candidate, work-vector-set candidate) pair tried with the fused kernel "
                              "itself (on zeros, two rounds), the fastest pair kept -- the same call optimize() makes for Z and ProjCGWork"})
    U = L.DeviceBasis(Z)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- warmup, then EXACTLY K timed steps of the running solve ------------------------------------
    # The W warmup iterations run in one lfpsqp_projcg call (which also does the solve's set-up: two more passes over U);
    # the timed region is a second call that RESUMES that solve for K iterations (LFPSQP_PROJCG_RESUME), so a "step" is
    # one projected-CG iteration and nothing else.  With W = 0 there is nothing to resume: the timed call then contains
    # the set-up as well (reported in config.timed_region).
    # untimed device warm-up: a GPU coming out of idle runs the same loop ~6 % slower for its first ~2 s (measured on fresh boxes:
    # 543 / 546 it/s without, 577 / 581 with).  The number of calls is the SAME on every rank (each call contains collectives):
    # one call is timed, the slowest rank's time sets the count.
    prewarm_iters = 0
    if args.prewarm_seconds > 0:
        t_pw = time.perf_counter()
        L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=50, work=work, n_global=n, want_lambda=False)
        ctx.sync()
        t_call = max(time.perf_counter() - t_pw, 1e-4)
        if dist is not None:
            import torch
            tt_ = torch.tensor([t_call], dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            t_call = float(tt_[0])
        ncalls = min(int(math.ceil(args.prewarm_seconds / t_call)), 2000)
        for _ in range(ncalls - 1):
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=50, work=work, n_global=n, want_lambda=False)
        ctx.sync()
        prewarm_iters = 50 * ncalls
    resumed = W > 0
    if W > 0:
        iw, _ = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=W, work=work, n_global=n, want_lambda=False)
        assert iw == W
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    iters, nr = L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=K, work=work, n_global=n, want_lambda=False, resume=resumed)
    ctx.sync()
    elapsed = time.perf_counter() - t0          # this rank's K steps (they contain the collectives, so no rank finishes a step early);
    if dist is not None:                        # the closing barrier of the bracket, then the MAX over ranks below -- the barrier's own
        dist.barrier()                          # host-side (gloo) latency is not part of the K steps
    prof_ms, prof_cnt = ctx.profile_read()
    ctx.set_profiling(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    assert iters == W + K, f"expected {W + K} iterations, got {iters} (nr={nr})"
    assert math.isfinite(nr)
    # the same window five more times (the solve keeps running: K further iterations each) -- `value` is the FIRST window, as the contract
    # says; these show how much one 20-iteration window of ~35 ms scatters
    x_norm = L.nrm2(x)
    more_windows = []
    if world == 1 and resumed:
        L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=W, work=work, n_global=n, want_lambda=False)      # (the norm above ended the first solve)
        for _ in range(5):
            ctx.sync(); t0w = time.perf_counter()
            L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=K, work=work, n_global=n, want_lambda=False, resume=True)
            ctx.sync(); more_windows.append(K / (time.perf_counter() - t0w))
    # the same K iterations as ONE fresh call (set-up inside the clock): what a caller of projcg! with maxit = K sees
    barrier()
    t0 = time.perf_counter()
    L.projcg_(x, None, A, U, b, None, tol=1e-300, maxit=K, work=work, n_global=n, want_lambda=False)
    ctx.sync()
    single_call = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([single_call], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        single_call = float(t[0])

    # ---- per-kernel roofline numbers (HIP events on the library's stream) ---------------
    def avg(slot):
        return prof_ms[slot] / prof_cnt[slot] if prof_cnt[slot] else float("nan")
    fused = prof_cnt[3] > 0                              # one pass over U per iteration (lfpsqp_projcg default)
    k1, k2, k3, kf = avg(0), avg(1), avg(2), avg(3)
    # algorithmic bytes per launch (FINDINGS.md §5).  K1 reads x,d,g(,a) writes x,d.  Fused F: U once + reads g,d,a, writes g.
    # Two-pass fallback: K2 reads d,g,a + U; K3 reads d,g,a writes g + U.
    bytes_k1 = (40.0 if fused else 48.0) * n_loc       # the fused flow's K1 needs no A (d'Ad comes out of F's sums)
    bytes_k2 = 8.0 * n_loc * m + 24.0 * n_loc + 8.0 * m
    bytes_k3 = 8.0 * n_loc * m + 32.0 * n_loc + 8.0 * m
    bytes_kf = 8.0 * n_loc * m + 32.0 * n_loc + 8.0 * m
    gbs = lambda by, ms: by / (ms * 1e-3) / 1e9
    # plain matvecs (the "achieved HBM GB/s on J matvec" half of the metric)
    v = ctx.vector(n_loc).hash_fill(5, r0)
    tt = ctx.vector(m).hash_fill(6)
    y = ctx.vector(n_loc)
    reps = 5
    L.gemv_t(Z, v, tt); L.gemv_n(Z, tt, y)
    ctx.timer_begin()
    for _ in range(reps):
        L.gemv_t(Z, v, tt)
    ms_t = ctx.timer_end() / reps
    ctx.timer_begin()
    for _ in range(reps):
        L.gemv_n(Z, tt, y, 1.0, 1.0)
    ms_n = ctx.timer_end() / reps
    bytes_t = 8.0 * n_loc * m + 8.0 * n_loc + 8.0 * m
    bytes_n = 8.0 * n_loc * m + 16.0 * n_loc + 8.0 * m

    out = {
        "metric": "projected-CG iters/sec at n=1e7, m=128 fp64",
        "value": K / elapsed,
        "unit": "iters/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"sustained projcg, dense basis n={n}, m={m}, A=diag(5.5+4.5u) (kappa 10), tol=1e-300 ({shape_name(n, m)})",
                   "timed_region": (f"{K} iterations of a running solve (resumed after the {W} warmup iterations; set-up outside)"
                                    if resumed else f"one projcg call: set-up (2 passes over U) + {K} iterations"),
                   "prewarm": f"{prewarm_iters} untimed iterations ({args.prewarm_seconds:g} s) before the warmup steps",
                   "placement": placement,
                   "n": n, "m": m, "rows_per_gpu": n_loc, "basis": basis_desc,
                   "parallelism": f"row-sharded x{world}" if world > 1 else "single GPU",
                   "comm": comm_used, "comm_probe": comm_probe, "device": ctx.device_name, "device_uuid": dev_uuid, "device_uuids": dev_uuids},
        "roofline": ({"bound": "hbm", "kernel": "onepass_kernel<PcgFuseE> (F: rp = g + alpha*A*d, gp = rp - U*Utr, g = gp, U'gp, U'(A gp): "
                                                 "ONE pass over U per projected-CG iteration)",
                      "achieved": gbs(bytes_kf, kf), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": gbs(bytes_kf, kf) / HBM_PEAK_GBS, "traffic": None,
                      "avg_launch_ms": kf, "algorithmic_bytes": bytes_kf} if fused else
                     {"bound": "hbm", "kernel": "gemv_t_kernel<PcgStepV> (K2: rp = g + alpha*A*d formed on the fly, fused with U'rp)",
                      "achieved": gbs(bytes_k2, k2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": gbs(bytes_k2, k2) / HBM_PEAK_GBS, "traffic": None,
                      "avg_launch_ms": k2, "algorithmic_bytes": bytes_k2}),
        "kernels": ({"K1_dir_dAd": {"ms": k1, "GBs": gbs(bytes_k1, k1)},
                     "F_fused_projection": {"ms": kf, "GBs": gbs(bytes_kf, kf)},
                     "iteration_algorithmic_GB": (bytes_k1 + bytes_kf) / 1e9,
                     "iteration_GBs": (bytes_k1 + bytes_kf) * K / elapsed / 1e9,
                     "reference_formulation_GB": (16.0 * n_loc * m + 96.0 * n_loc) / 1e9} if fused else
                    {"K1_dir_dAd": {"ms": k1, "GBs": gbs(bytes_k1, k1)},
                     "K2_step_UTrp": {"ms": k2, "GBs": gbs(bytes_k2, k2)},
                     "K3_proj_dots": {"ms": k3, "GBs": gbs(bytes_k3, k3)},
                     "iteration_algorithmic_GB": (bytes_k1 + bytes_k2 + bytes_k3) / 1e9,
                     "iteration_GBs": (bytes_k1 + bytes_k2 + bytes_k3) * K / elapsed / 1e9}),
        "matvec": {"gemv_t": {"ms": ms_t, "GBs": gbs(bytes_t, ms_t), "frac": gbs(bytes_t, ms_t) / HBM_PEAK_GBS},
                   "gemv_n": {"ms": ms_n, "GBs": gbs(bytes_n, ms_n), "frac": gbs(bytes_n, ms_n) / HBM_PEAK_GBS}},
    }

    # HBM traffic of the roofline kernel: PMC counters cannot be read from inside the timed run, they come from separate
    # rocprofv3 --pmc passes of this same command (tools/gpu_final.sh; tools/pmc_summary.py applies the guide's gfx950
    # correction to FETCH_SIZE).  The committed summary counts as a measurement of THIS build only if it carries the
    # checksum of the library loaded here; otherwise `traffic` stays null and the old figure is quoted under another key.
    try:
        import hashlib
        pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("pmc_summary.json"))
        if pmc and world == 1 and n == 10_000_000 and m == 128:
            full = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
            summ = full.get("kernels", full)
            lib_sha = hashlib.sha256(open(ctx.L.path, "rb").read()).hexdigest()
            for name, v in summ.items():
                if isinstance(v, dict) and ("onepass_kernel<lfpsqp::PcgFuseE<false, false>" if fused else "gemv_t_kernel<lfpsqp::PcgStepV") in name:
                    rec = {"bytes": v["traffic_GB"] * 1e9, "source": f"profiles/{pmc[-1]} ({v['launches']} launches)"}
                    if full.get("lib_sha256") == lib_sha:
                        out["roofline"]["traffic"] = rec["bytes"]
                        out["roofline"]["traffic_source"] = rec["source"] + ", same library build (sha256 match)"
                    else:
                        out["roofline"]["traffic_from_committed_profile"] = dict(rec, note="PMC passes of another build of the library")
    except (OSError, ValueError, KeyError, AttributeError):
        pass

    if not args.no_extras and world == 1:                 # single-GPU diagnostics; the N > 1 runs measure the metric only
        out["extras"] = extras(ctx, L, n, m, n_loc, r0, Z, gbs)
    if fused:
        # what a caller WITHOUT the policy gets: the pair of FIRST allocations of this process is trial (0, 0); its trial time, scaled by
        # (F in the timed run) / (trial time of the kept pair) -- the trial runs on zeros, 2-5 % faster than on data -- estimates its F
        f0 = kf if first_probe is None else first_probe[0] * kf / first_probe[1]
        other = elapsed / K * 1e3 - kf                                      # everything of a step that is not the fused kernel
        out["first_allocation"] = {"F_ms": f0, "frac": gbs(bytes_kf, f0) / HBM_PEAK_GBS, "iters_per_s": 1e3 / (f0 + other),
                                   "estimated": first_probe is not None,
                                   "note": ("the timed run itself: placement policy off" if first_probe is None else
                                            "fused kernel on the FIRST allocations of the basis and the work vectors in this process (trial pair 0 of "
                                            "the policy, scaled from the trial's zeros to data by the kept pair's timed / trial ratio), and the "
                                            "iteration rate that implies with this run's other per-step costs: what a caller without the policy gets")}
    if more_windows:
        out["further_windows_iters_per_s"] = [round(v, 1) for v in more_windows]
    out["check"] = {"x_norm": x_norm, "nr": nr, "iters": iters}    # global ||x|| after the W + K iterations (sanity / N-rank agreement)
    out["single_call"] = {"value": K / single_call, "ms_per_step": single_call / K * 1e3,
                          "note": f"one lfpsqp_projcg call with maxit = {K}: set-up (x = 0, r = -b, U'r, first projection: 2 passes over U) + {K} iterations"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(int(args.cpu_sample_n), m, n, args.side_scale)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if watchdog:
        faulthandler.cancel_dump_traceback_later()
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: run the N ranks as ONE child (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>`), pass its stdout (rank 0's JSON
    line) and stderr through, return its exit code.  Nothing here initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL / hipIpc between the ranks)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    import signal
    child = subprocess.Popen(cmd, env=env, start_new_session=True)      # its own process group: a timeout ends the launcher AND its ranks
    try:
        # the ranks' own watchdogs end a stalled run with a diagnosis; this limit only backs them up
        rc = child.wait(timeout=(args.watchdog_seconds + 120.0) if args.watchdog_seconds > 0 else None)
    except subprocess.TimeoutExpired:
        print("[bench] the ranks did not finish in time", file=sys.stderr)
        os.killpg(child.pid, signal.SIGKILL)
        child.wait()
        rc = 124
    except KeyboardInterrupt:
        os.killpg(child.pid, signal.SIGTERM)
        child.wait()
        raise
    if rc != 0:
        sys.exit(rc)
    return None


def bring_up_comm(args, L, dist, dev, rank, world, lib):
    """The all-reduce transport of an N-rank run -> (context, description, probe record).  Every decision is taken from values all ranks
    share (gloo all-reduces on the control plane), so the ranks cannot disagree about the transport; a transport that was ASKED for by
    name fails loudly instead of falling back."""
    import torch
    m_pay = 2 * args.m + 5                                     # the projected-CG iteration's payload (doubles)

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag[0]) == 1

    def check_and_time(ctx, reps=200):
        """a known sum (rank r contributes r + 1 + j/1024 in slot j) -> (correct on every rank, microseconds per all-reduce, max over ranks)"""
        import numpy as np
        ok, us = True, float("inf")
        try:
            v = ctx.vector(m_pay, (rank + 1.0) + np.arange(m_pay) / 1024.0)
            ctx.check(ctx.L.lfpsqp_allreduce(ctx.h, v.h, m_pay))
            want = world * (world + 1) / 2.0 + world * np.arange(m_pay) / 1024.0
            ok = bool(np.array_equal(v.download(), want))             # (exact: small dyadic rationals)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.check(ctx.L.lfpsqp_allreduce(ctx.h, v.h, m_pay))
            ctx.sync()
            us = (time.perf_counter() - t0) / reps * 1e6
            v.free()
        except L.LfpsqpError as e:
            print(f"[bench] rank {rank}: all-reduce check failed: {e}", file=sys.stderr)
            ok = False
        t = torch.tensor([us], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return agree(ok), float(t[0])

    def try_p2p():
        ctx = L.Context(dev, lib)
        ok, kind = True, "none"
        try:
            if args.p2p_allow_coarse:
                ctx.comm_p2p_allow_coarse(True)
            handle = ctx.comm_p2p_export()
        except L.LfpsqpError as e:
            print(f"[bench] rank {rank}: p2p mailbox export failed: {e}", file=sys.stderr)
            ok, handle = False, b"\0" * 64
        box = [None] * world
        dist.all_gather_object(box, handle)
        if agree(ok):
            try:
                ctx.comm_init_p2p(rank, world, box)
                kind = ctx.comm_p2p_info()[0]
            except L.LfpsqpError as e:
                print(f"[bench] rank {rank}: p2p init failed: {e}", file=sys.stderr)
                ok = False
            ok = agree(ok)
        else:
            ok = False
        kinds = [None] * world
        dist.all_gather_object(kinds, kind)
        return ctx, ok, "/".join(sorted(set(kinds)))

    def try_rccl():
        ctx = L.Context(dev, lib)
        ok, why = True, None
        try:
            box = [ctx.comm_unique_id() if rank == 0 else None]
        except L.LfpsqpError as e:
            box, ok, why = [None], False, f"unique id: {e}"
            print(f"[bench] rank {rank}: RCCL unique id failed: {e}", file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok = False
        if agree(ok):
            try:
                ctx.comm_init_rccl(rank, world, box[0])
            except L.LfpsqpError as e:
                ok, why = False, str(e)
                print(f"[bench] rank {rank}: native RCCL init failed: {e}", file=sys.stderr)
            ok = agree(ok)
        else:
            ok = False
        whys = [None] * world
        dist.all_gather_object(whys, why)                      # (the same record on every rank: rank 0's line quotes the first refusal)
        return ctx, ok, next((w for w in whys if w), None)

    if args.comm == "host-gloo":
        from lfpsqp_jl_amd.distributed import host_staged_allreduce_callback, torch_allreduce_callback
        ctx = L.Context(dev, lib)
        emulated = "emulator" in ctx.device_name      # tests/test_bench_harness.py: the library's buffers are host memory
        ctx.comm_init_callback(rank, world, torch_allreduce_callback(None) if emulated else host_staged_allreduce_callback(dev))
        return ctx, "host-gloo (functional test)", None
    if args.comm == "torch":
        from lfpsqp_jl_amd.distributed import torch_allreduce_callback
        ctx = L.Context(dev, lib)
        torch.cuda.set_device(dev)
        ctx.comm_init_callback(rank, world, torch_allreduce_callback(dev, dist.new_group(backend="nccl")))
        return ctx, "torch-nccl-callback", None
    devs = [None] * world
    dist.all_gather_object(devs, dev)
    shared = len(set(devs)) < world                  # ranks sharing a GPU (the 1-GPU functional runs): RCCL refuses duplicate devices
    probe = {"payload_doubles": m_pay, "ranks_share_a_gpu": shared}
    cands = []
    if args.comm in ("auto", "p2p"):
        ctx, ok, kind = try_p2p()
        us = None
        if ok:
            ok, us = check_and_time(ctx)
        probe["p2p"] = {"ok": ok, "mailbox_memory": kind, "us_per_allreduce": us}
        if ok:
            cands.append((us, "p2p", ctx, f"p2p (one-shot all-reduce over hipIpc-mapped mailboxes, {kind} memory)"))
        else:
            ctx.close()
            if args.comm == "p2p":
                sys.exit("bench.py --comm p2p: the peer-to-peer transport did not come up on every rank (see stderr); no fallback for a transport asked for by name")
    emulated = False
    if args.comm == "auto":
        c0 = L.Context(dev, lib)
        emulated = "emulator" in c0.device_name                 # (tests/test_bench_harness.py: no GPU behind the library, RCCL is not loaded at all)
        c0.close()
        if emulated:
            probe["rccl"] = {"ok": False, "skipped": "CPU emulator build of the library"}
    if args.comm == "rccl" or (args.comm == "auto" and not emulated):
        # (auto probes RCCL also when the ranks share a GPU: ncclCommInitRank then refuses on every rank -- "invalid usage", duplicate device,
        # within a second or two, measured on the MI355X box -- and the record says so; the run carries on with the transport that passed)
        ctx, ok, why = try_rccl()
        us = None
        if ok:
            ok, us = check_and_time(ctx)
        probe["rccl"] = {"ok": ok, "us_per_allreduce": us}
        if why:
            probe["rccl"]["refused"] = why
        if ok:
            cands.append((us, "rccl", ctx, "rccl (library-native communicator)"))
        else:
            ctx.close()
            if args.comm == "rccl":
                sys.exit("bench.py --comm rccl: the RCCL communicator did not come up on every rank (see stderr); no fallback for a transport asked for by name")
    if cands:
        cands.sort(key=lambda c: c[0])               # (times are max-over-ranks values: identical on every rank)
        for c in cands[1:]:
            c[2].close()
        probe["chosen"] = cands[0][1]
        return cands[0][2], cands[0][3] + (" [auto: faster of the transports that passed their check]" if args.comm == "auto" else ""), probe
    # auto, and neither library transport works: LOUDLY down to the torch callback
    print("[bench] WARNING: neither the p2p nor the RCCL transport of the library came up; falling back to the torch.distributed(nccl) callback",
          file=sys.stderr, flush=True)
    from lfpsqp_jl_amd.distributed import torch_allreduce_callback
    ctx = L.Context(dev, lib)
    torch.cuda.set_device(dev)
    ctx.comm_init_callback(rank, world, torch_allreduce_callback(dev, dist.new_group(backend="nccl")))
    probe["chosen"] = "torch"
    return ctx, "torch-nccl-callback (FALLBACK: library transports failed)", probe


def shape_name(n, m):
    if (n, m) == (10_000_000, 128):
        return "BASELINE configs[2] shape"
    if (n, m) == (5_000_000, 512):
        return "per-GPU shard of BASELINE configs[4]: n=4e7, m=512 over 8 GPUs"
    if (n, m) == (40_000_000, 512):
        return "BASELINE configs[4] shape"
    return "not a BASELINE shape"


def extras(ctx, L, n, m, n_loc, r0, Z, gbs):
    """Timings of the other hot-path rows at the bench size: the tangent setup
    (lfpsqp_factorize: MFMA Gram + rmul + host m x m algebra) and one fused Newton-retraction step."""
    import numpy as np
    J = ctx.matrix(n_loc, m).hash_fill(1, r0, n)
    Z2 = ctx.matrix(n_loc, m)
    Wg = np.zeros((m, m), order='F')
    L.ksvd_(J, Z2, W=Wg)                                      # warm
    ctx.sync(); t0 = time.perf_counter(); S, Vt, rank = L.ksvd_(J, Z2, W=Wg); ctx.sync(); fact_ms = (time.perf_counter() - t0) * 1e3
    # the same factorisation with the basis left in factored form U = J W (no basis-forming product: what optimize() runs per outer iteration)
    L.ksvd_(J, None, W=Wg)
    ctx.sync(); t0 = time.perf_counter(); L.ksvd_(J, None, W=Wg); ctx.sync(); fact_factored_ms = (time.perf_counter() - t0) * 1e3
    # ... and warm-started from the previous call's eigenvectors (lfpsqp_factorize_hint: what optimize() runs from its second outer iteration on)
    L.ksvd_(J, None, W=Wg, Vt_prev=Vt)
    ctx.sync(); t0 = time.perf_counter(); L.ksvd_(J, None, W=Wg, Vt_prev=Vt); ctx.sync(); fact_warm_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); L.gram(J); gram_ms = (time.perf_counter() - t0) * 1e3
    W = np.eye(m)
    t0 = time.perf_counter(); L.rmul(J, W, Z2); ctx.sync(); rmul_ms = (time.perf_counter() - t0) * 1e3
    L.ksvd_(J, Z2, W=Wg)
    # Newton retraction on c(x) = J'x - b from a perturbed point: 24 iterations forced by tol = 0 (the initial c! pass is
    # amortised).  "one stream": the basis carries its generator (Z = J*W), both products of a step run over J alone;
    # "two streams": generator unknown, the step reads Z and J.
    xs = ctx.vector(n_loc).hash_fill(2, r0)
    bdev = ctx.vector(m); L.gemv_t(J, xs, bdev)
    cons = L.DeviceConstraints(J, m, bdev.download())
    pert = ctx.vector(n_loc).hash_fill(7, r0, 1e-3, 0.0)
    xt = ctx.vector(n_loc); L.waxpby(1.0, xs, 1.0, pert, xt)
    xnew, cval = ctx.vector(n_loc), np.zeros(m)
    # Timed the way `value` is: MANY iterations of one running solve (tol = 0 forces exactly `its`), so that the call's set-up (the first
    # c! pass, the H2D copies of the small factors) weighs < 1 %; beside the wall time per iteration the KERNEL CHAIN of an iteration from
    # the library's own HIP events (every 4th launch of a kernel family is bracketed): what is left between the two is launch gaps.
    its_long = max(8, int(round(200 * SIDE_SCALE)))
    nr_ms, nr_chain = {}, {}
    for label, basis in (("one_stream", L.DeviceBasis(Z2, generator=(J, Wg))), ("two_streams", L.DeviceBasis(Z2))):
        nits = its_long if label == "one_stream" else 24
        nr = L.NR(basis, S, Vt, 0.0, nits, L.NRWork(m), False, None)
        L.retract_(cval, xnew, cons, xt, xs, nr)                  # warm
        ctx.set_profiling(True)
        ctx.sync(); t0 = time.perf_counter(); flag, it, _ = L.retract_(cval, xnew, cons, xt, xs, nr); ctx.sync()
        nr_ms[label] = (time.perf_counter() - t0) * 1e3 / max(it, 1)
        pms, pcnt = ctx.profile_read(); ctx.set_profiling(False)
        nr_chain[label] = (pms[4] / pcnt[4]) if pcnt[4] else None          # the one-pass step kernel (slot 4)
    # pcg! of the default (ProjPenalty) retraction: iterations of (J'J + mu I) x = b forced by tol = 0
    from lfpsqp_jl_amd.projpenalty import _JacPlain
    w = L.ProjPenaltyWork(ctx, m, n_loc, False)
    xp, rp_ = ctx.vector(n_loc), ctx.vector(n_loc)
    pcg_ms = pcg_chain = None
    for rep in range(2):
        xp.fill(0.0); rp_.copy_from(xs)
        ctx.set_profiling(True)
        ctx.sync(); t0 = time.perf_counter(); flag, pit = L.pcg_(1e-2, _JacPlain(J, w), L.no_precondition, xp, rp_, w.p, w.z, None, 0.0, its_long); ctx.sync()
        pcg_ms = (time.perf_counter() - t0) * 1e3 / max(pit, 1)
        pms, pcnt = ctx.profile_read(); ctx.set_profiling(False)
        if pcnt[5] and pcnt[6]:
            pcg_chain = {"fused_kernel_ms": pms[5] / pcnt[5], "P3_vector_kernel_ms": pms[6] / pcnt[6]}
    pcg_bytes = 8.0 * n_loc * m + 80.0 * n_loc               # one-pass iteration: J once + 10 n-vector passes
    # several trial points of one linesearch retracted together (lfpsqp_retract_nr_batch): one pass over J per Newton step for all of them --
    # 4 in the default EXACT mode (VALU form of the one-pass kernel on the single-trial step's spans: bit for bit the one-by-one retractions), 8 and 16
    # in the opt-in matrix-core mode (lfpsqp_ctx_set_nr_batch_mode); 100 steps forced by tol = 0
    nrb_ms, nrb_kernel_ms = {}, {}
    for nbt in (4, 8, 16):
        ctx.set_nr_batch_mode(nbt > 4)
        xts = [ctx.vector(n_loc) for _ in range(nbt)]
        xns = [ctx.vector(n_loc) for _ in range(nbt)]
        for j, xt_ in enumerate(xts):
            L.waxpby(1.0, xs, 0.5 ** j, pert, xt_)
        nrb = L.NR(L.DeviceBasis(Z2, generator=(J, Wg)), S, Vt, 0.0, max(6, int(round(100 * SIDE_SCALE))), L.NRWork(m), False, None)
        cvs = np.zeros((nbt, m))
        for rep in range(2):
            ctx.set_profiling(True)
            ctx.sync(); t0 = time.perf_counter(); got = L.retract_nr_batch_(cvs, xns, cons, xts, xs, nrb); ctx.sync()
            if got is not None:
                nrb_ms[nbt] = (time.perf_counter() - t0) * 1e3 / max(got[0][1], 1)
            pms, pcnt = ctx.profile_read(); ctx.set_profiling(False)
            nrb_kernel_ms[nbt] = (pms[7] / pcnt[7]) if pcnt[7] else None
        for v_ in xts + xns:
            v_.free()
    ctx.set_nr_batch_mode(False)
    nr_batch_ms, nr_batch_kernel_ms = nrb_ms.get(4), nrb_kernel_ms.get(4)
    # the fused projected-CG iteration on the basis in factored form (streams J, applies W in the post-kernel) against the materialised Z2
    Af = L.DiagOperator(0.0, ctx.vector(n_loc).hash_fill(3, r0, 4.5, 5.5))
    bf, xf, wf = ctx.vector(n_loc).hash_fill(4, r0), ctx.vector(n_loc), L.ProjCGWork(ctx, n_loc, m)
    pf = {}
    for tag, basis in (("materialised", L.DeviceBasis(Z2)), ("factored", L.DeviceBasis(None, m, generator=(J, Wg)))):
        L.projcg_(xf, None, Af, basis, bf, None, tol=1e-300, maxit=5, work=wf, n_global=n, want_lambda=False)
        ctx.sync(); t0 = time.perf_counter()
        itf, _ = L.projcg_(xf, None, Af, basis, bf, None, tol=1e-300, maxit=30, work=wf, n_global=n, want_lambda=False)
        ctx.sync(); pf[tag] = (time.perf_counter() - t0) * 1e3 / max(itf, 1)
        pf[tag + "_xnorm"] = L.nrm2(xf)
    # ... and over a matrix VIEW diag(rs) J + u w' (lfpsqp_mat_view: the streamed constraint gradients of the nonlinear class, whose jac! then writes
    # two n-vectors instead of n x m doubles): the same fused kernel behind the functor wrapper, the same iterates (rs = 1, u w' = 0 here)
    view_info = None
    try:
        rs_v, u_v, w_v = ctx.vector(n_loc).fill(1.0), ctx.vector(n_loc), ctx.vector(m)
        Jv = J.view(rs_v, u_v, w_v)
        bv = L.DeviceBasis(None, m, generator=(Jv, Wg))
        L.projcg_(xf, None, Af, bv, bf, None, tol=1e-300, maxit=5, work=wf, n_global=n, want_lambda=False)
        ctx.sync(); t0 = time.perf_counter()
        itv, _ = L.projcg_(xf, None, Af, bv, bf, None, tol=1e-300, maxit=30, work=wf, n_global=n, want_lambda=False)
        ctx.sync(); view_ms = (time.perf_counter() - t0) * 1e3 / max(itv, 1)
        view_info = {"projcg_iter_ms_over_a_view": view_ms, "projcg_iter_ms_factored": pf["factored"],
                     "xnorm_rel_diff": abs(L.nrm2(xf) - pf["factored_xnorm"]) / pf["factored_xnorm"],
                     "note": "projected-CG iteration with the generator of the factored basis given as a view diag(rs) J + u w' (rs = 1, u = 0) "
                             "against the plain matrix: what streaming the gradients of the nonlinear class costs in the hot loop (16 more bytes per row)"}
        Jv.free()
        for v_ in (rs_v, u_v, w_v):
            v_.free()
    except Exception as e:                                      # (a side measurement must never take the bench line down)
        view_info = {"error": repr(e)[:200]}
    for v_ in (bf, xf, wf.g, wf.d, wf.rp, Af.dg):
        v_.free()
    nr1_bytes = 8.0 * n_loc * m + 24.0 * n_loc                # J pass + xnew read/write + v
    nr2_bytes = 16.0 * n_loc * m + 24.0 * n_loc               # Z pass + J pass + xnew read/write + v
    flop = 2.0 * n_loc * m * m
    # sparse constraint gradients (lfpsqp_spmat): a banded Jct with 4 nonzeros per row at the bench's n x m; both products against
    # their algorithmic bytes (12 per nonzero + the vectors; beyond the 256 MB infinity cache only from ~2e7 nonzeros on)
    sparse = None
    try:
        k_sp = 4
        ii = np.arange(n_loc, dtype=np.int64)
        rows_sp = np.repeat(ii, k_sp)
        cols_sp = (((ii * m) // max(n_loc, 1))[:, None] + np.arange(k_sp)[None, :]) % m
        vals_sp = np.ones(n_loc * k_sp)
        Ssp = L.SparseMatrix(ctx, n_loc, m, rows_sp, cols_sp.ravel(), vals_sp)
        tsp, ysp = ctx.vector(m).hash_fill(6), ctx.vector(n_loc)
        L.spmv_t(Ssp, xs, tsp); L.spmv_n(Ssp, tsp, ysp)
        ctx.timer_begin()
        for _ in range(10):
            L.spmv_t(Ssp, xs, tsp)
        ms_st = ctx.timer_end() / 10
        ctx.timer_begin()
        for _ in range(10):
            L.spmv_n(Ssp, tsp, ysp)
        ms_sn = ctx.timer_end() / 10
        by_t = 12.0 * Ssp.nnz + 8.0 * n_loc
        by_n = 12.0 * Ssp.nnz + 8.0 * n_loc
        sparse = {"nnz": Ssp.nnz, "nnz_per_row": k_sp, "spmv_t_ms": ms_st, "spmv_t_GBs": gbs(by_t, ms_st), "spmv_n_ms": ms_sn,
                  "spmv_n_GBs": gbs(by_n, ms_sn), "dense_matrix_GB": 8.0 * n_loc * m / 1e9, "algorithmic_GB": by_t / 1e9}
        Ssp.free()
        # tangent setup of a sparse block (lfpsqp_factorize_sp: Gram matrix and basis-forming products from the nonzeros)
        # against the dense factorisation of the same matrix; random values so that the block has full rank
        vals_r = (np.random.default_rng(5).standard_normal((n_loc, k_sp)) + 2.0 * (np.arange(k_sp) == 0)).ravel()
        Sr = L.SparseMatrix(ctx, n_loc, m, rows_sp, cols_sp.ravel(), vals_r)
        Jd, Zs = Sr.to_dense(), ctx.matrix(n_loc, m)
        tf = {}
        Wn = np.zeros((m, m), order='F')
        for tag, Zarg, kw in (("dense", Zs, dict()), ("from_nonzeros", Zs, dict(Jsp=Sr)), ("from_nonzeros_factored", None, dict(Jsp=Sr, W=Wn))):
            L.ksvd_(Jd, Zarg, **kw)
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(3):
                Sg, _, rk = L.ksvd_(Jd, Zarg, **kw)
            ctx.sync(); tf[tag] = (time.perf_counter() - t0) * 1e3 / 3
        Sr.gram()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            Gs = Sr.gram()
        ctx.sync(); tg_sp = (time.perf_counter() - t0) * 1e3 / 3
        L.gram(Jd)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            Gd = L.gram(Jd)
        ctx.sync(); tg_d = (time.perf_counter() - t0) * 1e3 / 3
        sparse.update({"factorize_dense_ms": tf["dense"], "factorize_from_nonzeros_ms": tf["from_nonzeros"], "factorize_from_nonzeros_factored_basis_ms": tf["from_nonzeros_factored"], "factorize_rank": int(rk),
                       "factorize_cond": float(Sg[0] / Sg[-1]), "gram_dense_ms": tg_d, "gram_from_nonzeros_ms": tg_sp,
                       "gram_max_rel_diff": float(np.abs(Gs - Gd).max() / np.abs(Gd).max())})
        # Newton retraction on the same block (24 iterations forced by tol = 0): one dense pass per step vs the nonzeros alone
        Wg2 = np.zeros((m, m), order='F')
        Sg, Vtg, _ = L.ksvd_(Jd, Zs, W=Wg2, Jsp=Sr)
        bsp = ctx.vector(m); L.spmv_t(Sr, xs, bsp)
        tn = {}
        for tag, jsp in (("dense", None), ("on_nonzeros", Sr)):
            cons_s = L.DeviceConstraints(Jd, m, bsp.download(), Jsp=jsp)
            nrs = L.NR(L.DeviceBasis(Zs, generator=(Jd, Wg2)), Sg, Vtg, 0.0, 24, L.NRWork(m), False, None)
            L.retract_(cval, xnew, cons_s, xt, xs, nrs)
            ctx.sync(); t0 = time.perf_counter(); _, itn, _ = L.retract_(cval, xnew, cons_s, xt, xs, nrs); ctx.sync()
            tn[tag] = (time.perf_counter() - t0) * 1e3 / max(itn, 1)
        sparse.update({"nr_step_dense_ms": tn["dense"], "nr_step_on_nonzeros_ms": tn["on_nonzeros"]})
        # the projected CG itself on that block: fused dense iteration (one pass over Z) vs the basis in factored form on the nonzeros
        Asp = L.DiagOperator(0.0, ctx.vector(n_loc).hash_fill(3, r0, 4.5, 5.5))
        bsv, xsv, wsp = ctx.vector(n_loc).hash_fill(4, r0), ctx.vector(n_loc), L.ProjCGWork(ctx, n_loc, m)
        tp = {}
        for tag, basis in (("dense", L.DeviceBasis(Zs)), ("on_nonzeros", L.DeviceBasis(Zs, generator=(Jd, Wg2), sparse=Sr))):
            L.projcg_(xsv, None, Asp, basis, bsv, None, tol=1e-300, maxit=5, work=wsp, n_global=n, want_lambda=False)
            ctx.sync(); t0 = time.perf_counter()
            itp, _ = L.projcg_(xsv, None, Asp, basis, bsv, None, tol=1e-300, maxit=30, work=wsp, n_global=n, want_lambda=False)
            ctx.sync(); tp[tag] = (time.perf_counter() - t0) * 1e3 / max(itp, 1)
            tp[tag + "_xnorm"] = L.nrm2(xsv)
        sparse.update({"projcg_iter_dense_ms": tp["dense"], "projcg_iter_on_nonzeros_ms": tp["on_nonzeros"],
                       "projcg_xnorm_rel_diff": abs(tp["dense_xnorm"] - tp["on_nonzeros_xnorm"]) / tp["dense_xnorm"]})
        Sr.free()
    except Exception as e:      # diagnostics only
        sparse = {"error": repr(e)}
    rates = stream_rates(ctx, L, nbig=int(max(1 << 20, min(400_000_000, 40 * n_loc))))
    outer = outer_iteration(ctx, L, n_loc, m) if n_loc == n else None
    ops = None
    if n_loc == n and m >= 4:
        try:
            ops = operator_iterations(ctx, L, n, m, Z2)
        except Exception as e:      # diagnostics only
            ops = {"error": repr(e)[:200]}
    return {"outer_iteration": outer, "operators": ops, "stream_rates": rates, "placements": placements(ctx, L, n, m, n_loc, r0), "sparse": sparse, "factorize_ms": fact_ms, "factorize_factored_basis_ms": fact_factored_ms, "factorize_factored_basis_warm_ms": fact_warm_ms, "rank": int(rank), "gram_ms": gram_ms, "gram_TFLOPs_of_the_full_product": flop / gram_ms / 1e9,
            "gram_TFLOPs_executed": flop * gram_tile_share(m) / gram_ms / 1e9,
            "rmul_ms": rmul_ms, "rmul_TFLOPs": flop / rmul_ms / 1e9, "fp64_mfma_peak_TFLOPs": 78.6,
            "nr_step_ms": nr_ms["one_stream"], "nr_step_GBs": gbs(nr1_bytes, nr_ms["one_stream"]),
            "nr_step_two_streams_ms": nr_ms["two_streams"], "nr_step_two_streams_GBs": gbs(nr2_bytes, nr_ms["two_streams"]),
            "nr_step_kernel_ms": nr_chain["one_stream"], "nr_step_gap_note": "nr_step_ms = wall per iteration of a 200-iteration call; nr_step_kernel_ms = the one-pass step kernel alone (HIP events); the rest is the m x m Broyden kernel (~25 us) and the second-stage reduction",
            "nr_batch4_step_ms": nr_batch_ms, "nr_batch4_step_kernel_ms": nr_batch_kernel_ms,
            "nr_batch8_step_ms": nrb_ms.get(8), "nr_batch8_step_kernel_ms": nrb_kernel_ms.get(8), "nr_batch16_step_ms": nrb_ms.get(16), "nr_batch16_step_kernel_ms": nrb_kernel_ms.get(16),
            "nr_batch_note": "ms per Newton step of a 100-step call for 4 / 8 / 16 trial points without bounds (4: the default exact batch, VALU form of the one-pass kernel, bit for bit the one-by-one retractions; 8, 16: the opt-in matrix-core batch, v_mfma_f64_16x16x4_f64, csrc/nrbatch.h); wall includes the per-trial start (copy, first c! pass) and the trials' m x m Broyden kernels", "projcg_call_iter_materialised_ms": pf["materialised"], "projcg_call_iter_factored_ms": pf["factored"],
            "matrix_view": view_info, "projcg_factored_xnorm_rel_diff": abs(pf["factored_xnorm"] - pf["materialised_xnorm"]) / pf["materialised_xnorm"], "nr_iters_timed": its_long, "pcg_iter_ms": pcg_ms, "pcg_iter_GBs": gbs(pcg_bytes, pcg_ms), "pcg_iters_timed": int(pit), "pcg_iter_kernel_chain": pcg_chain,
            "note": "host wall clock around synchronous calls; gram/rmul include the small host<->device copies; the Gram kernel computes the upper "
                    "triangle only (16 x 16 tiles on and above the diagonal): gram_TFLOPs_executed counts that work, ..._of_the_full_product 2 n m^2"}


def operator_iterations(ctx, L, n, m, Z):
    """ms per projected-CG iteration with Hessians that are NOT diagonal, on the one-pass path (one rank): tridiagonal (lfpsqp_projcg_tridiag; the
    callback path lfpsqp_projcg_op with the same operator beside it) and diagonal + rank 4 (lfpsqp_projcg_lowrank), the diagonal operator on the
    same vectors as the yardstick.  Two solve lengths separate the iterations from what a solve costs outside them (the tridiagonal form's U'AU)."""
    import numpy as np
    U = L.DeviceBasis(Z)
    work = L.ProjCGWork(ctx, n, m)
    a, off, b, x = ctx.vector(n).hash_fill(3, 0, 4.5, 6.5), ctx.vector(n).hash_fill(15, 0, 0.8, 0.0), ctx.vector(n).hash_fill(4), ctx.vector(n)

    def per_iteration(A):
        t = {}
        for its in (10, 40):
            L.projcg_(x, None, A, U, b, None, tol=0.0, maxit=its, work=work, want_lambda=False)
            ctx.sync(); t0 = time.perf_counter()
            it, _ = L.projcg_(x, None, A, U, b, None, tol=0.0, maxit=its, work=work, want_lambda=False)
            ctx.sync(); t[its] = ((time.perf_counter() - t0) * 1e3, it)
        per = (t[40][0] - t[10][0]) / max(t[40][1] - t[10][1], 1)
        return per, t[10][0] - per * t[10][1]

    d_ms, d_out = per_iteration(L.DiagOperator(0.0, a))
    T = L.TridiagonalOperator(0.0, a, off)
    t_ms, t_out = per_iteration(T)
    T.fused = False
    tc_ms, _ = per_iteration(T)
    V = ctx.matrix(n, 4).hash_fill(17, 0, n, n ** -0.5)
    l_ms, l_out = per_iteration(L.LowRankOperator(0.0, a, V, 4, np.array([3.0, -0.4, 1.5, 0.7])))
    V.free()
    for v_ in (a, off, b, x, work.g, work.d, work.rp):
        v_.free()
    return {"diagonal_ms": d_ms, "tridiagonal_one_pass_ms": t_ms, "tridiagonal_callback_path_ms": tc_ms, "tridiagonal_ms_per_solve_outside_the_iterations": t_out - d_out,
            "diagonal_plus_rank4_one_pass_ms": l_ms,
            "note": "A = diag(2 .. 11) + couplings in [-0.8, 0.8] / + V diag(sigma) V' (4 columns); per iteration from solves of 10 and 40 iterations; the "
                    "tridiagonal form's set-up is U'AU, two weighted Gram passes on the matrix cores"}


def outer_iteration(ctx, L, n, m, iters=6):
    """Milliseconds per STEADY outer iteration of `optimize` (src/optimize.jl:257-443) at the bench's shape, on the device-resident nonlinear class
    with streamed gradients (c(x) = A'phi(x) - b, mixed kinds; f = |x - target|^2; Newton retraction): a callback after every iteration takes a
    timestamp with the device idle, tolerances off so that `iters` iterations run; the first interval (allocations, uploads) is reported apart.
    With the one-pass tangent step (lfpsqp_tangent_step, the default) and with the statement-by-statement sequence of the reference."""
    import numpy as np
    try:
        A = ctx.matrix(n, m).hash_fill(21, 0, n, 2.0 ** -11)
        cons = L.ElementwiseConstraints(ctx, A, np.zeros(m), kind=(np.arange(n) % 3).astype(np.float64), stream=True)
        x = ctx.vector(n).hash_fill(31, 0, 0.5, 0.0)
        cv = np.zeros(m)
        cons.jac_(cons.Jct, cv, x)
        cons.b = cons.b + cv                       # x feasible; the optimum of |x - target|^2 on the manifold lies nearby
        target = ctx.vector(n).hash_fill(41, 0, 0.5, 0.0)
        L.axpby(0.98, x, 0.02, target)
        prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target.download())
        x0h = x.download()
        out = {}
        for tag, fused in (("one_pass_tangent_step", True), ("statement_by_statement", False)):
            ctx.options.fused_tangent_step = fused
            best = None
            for rep in range(2):
                stamps = []

                def cb(i, xx):
                    ctx.sync()
                    stamps.append(time.perf_counter())
                ctx.sync(); t0 = time.perf_counter()
                xs_, obj, lamk, ti = prob.optimize(x0h, L.LFPSQPParams(do_project_retract=False, maxiter=iters, disp=L.DisplayOption.off, eps_kkt=0.0,
                                                                        eps_f=-1.0, eps_x=-1.0, callback=cb, callback_period=1))
                d = np.diff(np.array([t0] + stamps)) * 1e3
                if best is None or np.median(d[1:]) < np.median(best[1:]):
                    best = d
            out[tag] = {"ms_per_outer_iteration": float(np.median(best[1:])), "ms_first_iteration_with_setup": float(best[0]), "objective": float(obj[-1])}
        ctx.options.fused_tangent_step = True
        out["note"] = (f"steady outer iteration of optimize on the nonlinear class with streamed gradients at n = {n}, m = {m} (median of {iters - 1}): Gram pass + "
                       "tangent step (projection of d, lambda_kkt, Hessian term, projcg!'s initial projection: one pass) + truncated-Newton iterations + the "
                       "line search's Newton steps; statement_by_statement = src/optimize.jl:305-343 and src/projcg.jl:55-62 as five passes")
        for v_ in (x, target):
            v_.free()
        A.free()
        return out
    except Exception as e:                                      # (a side measurement must never take the bench line down)
        return {"error": repr(e)[:300]}


def gram_tile_share(m):
    """Share of the 16 x 16 output tiles of an m x m product that the Gram kernel computes (the upper block triangle of 128-column panels,
    inside a diagonal block its upper tile triangle)."""
    npan = (m + 127) // 128
    diag = npan * 36                 # 8 * 9 / 2 tiles of the 64 of a diagonal block
    off = npan * (npan - 1) // 2 * 64
    return (diag + off) / (npan * npan * 64.0)


def placements(ctx, L, n, m, n_loc, r0, R=3, its=12):
    """The fused kernel F on R further allocations of the basis, all alive at once in this process (FINDINGS.md §6: its time
    depends on where the 10 GB matrix landed and on the box): avg launch time of F over `its` iterations each."""
    import statistics
    keep, f_ms = [], []
    A = L.DiagOperator(0.0, ctx.vector(n_loc).hash_fill(3, r0, 4.5, 5.5))
    b = ctx.vector(n_loc).hash_fill(4, r0)
    x = ctx.vector(n_loc)
    work = L.ProjCGWork(ctx, n_loc, m)
    for r in range(R):
        Zr = ctx.matrix(n_loc, m)
        Zr.hash_fill(1, r0, n, 2.0 ** math.floor(math.log2(math.sqrt(3.0 / n))))
        keep.append(Zr)
        L.projcg_(x, None, A, L.DeviceBasis(Zr), b, None, tol=1e-300, maxit=2, work=work, n_global=n, want_lambda=False)
        ctx.set_profiling(True)
        L.projcg_(x, None, A, L.DeviceBasis(Zr), b, None, tol=1e-300, maxit=its, work=work, n_global=n, want_lambda=False)
        ms, cnt = ctx.profile_read()
        ctx.set_profiling(False)
        f_ms.append(ms[3] / cnt[3] if cnt[3] else float("nan"))
    for Zr in keep:
        Zr.free()
    return {"F_ms": f_ms, "min": min(f_ms), "median": statistics.median(f_ms), "max": max(f_ms), "allocations": R,
            "note": "scaled-hash bases, fresh allocations held simultaneously; same process and box as the timed run"}


def stream_rates(ctx, L, nbig=400_000_000):
    """What this box's HBM delivers to the library's own plain streaming kernels on 3.2 GB vectors (far beyond the 256 MB
    infinity cache): read-only (an 8-column GEMV-T: nine input streams, result stays on the device, no host sync), copy
    (1 read + 1 write), triad (waxpby: 2 reads + 1 write).  Context for the roofline fractions, which are quoted against
    8 TB/s; these are measurements of particular kernels, not ceilings."""
    a = ctx.vector(nbig).hash_fill(11)
    b = ctx.vector(nbig).hash_fill(12)
    c = ctx.vector(nbig)
    nrow8 = nbig // 8
    col = ctx.matrix(nrow8, 8).hash_fill(13, 0, nrow8)          # read probe: an 8-column GEMV-T (9 input streams of nrow8 doubles)
    a8 = ctx.vector(nrow8).hash_fill(14)
    t1 = ctx.vector(8)
    reps = 5
    L.gemv_t(col, a8, t1); c.copy_from(a); L.waxpby(1.0, a, 2.0, b, c)
    ctx.timer_begin()
    for _ in range(reps):
        L.gemv_t(col, a8, t1)
    ms_r = ctx.timer_end() / reps
    ctx.timer_begin()
    for _ in range(reps):
        c.copy_from(a)
    ms_c = ctx.timer_end() / reps
    ctx.timer_begin()
    for _ in range(reps):
        L.waxpby(1.0, a, 2.0, b, c)
    ms_t = ctx.timer_end() / reps
    g = lambda nb, ms: nb * 8.0 * nbig / (ms * 1e-3) / 1e9
    for v_ in (a, b, c, t1, a8):
        v_.free()
    col.free()
    return {"read_GBs": 9.0 * 8.0 * nrow8 / (ms_r * 1e-3) / 1e9, "copy_GBs": g(2, ms_c), "triad_GBs": g(3, ms_t), "vector_GB": 8.0 * nbig / 1e9}


def cpu_baseline(ns, m, n_full, side_scale=1.0):
    """The reference's CPU path as the oracle's C/OpenMP restatement (oracle/projcg_port.c), timed on this box's host cores
    (BASELINE.md 3): the (unfused) projcg! call sequence of src/projcg.jl:71-112, the two matvecs on their own (kgemv!, src/la_helper.jl:36-44),
    one Newton-retraction iteration (src/retractions.jl:133-165: a GEMV-N over U, c! = a GEMV-T over Jct, the m x m Broyden algebra) and a
    host triad.  At the FULL n when the host has the memory for it (the matrix alone is 8 n m bytes), else on a sample of ns rows with the
    rates rescaled by ns/n (every leg is linear in n); `n_measured` says which.  About 25 s of CPU work."""
    import numpy as np
    from oracle import port
    threads = port.usable_cpus()
    port.lib().port_set_num_threads(threads)
    avail = 0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
    except (OSError, ValueError):
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):      # a container's limit, if lower
        try:
            txt = open(path).read().strip()
            if txt != "max":
                used = 0
                try:
                    used = int(open(path.replace("memory.max", "memory.current").replace("limit_in_bytes", "usage_in_bytes")).read())
                except (OSError, ValueError):
                    pass
                avail = min(avail, max(0, int(txt) - used)) if avail else max(0, int(txt) - used)
        except (OSError, ValueError):
            pass
    need_full = 8.0 * n_full * m + 8.0 * n_full * 12 + (2 << 30)
    nm = n_full if (ns >= n_full or avail >= 1.5 * need_full) else ns        # measured size: full when it fits comfortably
    scale = 2.0 ** math.floor(math.log2(math.sqrt(3.0 / nm)))
    U = port.hash_matrix(1, nm, m, scale=scale)
    a = port.hash_vector(3, nm, 0, 4.5, 5.5)
    b = port.hash_vector(4, nm)
    port.projcg(a, U, b, None, 1e-300, 1)                     # touch everything once
    k = 2
    t0 = time.perf_counter()
    _, _, it, _ = port.projcg(a, U, b, None, 1e-300, k)
    dt = time.perf_counter() - t0
    k = max(3, min(900, int(12.0 * side_scale / (dt / k))))    # ~10 s of CPU work (the calibration iterations run cold, ~1.6 x slower)
    t0 = time.perf_counter()
    _, _, it, _ = port.projcg(a, U, b, None, 1e-300, k)
    dt = time.perf_counter() - t0
    fac = nm / n_full
    gb = lambda by, sec: by / sec / 1e9

    def timed(fn, budget=2.5):
        fn()
        t0 = time.perf_counter()
        fn()
        one = max(time.perf_counter() - t0, 1e-6)
        reps = max(2, min(200, int(budget * side_scale / one)))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps
    # the J matvecs on their own
    v, y, tm = port.hash_vector(5, nm), np.zeros(nm), port.hash_vector(6, m)
    s_t = timed(lambda: port.gemv_t(U, v))
    s_n = timed(lambda: port.gemv_n(U, tm, y, 1.0, 1.0))
    # one Newton-retraction iteration without bounds (retractions.jl:140-160): delta = -D c; xnew += U delta; c! = Jct' xnew - b; good Broyden.
    # U and Jct are two n x m matrices in the reference; the same array stands in for both here (identical traffic, half the host memory).
    D = np.asfortranarray(np.eye(m)); cval = port.hash_vector(8, m); bb = np.zeros(m)

    def nr_iter():
        dl = -(D @ cval)
        port.gemv_n(U, dl, y, 1.0, 1.0)
        c2 = port.gemv_t(U, y) - bb
        dc = c2 - cval
        t2 = D.T @ dl
        tv = dl - D @ dc
        den = float(t2 @ dc)
        if den != 0.0:
            D[:, :] += np.outer(tv, t2) / den
    s_nr = timed(nr_iter)
    # host triad on three vectors of 2e8 doubles (4.8 GB: beyond every cache level of the host; smaller when memory is short)
    nt = 200_000_000 if avail >= 20e9 else int(min(nm, 50_000_000))
    nt = max(100_000, int(nt * min(1.0, side_scale)))
    ta, tb, z = np.ones(nt), np.ones(nt), np.zeros(nt)
    s_tr = timed(lambda: port.triad(2.0, ta, tb, z), budget=1.0)
    by_t = 8.0 * nm * m + 8.0 * nm + 8.0 * m
    by_n = 8.0 * nm * m + 16.0 * nm + 8.0 * m
    return {"value": (it / dt) * fac, "unit": "iters/s", "cores": threads, "kind": "port", "n_measured": nm,
            "sample": (f"{it} iterations at n={nm}, m={m} (same generator)" +
                       (f", rate scaled by {nm}/{n_full}; measured {it / dt:.3f} it/s on the sample" if nm != n_full else " = the full size") +
                       "; unfused reference call sequence (src/projcg.jl:71-112), OpenMP on all loops"),
            "gemv_t": {"ms": s_t * 1e3 / fac, "GBs": gb(by_t, s_t), "note": "t = U'v (kgemv! 'T')" + ("" if nm == n_full else "; ms scaled to the full n")},
            "gemv_n": {"ms": s_n * 1e3 / fac, "GBs": gb(by_n, s_n), "note": "y = U t + y (kgemv! 'N')" + ("" if nm == n_full else "; ms scaled to the full n")},
            "nr_iteration": {"ms": s_nr * 1e3 / fac, "note": "one Newton-retraction iteration without bounds: GEMV-N over U, c! = GEMV-T over Jct, "
                                                             "m x m good-Broyden algebra (src/retractions.jl:140-160)" + ("" if nm == n_full else "; ms scaled to the full n")},
            "host_triad_GBs": gb(24.0 * nt, s_tr), "host_triad_doubles_per_vector": nt, "host_mem_available_GB": avail / 1e9}


if __name__ == "__main__":
    main()
