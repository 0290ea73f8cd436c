"""Worker of tests/test_many_ranks.py: rank `r` of `w` emulator processes, transport "p2p" (the library's one-shot all-reduce, mailboxes over
POSIX shared memory, handles exchanged through files) or "gloo" (callback transport).  Row-sharded: plain collectives, the sustained
projected-CG workload of bench.py, config 3 (both retractions) and config 4 (slack variable on the LAST rank, which may own no other row).
argv: rank world dir transport n m"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import lfpsqp_jl_amd as L  # noqa: E402
from oracle import synth  # noqa: E402


def exchange(d, tag, rank, world, blob):
    tmp = os.path.join(d, f"{tag}{rank}.tmp")
    open(tmp, "wb").write(blob)
    os.rename(tmp, os.path.join(d, f"{tag}{rank}.bin"))
    out, t0 = [], time.time()
    for r in range(world):
        f = os.path.join(d, f"{tag}{r}.bin")
        while not os.path.exists(f):
            assert time.time() - t0 < 300, "peer handle did not appear"
            time.sleep(0.01)
        out.append(open(f, "rb").read())
    return out


def main():
    rank, world, d, transport = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    n, m = int(sys.argv[5]), int(sys.argv[6])
    lib = L.load_library(os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so"))
    ctx = L.Context(0, lib)
    if transport == "p2p":
        hs = exchange(d, "h", rank, world, ctx.comm_p2p_export())
        ctx.comm_init_p2p(rank, world, hs)
        kind, _ = ctx.comm_p2p_info()
        assert kind == "fine-grained", kind
    else:
        import torch.distributed as dist
        from lfpsqp_jl_amd.distributed import torch_allreduce_callback
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", init_method=f"file://{d}/gloo_rdv", rank=rank, world_size=world)
        ctx.comm_init_callback(rank, world, torch_allreduce_callback(None))
    res = {}
    r0, r1 = ctx.shard_range(n)
    nl = r1 - r0
    res.update(r0=r0, r1=r1)
    # 1. plain collectives: sum in FIXED rank order (p2p), max, a payload longer than one mailbox slot (pieces)
    v = ctx.vector(9000).hash_fill(7 + rank)
    ctx.check(ctx.L.lfpsqp_allreduce(ctx.h, v.h, 9000))
    res["sum9000"] = v.download()
    a = ctx.vector(nl).hash_fill(11, r0)
    res["dot"], res["amax"] = L.dot(a, a), L.amax(a)
    # 2. the bench's workload: tangent setup + sustained projected CG (one all-reduce of 2m + 5 doubles per iteration)
    J = ctx.matrix(nl, m).hash_fill(1, r0, n)
    Z = ctx.matrix(nl, m)
    S, Vt, rk = L.ksvd_(J, Z)
    A = L.DiagOperator(0.0, ctx.vector(nl).hash_fill(3, r0, 4.0, 5.0))
    b = ctx.vector(nl).hash_fill(4, r0)
    x, lam = ctx.vector(nl), ctx.vector(m)
    it, nr = L.projcg_(x, lam, A, L.DeviceBasis(Z), b, None, tol=1e-10, maxit=400, n_global=n)
    res.update(S=S, Vt=Vt, it=it, nr=nr, x=x.download(), lam=lam.download(), Z=Z.download())
    # 3. config 3 through the outer driver, Newton and ProjPenalty retractions
    Jct = ctx.matrix(nl, m).hash_fill(1, r0, n)
    xs = ctx.vector(nl).hash_fill(2, r0)
    bb = ctx.vector(m)
    L.gemv_t(Jct, xs, bb)
    for tag, dpr in (("nr", False), ("pp", True)):
        P = L.QuadLinearBallBox(ctx, nl, m, Jct, bb.download(), n_global=n)
        tr = []
        xo, obj, lamk, ti = P.optimize(np.ones(nl), L.LFPSQPParams(do_project_retract=dpr, disp=L.DisplayOption.off), trace=tr)
        res.update({f"c3{tag}_x": xo, f"c3{tag}_obj": obj, f"c3{tag}_lam": lamk, f"c3{tag}_iter": ti.iter})
    # 4. config 4: ball (slack variable on the last rank) + four-way bounds, Newton retraction, from a start near the feasible set
    last = rank == world - 1
    p_loc = 1 if last else 0
    J4 = ctx.matrix(nl + p_loc, m + 1).hash_fill(1, r0, n, 1.0, nl, m)
    xs4 = ctx.vector(nl + p_loc).hash_fill(2, r0)
    if last:
        ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs4.h, nl, 1, 0.0))
    b4 = ctx.vector(m + 1)
    L.gemv_t(J4, xs4, b4, ncols=m)
    i = np.arange(r0, r1)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    x0 = 0.97 * synth.hash_vector(2, n)[r0:r1] + 0.03 * 0.5
    P4 = L.QuadLinearBallBox(ctx, nl, m, J4, b4.download()[:m], R2=n / 2.0, xl=xl, xu=xu, n_global=n, owns_slack=last)
    tr = []
    xo, obj, lamk, ti = P4.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3), trace=tr)
    res.update(c4_x=xo, c4_obj=obj, c4_lam=lamk, c4_iter=ti.iter, c4_r1=np.array([t.get("retract_iter1") or 0 for t in tr]),
               c4_alpha=np.array([t.get("alpha") or 0.0 for t in tr]))
    if transport == "p2p":
        res["p2p_collectives"] = ctx.comm_p2p_info()[1]
    np.savez(os.path.join(d, f"out_{transport}_{rank}.npz"), **res)
    # leave together: a rank that unmaps its mailbox while a peer is still reading it would fault the peer
    exchange(d, f"done_{transport}_", rank, world, b"x")
    ctx.close()
    if transport != "p2p":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
