"""Worker of tests/test_p2p_transport.py: rank `r` of `w` processes with the library's one-shot peer-to-peer all-reduce (handles exchanged
through files) or, for comparison, the gloo callback transport.  argv: rank world dir transport lib n m [device]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import lfpsqp_jl_amd as L  # noqa: E402


def main():
    rank, world, d, transport, libpath = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    n, m = int(sys.argv[6]), int(sys.argv[7])
    lib = L.load_library(libpath) if libpath != "default" else None
    if transport != "p2p":                      # (torch first, as bench.py does: it must find the GPU before the library's context exists)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", init_method=f"file://{d}/gloo_rdv", rank=rank, world_size=world)
        if libpath == "default":
            torch.cuda.set_device(0)
    ctx = L.Context(0, lib)
    if transport == "p2p":
        h = ctx.comm_p2p_export()
        tmp = os.path.join(d, f"h{rank}.tmp")
        open(tmp, "wb").write(h)
        os.rename(tmp, os.path.join(d, f"h{rank}.bin"))
        hs = []
        t0 = time.time()
        for r in range(world):
            f = os.path.join(d, f"h{r}.bin")
            while not os.path.exists(f):
                assert time.time() - t0 < 120, "peer handle did not appear"
                time.sleep(0.01)
            hs.append(open(f, "rb").read())
        ctx.comm_init_p2p(rank, world, hs)
    else:
        from lfpsqp_jl_amd.distributed import host_staged_allreduce_callback, torch_allreduce_callback
        emulated = "emulator" in ctx.device_name
        ctx.comm_init_callback(rank, world, torch_allreduce_callback(None) if emulated else host_staged_allreduce_callback(0))
    res = {}
    r0, r1 = ctx.shard_range(n)
    nl = r1 - r0
    # 1. plain collectives: a short vector, one longer than a mailbox slot (pieces), max-reduction through amax
    v = ctx.vector(9000).hash_fill(7 + rank)
    ctx.check(ctx.L.lfpsqp_allreduce(ctx.h, v.h, 9000))
    res["sum9000"] = v.download()
    a = ctx.vector(nl).hash_fill(11, r0)
    res["dot"], res["amax"] = L.dot(a, a), L.amax(a)
    # 2. tangent setup + projected CG, row-sharded (Gram all-reduce of m*m doubles, 2m + 5 doubles per CG iteration)
    J = ctx.matrix(nl, m).hash_fill(1, r0, n)
    Z = ctx.matrix(nl, m)
    S, Vt, rk = L.ksvd_(J, Z)
    A = L.DiagOperator(0.0, ctx.vector(nl).hash_fill(3, r0, 4.0, 5.0))
    b = ctx.vector(nl).hash_fill(4, r0)
    x, lam = ctx.vector(nl), ctx.vector(m)
    t0 = time.perf_counter()
    it, nr = L.projcg_(x, lam, A, L.DeviceBasis(Z), b, None, tol=1e-10, maxit=400, n_global=n)
    ctx.sync()
    res.update(S=S, it=it, nr=nr, x=x.download(), lam=lam.download(), r0=r0, r1=r1, projcg_s=time.perf_counter() - t0)
    # 3. latency of one small collective: K all-reduces of 261 doubles (2m + 5 at m = 128) back to back
    w = ctx.vector(261).hash_fill(5)
    ctx.sync()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.check(ctx.L.lfpsqp_allreduce(ctx.h, w.h, 261))
    ctx.sync()
    res["us_per_allreduce"] = (time.perf_counter() - t0) / K * 1e6
    np.savez(os.path.join(d, f"out_{transport}_{rank}.npz"), **res)
    ctx.close()
    if transport != "p2p":          # leave the group in order: a process that exits with gloo's threads alive can abort in their destructors
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
