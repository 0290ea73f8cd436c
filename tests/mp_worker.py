"""Worker of tests/test_multirank_gloo.py: rank `r` of `w` processes, emulator build of the C-ABI
library, all-reduce callback over torch.distributed/gloo.  Writes its shard results to an .npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import lfpsqp_jl_amd as L  # noqa: E402
from lfpsqp_jl_amd.distributed import torch_allreduce_callback  # noqa: E402
from oracle import synth  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: no host-name lookups for the pair connections
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lib = L.load_library(os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so"))
    ctx = L.Context(0, lib)
    ctx.comm_init_callback(rank, world, torch_allreduce_callback(None))
    res = {}
    # ---- sharded projcg on the bench workload -------------------------------------------------
    n, m = 3200, 6
    r0, r1 = ctx.shard_range(n)
    nl = r1 - r0
    J = ctx.matrix(nl, m).hash_fill(1, r0, n)
    Z = ctx.matrix(nl, m)
    S, Vt, rank_ = L.ksvd_(J, Z)
    A = L.DiagOperator(0.0, ctx.vector(nl).hash_fill(3, r0, 4.0, 5.0))
    b = ctx.vector(nl).hash_fill(4, r0)
    x, lam = ctx.vector(nl), ctx.vector(m)
    it, nr = L.projcg_(x, lam, A, L.DeviceBasis(Z), b, None, tol=1e-10, maxit=500, n_global=n)
    res.update(r0=r0, r1=r1, S=S, x=x.download(), lam=lam.download(), it=it, nr=nr, Z=Z.download(), Vt=Vt)
    # ---- a rank with ZERO rows (n smaller than one shard granule): collectives must still line up ----------
    ne, me = 1500, 4
    e0, e1 = ctx.shard_range(ne)
    Je = ctx.matrix(e1 - e0, me).hash_fill(1, e0, ne)
    Ze = ctx.matrix(e1 - e0, me)
    Se, Vte, rke = L.ksvd_(Je, Ze)
    xe, lame = ctx.vector(e1 - e0), ctx.vector(me)
    ite, nre = L.projcg_(xe, lame, L.DiagOperator(0.0, ctx.vector(e1 - e0).hash_fill(3, e0, 4.0, 5.0)), L.DeviceBasis(Ze),
                         ctx.vector(e1 - e0).hash_fill(4, e0), None, tol=1e-10, maxit=300, n_global=ne)
    res.update(e0=e0, e1=e1, e_x=xe.download(), e_it=ite, e_S=Se, e_lam=lame.download())
    # ... and a sparse block on the same sharding: the exact Gram accumulation's collectives (agreement, bound, sum) with an empty shard
    ke = 2
    rows_e = np.repeat(np.arange(ne), ke)
    cols_e = (((np.arange(ne) * me) // ne)[:, None] + np.arange(ke)[None, :]) % me
    vals_e = (np.random.default_rng(4).standard_normal((ne, ke)) + 2.0 * (np.arange(ke) == 0)).ravel()
    sel_e = (rows_e >= e0) & (rows_e < e1)
    Se_sp = L.SparseMatrix(ctx, e1 - e0, me, rows_e[sel_e] - e0, cols_e.ravel()[sel_e], vals_e[sel_e])
    we = ctx.vector(e1 - e0).hash_fill(7, e0, 0.4, 0.6)
    res.update(e_gram=Se_sp.gram(w2=we), e_gram_plain=Se_sp.gram())
    # ---- sharded config 3 through the outer driver (NR and ProjPenalty) ----------------------
    n3, m3 = 2800, 5
    q0, q1 = ctx.shard_range(n3)
    Jct = ctx.matrix(q1 - q0, m3).hash_fill(1, q0, n3)
    xs = ctx.vector(q1 - q0).hash_fill(2, q0)
    bb = ctx.vector(m3)
    L.gemv_t(Jct, xs, bb)
    for tag, dpr in (("nr", False), ("pp", True)):
        P = L.QuadLinearBallBox(ctx, q1 - q0, m3, Jct, bb.download(), n_global=n3)
        xo, obj, lamk, ti = P.optimize(np.ones(q1 - q0), L.LFPSQPParams(do_project_retract=dpr, disp=L.DisplayOption.off))
        res.update({f"c3{tag}_x": xo, f"c3{tag}_obj": obj, f"c3{tag}_lam": lamk, f"c3{tag}_iter": ti.iter, "q0": q0, "q1": q1})
    # ---- sharded config 4: the slack variable lives on the LAST rank --------------------------
    n4, m4 = 2600, 4
    s0, s1 = ctx.shard_range(n4)
    last = rank == world - 1
    p_loc = 1 if last else 0
    J4 = ctx.matrix(s1 - s0 + p_loc, m4 + 1).hash_fill(1, s0, n4, 1.0, s1 - s0, m4)
    xs4 = ctx.vector(s1 - s0 + p_loc).hash_fill(2, s0)
    if last:
        ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs4.h, s1 - s0, 1, 0.0))
    b4 = ctx.vector(m4 + 1)
    L.gemv_t(J4, xs4, b4, ncols=m4)
    i = np.arange(s0, s1)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf)
    xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    x0 = 0.97 * synth.hash_vector(2, n4)[s0:s1] + 0.03 * 0.5
    P4 = L.QuadLinearBallBox(ctx, s1 - s0, m4, J4, b4.download()[:m4], R2=n4 / 2.0, xl=xl, xu=xu, n_global=n4, owns_slack=last)
    xo, obj, lamk, ti = P4.optimize(x0, L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=3))
    res.update(c4_x=xo, c4_obj=obj, c4_lam=lamk, c4_iter=ti.iter, s0=s0, s1=s1)
    # the same problem from a start whose first linesearches FAIL repeatedly: the batched trial retractions (ls_batch) and
    # their collectives run sharded
    tr = []
    xo, obj, lamk, ti = P4.optimize(0.5 * np.ones(s1 - s0), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=2,
                                                                           maxiter_retract=25), trace=tr)
    res.update(c4b_x=xo, c4b_obj=obj, c4b_iter=ti.iter, c4b_r1=np.array([t.get("retract_iter1") or 0 for t in tr]),
               c4b_alpha=np.array([t.get("alpha") or 0.0 for t in tr]))
    # ... and one by one (ls_batch = 1): the default's EXACT batch must be the one-by-one search bit for bit on every rank -- the trials'
    # sums are all-reduced element by element in a fixed rank order, batched or not
    ctx.options.ls_batch = 1
    tr1 = []
    xo1, obj1, lamk1, ti1 = P4.optimize(0.5 * np.ones(s1 - s0), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=2,
                                                                               maxiter_retract=25), trace=tr1)
    ctx.options.ls_batch = 0
    same = (ti1.iter == ti.iter and len(tr1) == len(tr) and np.array_equal(xo1, xo) and np.array_equal(obj1, obj)
            and all(np.array_equal(a_["x"], b_["x"]) and a_.get("retract_iter1") == b_.get("retract_iter1") and a_.get("alpha") == b_.get("alpha")
                    for a_, b_ in zip(tr1, tr)))
    res.update(c4b_exact_batch_is_one_by_one=bool(same), c4b_failed_retractions=int(sum(1 for t in tr if (t.get("retract_iter1") or 0) >= 25)))
    # ---- round-2 paths, sharded: refinement rounds of the factorisation (ill-conditioned block; the small Jacobi runs replicated),
    #      a general operator behind the callback (lfpsqp_projcg_op), sparse equalities through the default retraction --------------
    ni, mi = 3000, 6
    rng = np.random.default_rng(31)
    Q1, _ = np.linalg.qr(rng.standard_normal((ni, mi)))
    Q2, _ = np.linalg.qr(rng.standard_normal((mi, mi)))
    Jill = np.asfortranarray((Q1 * np.logspace(0, -8, mi)) @ Q2.T)
    i0, i1 = ctx.shard_range(ni)
    Ji, Zi = ctx.matrix(i1 - i0, mi, Jill[i0:i1]), ctx.matrix(i1 - i0, mi)
    Si, Vti, rki = L.ksvd_(Ji, Zi)
    ai = ctx.vector(i1 - i0).hash_fill(3, i0, 4.0, 5.0)
    ei_full = 0.8 * synth.hash_vector(15, ni)[:ni - 1]
    # tridiagonal operator restricted to this rank's rows (couplings across the shard boundary dropped on both ranks: still symmetric)
    e_loc = ei_full[i0:i1 - 1] if i1 - i0 > 0 else np.zeros(0)

    class Tri:
        def __init__(self):
            nl_ = i1 - i0
            self.up = ctx.vector(nl_, np.concatenate([e_loc, [0.0]]) if nl_ else None)
            self.dn = ctx.vector(nl_, np.concatenate([[0.0], e_loc]) if nl_ else None)
            self.sh, self.t = ctx.vector(nl_), ctx.vector(nl_)

        def mul_(self, dest, v, al=None, be=None):
            nl_ = i1 - i0
            L.vmul(ai, v, dest)
            if nl_ > 1:
                self.sh.fill(0.0); self.sh.copy_range_from(v, nl_ - 1, 0, 1)
                L.vmul(self.up, self.sh, self.t); L.axpby(1.0, self.t, 1.0, dest)
                self.sh.fill(0.0); self.sh.copy_range_from(v, nl_ - 1, 1, 0)
                L.vmul(self.dn, self.sh, self.t); L.axpby(1.0, self.t, 1.0, dest)
            return dest

        def adjoint(self):
            return self
    xi, lami = ctx.vector(i1 - i0), ctx.vector(mi)
    iti, nri = L.projcg_(xi, lami, Tri(), L.DeviceBasis(Zi, rki), ctx.vector(i1 - i0).hash_fill(4, i0), None, tol=1e-10, maxit=400, n_global=ni)
    res.update(i0=i0, i1=i1, ill_S=Si, ill_rank=rki, ill_Z=Zi.download(), op_x=xi.download(), op_it=iti, op_lam=lami.download())
    nsp, msp, ksp = 3000, 8, 3
    g0, g1 = ctx.shard_range(nsp)
    rows = np.repeat(np.arange(nsp), ksp)
    cols = (((np.arange(nsp) * msp) // nsp)[:, None] + np.arange(ksp)[None, :]) % msp
    vals = (np.random.default_rng(9).standard_normal((nsp, ksp)) + 2.0 * (np.arange(ksp) == 0)).ravel()
    sel = (rows >= g0) & (rows < g1)
    Ssp = L.SparseMatrix(ctx, g1 - g0, msp, rows[sel] - g0, cols.ravel()[sel], vals[sel])
    Jsp_dense = Ssp.to_dense()
    xss = ctx.vector(g1 - g0).hash_fill(2, g0)
    bsp = ctx.vector(msp)
    L.spmv_t(Ssp, xss, bsp)
    Psp = L.QuadLinearBallBox(ctx, g1 - g0, msp, Jsp_dense, bsp.download(), n_global=nsp, Jsp=Ssp)
    x0sp = xss.download() + 0.05 * synth.hash_vector(6, nsp)[g0:g1]
    xo, obj, lamk, ti = Psp.optimize(x0sp, L.LFPSQPParams(disp=L.DisplayOption.off, maxiter=3))
    res.update(g0=g0, g1=g1, sp_x=xo, sp_obj=obj, sp_iter=ti.iter, sp_b=bsp.download())
    # ... and with the Newton retraction: tangent setup, projected CG and every Newton step on the nonzeros of the shard
    xo, obj, lamk, ti = Psp.optimize(x0sp, L.LFPSQPParams(disp=L.DisplayOption.off, maxiter=3, do_project_retract=False))
    res.update(spn_x=xo, spn_obj=obj, spn_iter=ti.iter)
    # the Gram matrix of the sharded sparse block from its nonzeros, weighted (the bound on the terms is a max-all-reduce, the sum of the
    # ranks' exactly accumulated partial matrices a floating-point all-reduce)
    wsp = ctx.vector(g1 - g0).hash_fill(7, g0, 0.4, 0.6)
    res.update(sp_gram=Ssp.gram(w2=wsp), sp_gram_plain=Ssp.gram())
    # ---- round 3, sharded: the nonlinear (elementwise) constraint class -- dense A with mixed kinds + the common quadratic term (its sum
    #      of squares is a collective), Newton retraction and `optimize`; and the sin system on the nonzeros of the shard ------------------
    from tests.test_elementwise import ew_test_data
    ne_, me_ = 2600, 6
    h0, h1 = ctx.shard_range(ne_)
    Ar, kr, qr, br, target, x0e = ew_test_data(ne_, me_)
    cons = L.ElementwiseConstraints(ctx, ctx.matrix(h1 - h0, me_, np.asfortranarray(Ar[h0:h1])), br, kind=kr[h0:h1], qw=qr)
    prob = L.SeparableElementwiseBox(ctx, cons, 0, 1.0, target[h0:h1], n_global=ne_)
    tr = []
    xo, obj, lamk, ti = prob.optimize(x0e[h0:h1], L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=5), trace=tr)
    res.update(h0=h0, h1=h1, ew_x=xo, ew_obj=obj, ew_iter=ti.iter, ew_lam=lamk, ew_r1=np.array([t.get("retract_iter1") or 0 for t in tr]))
    ns_, ms_ = 2600, 40
    k0, k1 = ctx.shard_range(ns_)
    i_ = np.arange(ms_)
    rows_s = np.concatenate([2 * i_ + 1, 2 * i_]); cols_s = np.concatenate([i_, i_]); vals_s = np.concatenate([np.ones(ms_), -np.ones(ms_)])
    sel = (rows_s >= k0) & (rows_s < k1)
    As = L.SparseMatrix(ctx, k1 - k0, ms_, rows_s[sel] - k0, cols_s[sel], vals_s[sel])
    kind_s = np.zeros(ns_); kind_s[0:2 * ms_:2] = 1
    cons_s = L.ElementwiseConstraints(ctx, As, np.zeros(ms_), kind=kind_s[k0:k1])
    targ_s = 0.5 * synth.hash_vector(21, ns_)
    prob_s = L.SeparableElementwiseBox(ctx, cons_s, 0, 1.0, targ_s[k0:k1], n_global=ns_)
    xo, obj, lamk, ti = prob_s.optimize(np.zeros(k1 - k0), L.LFPSQPParams(do_project_retract=False, disp=L.DisplayOption.off, maxiter=5))
    res.update(k0=k0, k1=k1, sin_x=xo, sin_obj=obj, sin_iter=ti.iter)
    np.savez(out, **res)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
