"""Placement trials for the buffers of the projected-CG loop -- the round-2 HARNESS, kept for comparison (bench.py --placement grid).
The product's policy lives in the library: lfpsqp_basis_work_alloc_placed / lfpsqp_mat_alloc_placed / lfpsqp_vecs_alloc_placed
(csrc/context.hip), used by optimize_core and by bench.py's default.

On MI355X the kernels that run a store stream inside a matrix read stream (the fused projected-CG iteration, GEMV-N, the Newton step)
are 10-15 % faster or slower depending on where the matrix and the n-vectors were allocated -- a property of the PAIR of allocations,
reproducible inside a process, different from one hour to the next.  ``best_projcg_buffers`` allocates a few candidates
of each, times the fused kernel on every pair for a dozen iterations and keeps the fastest pair.  Results do not depend on the choice."""
from __future__ import annotations

from .device import Context
from .projcg import DeviceBasis, ProjCGWork, projcg_


def best_projcg_buffers(ctx: Context, make_basis, n_loc: int, m: int, A, b, *, n_global: int | None = None, nbasis: int = 3, nwork: int = 4,
                        iters: int = 12, try_alternating: bool = False, extend: tuple | None = None):
    """``make_basis()`` -> (DeviceMatrix, description): called ``nbasis`` times (same contents, new allocation each time).
    Returns (Z, description, x, work, info) with info = {"grid": F ms per (basis, work set), "basis": i, "work": j, "residual_buffers": 0 | 1}.
    ``try_alternating``: also time the alternating residual buffers on the chosen pair and leave the context in the faster scheme
    (single-rank use: the decision is taken from local wall times).
    ``extend = (target_ms, max_extra)``: while the best pair's F is above ``target_ms``, up to ``max_extra`` further basis allocations are tried
    against every work set, one at a time (the loser of each comparison is freed at once).  Single-rank use as well: how many extra trials run
    is decided from local times.
    Every rank makes the same (collective-carrying) calls when neither option is used; the choice itself is local."""
    n_global = n_loc if n_global is None else n_global
    nbasis, nwork = max(1, int(nbasis)), max(1, int(nwork))
    # the candidates of the basis live side by side (plus one scratch copy while make_basis orthonormalises): no more of them than fit in
    # three quarters of the free device memory (round-2 advisor finding: 3 x 164 GB at n = 4e7, m = 512 would not)
    free_b = ctx.free_memory()
    if free_b is not None and n_loc * m > 0:
        nbasis = max(1, min(nbasis, int(0.75 * free_b / (8.0 * n_loc * m)) - 1))

    def trial(Uk, xk, wk):
        projcg_(xk, None, A, Uk, b, None, tol=1e-300, maxit=2, work=wk, n_global=n_global, want_lambda=False)      # touch
        ctx.set_profiling(True)
        projcg_(xk, None, A, Uk, b, None, tol=1e-300, maxit=iters, work=wk, n_global=n_global, want_lambda=False)
        pms, pcnt = ctx.profile_read()
        ctx.set_profiling(False)
        slot = 3 if pcnt[3] > 0 else 2                  # fused kernel F, else the second pass of the two-pass iteration
        return pms[slot] / pcnt[slot] if pcnt[slot] else float("inf")

    cands, pads = [], []
    for k in range(nwork):
        cands.append((ctx.vector(n_loc), ProjCGWork(ctx, n_loc, m)))
        if k + 1 < nwork:
            pads.append(ctx.vector(1_000_003 * (k + 1)))                   # shifts where the next set lands
    bases = []
    for k in range(nbasis):
        bases.append(make_basis())
        if k + 1 < nbasis:
            pads.append(ctx.vector(3_000_017 * (k + 1)))
    grid = [[trial(DeviceBasis(Zk), xk, wk) for xk, wk in cands] for Zk, _ in bases] if nbasis * nwork > 1 else [[0.0]]
    bi, wi = min(((i, j) for i in range(nbasis) for j in range(nwork)), key=lambda ij: grid[ij[0]][ij[1]])
    Z, desc = bases[bi]
    for k, (Zk, _) in enumerate(bases):
        if k != bi:
            Zk.free()
    info = {"grid": grid, "basis": bi, "work": wi, "residual_buffers": 0}
    if extend is not None and nbasis * nwork > 1:
        target_ms, max_extra = float(extend[0]), int(extend[1])
        best = grid[bi][wi]
        extra = 0
        while best > target_ms and extra < max_extra:
            pads.append(ctx.vector(3_000_017 * (nbasis + extra)))
            Zn, descn = make_basis()
            row = [trial(DeviceBasis(Zn), xk, wk) for xk, wk in cands]
            grid.append(row)
            extra += 1
            j = min(range(nwork), key=lambda jj: row[jj])
            if row[j] < best:
                Z.free()
                Z, desc, best, wi = Zn, descn, row[j], j
                info.update(basis=nbasis + extra - 1, work=j)
            else:
                Zn.free()
        info["extra_basis_trials"] = extra
    x, work = cands[wi]
    for k, (xk, wk) in enumerate(cands):            # the losing work sets are released now, not whenever the garbage collector gets to them
        if k != wi:
            for v_ in (xk, wk.g, wk.d, wk.rp, wk.Utr):
                v_.free()
    for v_ in pads:
        v_.free()
    del pads
    if try_alternating and nbasis * nwork > 1:
        # the alternating residual buffers (lfpsqp_ctx_set_residual_buffers) on the chosen pair: ~12 % faster when even the best pair is a
        # slow one, 2-4 % slower otherwise.  Compared by the wall time of whole calls (the sampled kernel time would alias with the
        # period-2 alternation).
        import time
        U = DeviceBasis(Z)

        def wall(mode):
            ctx.set_residual_buffers(mode)
            projcg_(x, None, A, U, b, None, tol=1e-300, maxit=2, work=work, n_global=n_global, want_lambda=False)
            best = float("inf")
            for _ in range(2):
                ctx.sync(); t0 = time.perf_counter()
                projcg_(x, None, A, U, b, None, tol=1e-300, maxit=iters + 4, work=work, n_global=n_global, want_lambda=False)
                ctx.sync(); best = min(best, time.perf_counter() - t0)
            return best * 1e3 / (iters + 4)
        w0, w1 = wall(0), wall(1)
        info["call_ms_per_iteration"] = {"in_place": w0, "alternating": w1}
        info["residual_buffers"] = 1 if w1 < 0.98 * w0 else 0
        ctx.set_residual_buffers(info["residual_buffers"])
    return Z, desc, x, work, info
