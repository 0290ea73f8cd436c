"""The outer LFPSQP driver on device-resident state (reference src/optimize.jl:119-443).

``optimize(f, grad_, c_, jac_, hess_lag_vec_, x0, xl, xu, m, param, ctx=...)`` keeps the
reference's explicit-derivative signature and return value
``(x, obj_values, lam_kkt, TerminationInfo)``; only scalars and m-vectors cross PCIe per
outer iteration.  Callback contract (the device analogue of the reference's):

    f(x)                      -> float; x is the device iterate (stacked [x; y] when bounds exist,
                                 f must only look at the first n entries)
    grad_(g, x)               writes the first n entries of the device vector g
    c_                        DeviceConstraints, or a host callable c_(cval, x_host)
    jac_(Jct, cval, x)        refreshes the device n x m matrix Jct (= Jc' of the reference) and cval
    hess_lag_vec_             either an object with ``diag_(hx, x, lam)`` filling the N-vector hx with
                              the diagonal of the Lagrangian Hessian (fused projcg path), or a callable
                              hess_lag_vec_(dest, src, x, lam) on device vectors (generic path)

Deviations from the reference, all invisible in the iterates up to rounding: the thin SVD is the
Gram-based factorisation of lfpsqp_factorize (basis rotation / sign freedom; the rank is the reference's
``sigma >= eps_rank`` on singular values of dgesvd's accuracy, so the Newton / ProjPenalty choice is the
reference's also for ill-conditioned Jacobians), ``Jct`` is not destroyed by it, and the 2N x M factor with
bounds is never materialised.  Where the reference itself fails -- a rank-deficient block without bounds ends
in a DimensionMismatch at src/projcg.jl:118 -- this driver carries on with ProjPenalty."""
from __future__ import annotations

import math

import numpy as np

from .device import Context, DeviceMatrix, DeviceVector, amax, axpby, dot, gemv_n, gemv_t, nrm2, waxpby
from .factorize import ksvd_
from .inequality import (InequalityData, InequalityDecomp, InequalityDecompProject, StackedVector, generate_initial_y_,
                         inequality_gradient_)
from .linesearch import ArmijoWork, ExactLinesearchWork, armijo_, exact_linesearch_
from .params import DisplayOption, LFPSQPParams, LinesearchOption, TerminationCondition, TerminationInfo
from .projcg import DeviceBasis, DiagOperator, ProjCGWork, projcg_
from .retractions import NR, Euclidean, NRWork, YRetract


def _print_iter_header():
    print("   step |          f     ||c||      |Δf|    ||Δx||  |   S iter      res  |   M   iter  (pcg)  |        α  flag")
    print("-" * 110)


def _print_first_line(fval, normc):
    print("      0 | %10.3e  %8.1e                      |                    |                    |               " % (fval, normc))


def _print_iter(i, fval, normc, fstep, normx, steptype, tn_iter, tn_res, methodtype, iter1, iter2, alpha, flag):
    print("%7d | %10.3e  %8.1e  %8.1e  %8.1e  |  %s %4d %8.1e  |  %s %6d %6d  | %8.1e  %4d" %
          (i, fval, normc, fstep, normx, "GD" if steptype == 0 else "TN", tn_iter, tn_res, "NR" if methodtype == 0 else "PP",
           iter1, iter2, alpha, flag), flush=True)


class _GenericHessian:
    """LinearMap((dest, src) -> hess_lag_vec!(dest, src, x, lam)) (src/optimize.jl:230) for callables."""

    def __init__(self, fn, x, lam):
        self.fn, self.x, self.lam = fn, x, lam
        self._tmp = None

    def mul_(self, dest, v, a=None, b=None):
        if a is None:
            self.fn(dest, v, self.x, self.lam)
        else:
            if self._tmp is None:
                self._tmp = DeviceVector(dest.ctx, dest.n)
            self.fn(self._tmp, v, self.x, self.lam)
            waxpby(a, self._tmp, b, dest, dest)
        return dest

    def adjoint(self):
        return self


class _GenericAugHessian:
    """LinearMap((dest, src) -> augmented_hess_lag_vec!(dest, src, hess_lag_vec!, x, lam, lamy, idata))
    (src/optimize.jl:228, src/inequality_helper.jl:144-158) for a general Hessian callable with bounds:
    dest_x = H src_x + 2 lamy.*q.*src_x ; dest_y = 2 lamy.*s.*src_y.  Fallback path (extra vector passes)."""

    def __init__(self, fn, x, lam_dev, lamy, idata, n):
        self.fn, self.x, self.lam, self.lamy, self.idata, self.n = fn, x, lam_dev, lamy, idata, n
        ctx = x.ctx
        self.zero_hx = DeviceVector(ctx, n)
        self.diag = StackedVector(ctx, n)
        self.src_x, self.x_x, self.h_x = DeviceVector(ctx, n), DeviceVector(ctx, n), DeviceVector(ctx, n)
        self.hfull = StackedVector(ctx, n)
        self.tmp = StackedVector(ctx, n)

    def _apply(self, dest, v):
        from .device import vmul
        from .inequality import augmented_hess_diag_
        n = self.n
        augmented_hess_diag_(self.diag, self.zero_hx, self.lamy, self.idata)       # [2 lamy q ; 2 lamy s]
        self.src_x.copy_range_from(v, n)
        self.x_x.copy_range_from(self.x, n)
        self.fn(self.h_x, self.src_x, self.x_x, self.lam)                          # H src_x
        self.hfull.copy_range_from(self.h_x, n)                                    # [H src_x ; 0]
        vmul(self.diag, v, dest)
        axpby(1.0, self.hfull, 1.0, dest)

    def mul_(self, dest, v, a=None, b=None):
        if a is None:
            self._apply(dest, v)
        else:
            self._apply(self.tmp, v)
            waxpby(a, self.tmp, b, dest, dest)
        return dest

    def adjoint(self):
        return self


def _amax_host(v):
    return float(np.max(np.abs(v), initial=0.0))


def _allocate_projcg_work(ctx, n, m, ineq, Jct, factored, diagonal_hessian):
    """ProjCGWork (src/optimize.jl:214) and, unless the basis stays in factored form, the basis Z (:191) -- allocated TOGETHER and by trial
    (lfpsqp_vecs_alloc_placed / lfpsqp_basis_work_alloc_placed: the speed of the fused kernel is a property of the pair of allocations)."""
    if factored:
        projcgwork = ProjCGWork(ctx, n, m, n if ineq else None, against=Jct, extra=1)
        idecomp = InequalityDecomp(ctx, n, m, Jct, factored=True)
    else:
        projcgwork = ProjCGWork(ctx, n, m, n if ineq else None, against=(("new", n, m) if m > 0 else None), extra=1 if diagonal_hessian else 0)
        idecomp = InequalityDecomp(ctx, n, m, Jct, Z=projcgwork.basis)
    return projcgwork, idecomp


def optimize_core(f, grad_, c_, jac_, hess_lag_vec_, x0, xl, xu, m: int, param: LFPSQPParams | None = None, *, ctx: Context,
                  n_global: int | None = None, trace=None):
    """src/optimize.jl:119-443.  x0 / xl / xu are host arrays (this rank's shard)."""
    from .projpenalty import LazyProjPenaltyWork, ProjPenalty
    from . import _capi
    import ctypes as C
    param = param or LFPSQPParams()
    x0 = np.asarray(x0, dtype=np.float64)
    n = len(x0)
    n_global = n if n_global is None else n_global
    if xl is not None and xu is not None:
        if not (len(xl) == len(xu) == len(x0)):
            raise ValueError("xl, xu, and x0 must all be the same length")
    if (xl is None and xu is None) or (np.all(np.asarray(xl) == -np.inf) and np.all(np.asarray(xu) == np.inf)):
        ineq = False
        ineqdata = None
    else:
        ineq = True
        xl = np.asarray(xl, dtype=np.float64)
        xu = np.asarray(xu, dtype=np.float64)
        if np.any(xl > xu):
            raise ValueError("Infeasible: lower bounds cannot be greater than upper bounds")
        ineqdata = InequalityData(ctx, xl, xu)
        lamy_kkt = DeviceVector(ctx, n)
        hx = DeviceVector(ctx, n)

    def newvec():
        return StackedVector(ctx, n) if ineq else DeviceVector(ctx, n)

    x = newvec()
    x.upload(x0, 0)
    if ineq:
        generate_initial_y_(x, ineqdata)                                   # :179-182
    obj_values = []
    xnew, g, d, newton_d = newvec(), newvec(), newvec(), newvec()
    # device-resident constraint classes own their (mostly constant) Jct; otherwise the driver allocates it
    Jct = getattr(c_, "Jct", None) or DeviceMatrix(ctx, n, m, placed=True)
    assert Jct.n == n and Jct.m == m
    tmp_m = DeviceVector(ctx, max(m, 1))
    tmp_w = DeviceVector(ctx, n) if ineq else None
    cval = np.zeros(m)
    lam_kkt = np.zeros(m)
    lam_dev = DeviceVector(ctx, max(m, 1))
    term_cond = TerminationCondition.f_tol
    prev_grad_norm = 0.0
    noise = None

    # The basis Z (src/optimize.jl:191), ProjCGWork (:214) and the operator diagonal the fused iteration reads beside them: allocated TOGETHER,
    # by trial over pairs of candidate allocations (lfpsqp_basis_work_alloc_placed, FINDINGS.md 6: the speed of the fused kernel is a property of
    # the pair of allocations)
    diagonal_hessian = hasattr(hess_lag_vec_, "diag_")
    # ... unless the basis can stay in FACTORED form U = Jct W (FINDINGS.md 5.3): then there is no Z at all -- the fused projected-CG iteration, the
    # Newton step and the projections stream Jct and apply the m x m factor W on the side, the tangent setup skips its basis-forming product,
    # and the work vectors are placed against Jct.
    # The LIBRARY says whether this context can run projcg without Z for this Jct (one-pass kernels on, shape and leading dimension inside
    # their limits, or a sparse twin the nonzero path covers); where it cannot (LFPSQP_ONEPASS=-1, ld beyond the 32-bit lane offsets ...)
    # Z is materialised and every path has its two-pass form.
    # (a tridiagonal Hessian sent through the callback path -- DeviceOptions.tridiagonal_one_pass off -- needs the materialised basis)
    tri_callback = diagonal_hessian and getattr(hess_lag_vec_, "offdiag", None) is not None and not bool(getattr(ctx.options, "tridiagonal_one_pass", True))
    factored = (bool(ctx.options.factored_basis) and diagonal_hessian and not tri_callback and 4 <= m <= 1024
                and ctx.factored_basis_supported(Jct, getattr(c_, "Jsp", None)))
    # Allocation by trial costs tens of milliseconds (18 timed launches of F at n = 1e7, m = 128: 36 ms) and returns 3 % of every projected-CG
    # iteration: it pays after several hundred iterations.  A Lagrangian Hessian that is a multiple of the identity (f = |x - xc|^2 under linear
    # equalities, no bounds: BASELINE config 3) ends every truncated-Newton solve after ONE iteration and the outer loop after one or two --
    # such a run takes its first allocations.
    saved_tries = ctx.options.placement_tries
    few_cg = bool(getattr(hess_lag_vec_, "scalar_hessian", False)) and not ineq
    if few_cg and saved_tries > 1:
        ctx.set_placement(1)
    try:
        projcgwork, idecomp = _allocate_projcg_work(ctx, n, m, ineq, Jct, factored, diagonal_hessian)
    finally:
        if ctx.options.placement_tries != saved_tries:
            ctx.set_placement(saved_tries)
    Z = idecomp.Z
    # The tangent step with fewer passes (lfpsqp_tangent_step): plain factored basis over dense gradients, truncated-Newton steps on
    # (with bounds: the stacked form of the same pass; not over a matrix view, and not for a class whose Hessian term needs A*lambda)
    fuse_tangent = (factored and param.do_newton and getattr(c_, "Jsp", None) is None and m > 0 and bool(ctx.options.fused_tangent_step)
                    and not (ineq and (getattr(Jct, "is_view", False) or getattr(getattr(hess_lag_vec_, "cons", None), "kind", None) is not None)))
    ineq_rhs = DeviceVector(ctx, n) if (fuse_tangent and ineq) else None
    Ggram = np.zeros((m, m), order='F') if fuse_tangent else None         # the factorisation's Gram matrix: U'U = W'GW for the tangent step
    prev_rank = -1                           # rank of the previous outer iteration's factorisation (its Vt warm-starts the next one)
    Sig, Vt = idecomp.Sigma, idecomp.Vt
    Wgen = np.zeros((m, m), order='F') if m > 0 else None                  # ksvd_'s small factor: Z == Jct @ Wgen
    ineqproject = InequalityDecompProject(idecomp) if ineq else None

    # a TRIDIAGONAL Lagrangian Hessian: ``diag_`` fills the diagonal, ``offdiag`` (device n-vector, entry i couples variables i and i+1) holds the
    # couplings -- projcg_ keeps one pass per iteration with it (lfpsqp_projcg_tridiag; no bounds, one rank: the operator's own limits)
    tri_off = getattr(hess_lag_vec_, "offdiag", None) if diagonal_hessian else None
    if tri_off is not None:
        if ineq:
            raise NotImplementedError("a tridiagonal Hessian with bounds: pass hess_lag_vec_ as a callable (the generic path)")
        # (the tangent step's pass still hands projcg_ r0 and U'r0 -- neither involves A --, but never its folded initial projection, whose
        # sums are formed with the diagonal alone: init_fold stays off below)
        if not bool(getattr(ctx.options, "tridiagonal_one_pass", True)):
            fuse_tangent = False              # (the callback path starts its solves itself)
        if getattr(Jct, "is_view", False) or ctx.nranks > 1:
            # lfpsqp_projcg_tridiag answers a matrix view or an active communicator with LFPSQP_ERR_UNSUPPORTED (the couplings would cross the
            # shard boundaries): those solves take the callback path, which makes its own start -- a start handed over by the one-pass tangent
            # step (start_given) could not be honoured there and projcg_ would raise
            fuse_tangent = False
    if tri_off is not None:
        from .projcg import TridiagonalOperator
        a_diag = projcgwork.placed_extra[0] if projcgwork.placed_extra else newvec()
        newton_map = TridiagonalOperator(0.0, a_diag, tri_off)
        newton_map.fused = bool(getattr(ctx.options, "tridiagonal_one_pass", True))
    elif diagonal_hessian:
        a_diag = projcgwork.placed_extra[0] if projcgwork.placed_extra else newvec()
        newton_map = DiagOperator(0.0, a_diag)
    elif ineq:
        newton_map = _GenericAugHessian(hess_lag_vec_, x, lam_dev, lamy_kkt, ineqdata, n)
    else:
        newton_map = _GenericHessian(hess_lag_vec_, x, lam_dev)

    nr = NR(None, Sig, Vt, param.eps_c, param.maxiter_retract, NRWork(m), ineq, ineqdata)
    pp = ProjPenalty(jac_, None, Sig, Vt, m, param.mu0, param.eps_c, param.maxiter_retract, param.maxiter_pcg,
                     LazyProjPenaltyWork(ctx, m, n, ineq, against=Jct if (m > 0 and not few_cg) else None), ineq, idecomp, ineqdata)
    euc = Euclidean()
    yr = YRetract(ineqdata) if ineq else None
    ctx.set_nr_batch_mode(bool(getattr(ctx.options, "ls_batch_matrix_cores", False)))
    armijo_work = ArmijoWork(x)
    exact_work = ExactLinesearchWork(x) if param.linesearch == LinesearchOption.exact and not param.disable_linesearch else None

    i = 0
    cval_current = m > 0                     # cval == c(x): true after the evaluation below, and after every accepted retraction
    f_diff = step_diff = kkt_diff = math.inf
    fval = f(x)
    obj_values.append(fval)
    if m > 0:
        if hasattr(c_, "c_"):
            c_.c_(cval, x)
        else:
            c_(cval, x.download(n, 0))
    disp = param.disp == DisplayOption.iter and ctx.rank == 0
    if disp:
        _print_iter_header()
        _print_first_line(fval, _amax_host(cval))

    while True:
        grad_(g, x)                                                        # :259
        waxpby(-1.0, g, 0.0, g, d)                                         # :262
        if param.beta > 0:                                                 # :264-273 (randn! noise; Julia's RNG stream cannot be matched)
            if noise is None:
                noise = newvec()
            z = np.random.standard_normal(2 * n if ineq else n)
            if ineq:
                noise.upload2(z)
            else:
                noise.upload(z)
            scale = param.beta * max(1 - i / param.t_beta, 0.0) if param.t_beta > 0 else param.beta
            axpby(scale, noise, 1.0, d)
        if ineq:
            inequality_gradient_(idecomp, x, ineqdata)                     # :277
        rank = m
        if m > 0:
            # :283-284 (the device keeps only Jct).  A device-resident class does not evaluate c(x) again when cval holds it: x is the point the
            # initial c! or the line search's accepted retraction evaluated it at (the reference's jac! recomputes it: one pass for nothing)
            if cval_current and getattr(jac_, "__self__", None) is c_ and hasattr(c_, "_c"):
                jac_(Jct, cval, x, evaluate=False)
            else:
                jac_(Jct, cval, x)
            Jtd = None
            init_fold = False
            vt_prev = Vt if (i > 0 and prev_rank == m and ctx.options.warm_factorize) else None
            if fuse_tangent and ineq:                                      # ... with bounds: Jct'(sx .* dx + sy .* dy), the m-part of Q'd
                ctx.check(ctx.L.lfpsqp_ineq_rhs(ctx.h, d.h, idecomp.Dx.h, idecomp.Dy.h, ineq_rhs.h))
                S_, Vt_, rank, Jtd = ksvd_(Jct, Z, w2=idecomp.sx, eps_rank=param.eps_rank, W=Wgen, Vt_prev=vt_prev, rhs=ineq_rhs, G_out=Ggram)
            elif fuse_tangent:                                             # Jct'd rides with the Gram pass (d is final before jac! runs)
                S_, Vt_, rank, Jtd = ksvd_(Jct, Z, eps_rank=param.eps_rank, W=Wgen, Vt_prev=vt_prev, rhs=d, G_out=Ggram)    # :286-302
            else:
                S_, Vt_, rank = ksvd_(Jct, Z, w2=idecomp.sx if ineq else None, eps_rank=param.eps_rank, W=Wgen,
                                      Jsp=getattr(c_, "Jsp", None), Vt_prev=vt_prev)                            # :286-302
            prev_rank = rank
            idecomp.W = Wgen
            idecomp.Jsp = getattr(c_, "Jsp", None)
            Sig[:] = S_
            Vt[:, :] = Vt_
            fused_now = fuse_tangent and rank >= 1
            if fused_now:
                # :305-343, :366-381 and src/projcg.jl:56-59 in one pass: d projected, lambda_kkt, the Hessian diagonal completed, r0 = -d and
                # U'r0 left in projcgwork for projcg_(start_given=True)
                if ineq:
                    idecomp.rank = rank
                    bs = ineqproject._c()
                else:
                    bs = DeviceBasis(None, rank, generator=(Jct, Wgen))._c()
                hdst = hx if ineq else a_diag                              # where the objective's part of the diagonal goes (bounds: the x-half alone)
                split = hasattr(hess_lag_vec_, "diag_objective_") and getattr(hess_lag_vec_, "cons", None) is not None
                th = np.zeros(m)
                dss = C.c_double()
                if split:
                    hess_lag_vec_.diag_objective_(hdst, x)
                    cc = hess_lag_vec_.cons._c()
                else:                                                      # a diagonal Hessian without the split: it needs lambda_kkt first
                    th[:rank] = Wgen[:, :rank].T @ Jtd
                    th[:rank] /= S_[:rank]
                    lam_kkt[:] = Vt_.T @ th
                    hess_lag_vec_.diag_(hdst, x, lam_kkt)
                    cc = None
                wc = projcgwork._c()
                sig_c = np.ascontiguousarray(S_, dtype=np.float64)
                vt_c = np.asfortranarray(Vt_, dtype=np.float64)
                idc = ineqdata._c() if ineq else None
                # (flag 1 = LFPSQP_TANGENT_INIT_PROJCG: the pass is projcg!'s initial projection as well -- src/projcg.jl:58-62 -- with U'r0 from
                # the Gram matrix; projcg_ then starts with its first iteration, start_projected=True.  Only where the Gram matrix resolves
                # I - U'U: a full-rank block with cond^2 <= 10, the fast path of the factorisation; otherwise projcg_ measures U'r0 itself)
                init_fold = bool(tri_off is None and rank == m and S_[0] * S_[0] <= 10.0 * S_[m - 1] * S_[m - 1])
                ctx.check(ctx.L.lfpsqp_tangent_step(ctx.h, C.byref(bs), sig_c.ctypes.data, vt_c.ctypes.data, m, Jtd.ctypes.data, Ggram.ctypes.data, d.h,
                                                    C.byref(cc) if cc is not None else None, x.h, a_diag.h,
                                                    C.byref(idc) if ineq else None, hx.h if ineq else None, idecomp.S.h if ineq else None,
                                                    lamy_kkt.h if ineq else None, C.byref(wc), 1 if init_fold else 0, th.ctypes.data,
                                                    lam_kkt.ctypes.data, C.byref(dss)))
            elif not ineq:                                                 # :305-308
                jsp_ = getattr(c_, "Jsp", None)
                if jsp_ is not None or Z is None:                          # sparse twin / factored basis: U = Jct W applied without Z
                    Ub = DeviceBasis(Z, rank, generator=(Jct, Wgen), sparse=jsp_)
                    Ub.adjoint().mul_(tmp_m, d)
                    Ub.mul_(d, tmp_m, -1.0, 1.0)
                else:
                    gemv_t(Z, d, tmp_m, ncols=rank)
                    gemv_n(Z, tmp_m, d, -1.0, 1.0, ncols=rank)
        fused_now = bool(fuse_tangent and m > 0 and rank >= 1)
        if ineq and not fused_now:                                         # :312-318
            idecomp.rank = rank
            ineqproject.mul_t(tmp_w, tmp_m, d)
            ineqproject.mul_n(d, tmp_w, tmp_m, -1.0, 1.0)
        kkt_diff = amax(d)                                                 # :320
        pp.rank = rank
        steptype = 0
        tn_iter = 0
        tn_res = 0.0
        if m > 0 and fuse_tangent and rank >= 1:
            pass                                                           # (lambda_kkt came back from the tangent step; lam_dev has no reader here)
        elif m > 0:                                                        # :331-343
            th = tmp_m.download(m)
            th[:rank] /= Sig[:rank]
            th[rank:m] = 0.0
            lam_kkt[:] = Vt.T @ th
            lam_dev.upload(lam_kkt)
        if ineq and not fused_now:                                         # calculate_lambda_kkt!, :286-308
            ctx.check(ctx.L.lfpsqp_calculate_lambda_y(ctx.h, Jct.h, m, lam_dev.h, idecomp.Dx.h, idecomp.S.h, tmp_w.h, lamy_kkt.h))

        if trace is not None:
            trace.append(dict(iter=i, x=(x.download2() if ineq else x.download()), fval=fval, kkt_diff=kkt_diff, rank=rank,
                              lam_kkt=lam_kkt.copy(), cval=cval.copy()))

        if f_diff <= param.eps_f:                                          # :347-359
            term_cond = TerminationCondition.f_tol
            break
        elif step_diff <= param.eps_x:
            term_cond = TerminationCondition.x_tol
            break
        elif i >= param.maxiter:
            term_cond = TerminationCondition.max_iter
            break
        elif kkt_diff <= param.eps_kkt:
            term_cond = TerminationCondition.kkt_tol
            break

        if param.do_newton:                                                # :364-390
            if diagonal_hessian and fuse_tangent and rank >= 1:
                pass                                                       # (a_diag was completed by the tangent step)
            elif diagonal_hessian:
                if ineq:
                    hess_lag_vec_.diag_(hx, x, lam_kkt)
                    idc = ineqdata._c()
                    ctx.check(ctx.L.lfpsqp_augmented_diag(ctx.h, hx.h, lamy_kkt.h, C.byref(idc), a_diag.h))
                else:
                    hess_lag_vec_.diag_(a_diag, x, lam_kkt)
            if ineq:
                Qview = ineqproject
            else:
                jsp_ = getattr(c_, "Jsp", None)                              # sparse twin: the projected CG runs on the nonzeros
                Qview = DeviceBasis(Z, rank, generator=(Jct, Wgen), sparse=jsp_) if (jsp_ is not None or Z is None) else DeviceBasis(Z, rank)
            # (the one-pass tangent step returns the all-reduced d'd of the projected step: no further pass over d, no host sync)
            grad_norm = math.sqrt(dss.value) if fused_now else nrm2(d)
            with np.errstate(divide='ignore', invalid='ignore'):
                ratio = float(np.float64(grad_norm) / np.float64(prev_grad_norm))
            tol = param.tn_kappa * min(1.0, ratio) * grad_norm
            if math.isnan(ratio):
                tol = math.nan
            prev_grad_norm = grad_norm
            tn_iter, tn_res = projcg_(newton_d, None, newton_map, Qview, d, None, tol=tol, maxit=param.tn_maxiter,
                                      work=projcgwork, n_global=(2 * n_global if ineq else n_global), want_lambda=False,
                                      start_projected=bool(fuse_tangent and rank >= 1 and init_fold),
                                      start_given=bool(fuse_tangent and rank >= 1 and not init_fold))
            if dot(newton_d, d) > 0.0:
                d.copy_from(newton_d)
                steptype = 1
            if trace is not None:
                trace[-1].update(tn_iter=tn_iter, tn_res=tn_res, steptype=steptype, tn_tol=tol)

        if m > 0:                                                          # :396-412
            if rank == m and not param.do_project_retract:
                nr.U = ineqproject if ineq else DeviceBasis(Z, rank, generator=(Jct, Wgen))
                retract_method, mtype = nr, 0
            else:
                retract_method, mtype = pp, 1
        else:
            retract_method, mtype = (yr, 0) if ineq else (euc, 0)

        if param.linesearch == LinesearchOption.armijo or param.disable_linesearch:
            flag, iter1, iter2, newf, f_diff, step_diff, alpha = armijo_(
                xnew, x, n, d, g, f, fval, retract_method, cval, c_, param, armijo_work)
        else:
            flag, iter1, iter2, newf, f_diff, step_diff, alpha = exact_linesearch_(
                xnew, x, n, d, f, fval, retract_method, cval, c_, param, exact_work)

        x.copy_from(xnew)                                                  # :424-427
        # (Armijo's accepted trial is the last one retracted, and its retraction returned c!(xnew); the exact search returns a SAVED best point
        # while cval holds the values of the last trial retracted -- as in the reference, src/linesearch.jl:96-230)
        cval_current = (flag == 0 and m > 0 and (param.linesearch == LinesearchOption.armijo or param.disable_linesearch))
        fval = newf
        obj_values.append(fval)
        if disp:
            _print_iter(i + 1, fval, _amax_host(cval), f_diff, step_diff, steptype, tn_iter, tn_res, mtype, iter1, iter2, alpha, flag)
        if trace is not None:
            trace[-1].update(mtype=mtype, retract_iter1=iter1, retract_iter2=iter2, alpha=alpha, ls_flag=flag)
        i += 1
        if param.callback is not None and i % param.callback_period == 0:
            param.callback(i, x)
            cval_current = False                 # the callback holds the live iterate: if it edits x, cval is stale (the reference's jac! recomputes it, :283)

    if i == param.maxiter and disp:
        print("Warning: Maximum # of outer iterations reached")
    return x.download(n, 0), np.array(obj_values), lam_kkt, TerminationInfo(term_cond, f_diff, step_diff, kkt_diff, i)
