"""armijo! / exact_linesearch! (reference src/linesearch.jl): host control flow, every trial
point retracted on the device.  Vectors are device vectors; ``f`` maps a device vector to a float."""
from __future__ import annotations

import math

from .device import DeviceVector, dot, nrm2, waxpby
from .retractions import retract_, retract_nr_batch_, retract_nr_batch_width_


class ArmijoWork:  # src/linesearch.jl:1-5
    def __init__(self, like: DeviceVector):
        self._mk = (lambda: like.__class__(like.ctx, like.N)) if hasattr(like, "N") else (lambda: DeviceVector(like.ctx, like.n))
        self.xtilde = self._mk()
        self.batch = None          # lazily: (xtildes, xnews) for batched trial retractions
        self.ls_batch = int(like.ctx.options.ls_batch)      # DeviceOptions: trial retractions per pass
        self.prev_failed = False   # did the previous search see a failed retraction? (then this one batches from its first trial)
        self.prev_failures = 0     # ... and how many: the automatic batch width follows it

    def batch_vectors(self, k):
        return _grow_batch(self, k)


class ExactLinesearchWork:  # :7-14
    def __init__(self, like: DeviceVector):
        mk = (lambda: like.__class__(like.ctx, like.N)) if hasattr(like, "N") else (lambda: DeviceVector(like.ctx, like.n))
        self._mk = mk
        self.tmp_n1, self.tmp_n2, self.tmp_n3, self.tmp_n4 = mk(), mk(), mk(), mk()
        self.batch = None          # lazily: (xtildes, xnews) for batched trial retractions of the shrinking phase
        self.ls_batch = int(like.ctx.options.ls_batch)
        self.prev_failures = 0

    def batch_vectors(self, k):
        return _grow_batch(self, k)


def _grow_batch(work, k):
    """(xtildes, xnews): up to k pairs of trial vectors, kept for the rest of the run (k = 16 with the stacked bound layout at n = 1e7 is 5 GB).
    The lists GROW pair by pair; when device memory runs out they stay as long as they got and the search batches that many -- a run that
    one-by-one retractions would have completed does not fail for want of a wider pass."""
    from ._capi import LfpsqpError
    if work.batch is None:
        work.batch = ([], [])
    while len(work.batch[0]) < k:
        a = b = None
        try:
            a = work._mk()
            b = work._mk()
        except LfpsqpError:
            if a is not None:
                a.free()
            break
        work.batch[0].append(a)
        work.batch[1].append(b)
    return work.batch


def _batch_width(work, retract_method, c_):
    """Trial retractions per pass for this search: DeviceOptions.ls_batch (1 = off, k > 1 = at most k, 0 = automatic: what the
    previous search's failures suggest, at least 4), never more than the library takes for this shape."""
    opt = int(getattr(work, "ls_batch", 1))
    if opt == 1:
        return 1
    width = retract_nr_batch_width_(c_, retract_method)
    if width < 2:
        return 1
    if opt > 1:
        return min(opt, width)
    return min(width, max(4, int(getattr(work, "prev_failures", 0)) + 2))


def _batch_cap(work, retract_method, c_):
    """The widest pass of this search: a batch whose trials ALL failed is followed by one twice as wide, up to this (the first search of a
    run knows nothing of the failures ahead: config 4's takes 25 trial steps -- six passes of four, or 4 + 8 + 16)."""
    opt = int(getattr(work, "ls_batch", 1))
    if opt == 1:
        return 1
    width = retract_nr_batch_width_(c_, retract_method)
    return 1 if width < 2 else (min(opt, width) if opt > 1 else width)


def _step_norm(step, n_head):
    """norm(view(step, 1:n)) (src/linesearch.jl:66): only the first n entries."""
    from .device import nrm2_head
    return nrm2_head(step, n_head)


def armijo_(xnew, x, n, d, g, f, fval, retract_method, cval, c_, param, work):
    """src/linesearch.jl:32-89."""
    f_diff = math.inf
    step_diff = math.inf
    alpha = param.alpha
    flag = 0
    tot_iter1 = tot_iter2 = 0
    newf = 0.0
    ar_dot = dot(d, g)
    xtilde = work.xtilde
    step = xtilde
    # Trial retractions ahead of time: alpha -> (flag, iter1, iter2, xnew_b, cval_b).  Filled, after the first failure of
    # this search (or from its first trial when the previous search had failures), with the next `ls_batch` (ctx.options.ls_batch, a device option: not in LFPSQPParams) steps of the reference's own sequence alpha*s, alpha*s^2, ... -- the loop
    # below consumes them exactly as it would have computed them one by one.
    ahead = {}
    nbatch = _batch_width(work, retract_method, c_)          # (only Newton retractions on device-resident constraints batch)
    nbatch_cap = _batch_cap(work, retract_method, c_) if nbatch > 1 else 1
    failed_once = bool(getattr(work, "prev_failed", False))   # searches in a failing regime batch from their first trial
    any_failed = False
    n_failed = 0
    while step_diff > param.eps_x:
        if alpha in ahead:
            flag, iter1, iter2, xb, cb = ahead.pop(alpha)
            xnew.copy_from(xb)
            cval[:] = cb
        else:
            got = None
            if failed_once and nbatch > 1 and not param.disable_linesearch:
                xts, xns = (vs_[:nbatch] for vs_ in work.batch_vectors(nbatch))      # (the cache may hold more from an earlier, wider batch)
                if len(xts) < nbatch:                          # device memory ran out: as wide as the vectors that exist
                    nbatch = nbatch_cap = max(len(xts), 1)
                    if nbatch < 2:
                        failed_once = False
                        continue
                alphas = [alpha]
                for _ in range(nbatch - 1):
                    alphas.append(alphas[-1] * param.s)
                for a_, xt_ in zip(alphas, xts):
                    waxpby(1.0, x, a_, d, xt_)
                import numpy as _np
                cvs = _np.zeros((nbatch, len(cval)))
                got = retract_nr_batch_(cvs, xns, c_, xts, x, retract_method)
                if got is not None:
                    for a_, res, xb, cb in zip(alphas[1:], got[1:], xns[1:], cvs[1:]):
                        ahead[a_] = (res[0], res[1], res[2], xb, cb.copy())
                    flag, iter1, iter2 = got[0]
                    xnew.copy_from(xns[0])
                    cval[:] = cvs[0]
                    if all(res[0] > 0 for res in got):         # every trial of the pass failed: the next pass takes twice as many
                        nbatch = min(nbatch_cap, 2 * nbatch)
                else:
                    nbatch = 1                                 # this configuration cannot batch
            if got is None:
                waxpby(1.0, x, alpha, d, xtilde)             # xtilde = x + alpha d
                flag, iter1, iter2 = retract_(cval, xnew, c_, xtilde, x, retract_method)
        if flag > 0:
            failed_once = any_failed = True
            n_failed += 1
        tot_iter1 += iter1
        tot_iter2 += iter2
        if flag > 0:
            alpha *= param.s
            continue
        waxpby(1.0, xnew, -1.0, x, step)                 # step = xnew - x
        newf = f(xnew)
        step_diff = _step_norm(step, n)
        f_diff = abs(newf - fval)
        if param.disable_linesearch:
            break
        if (newf - fval) <= param.sigma * alpha * ar_dot:
            break
        alpha *= param.s
        if alpha < 1e-100:
            flag = 99
            break
    work.prev_failed = any_failed
    work.prev_failures = n_failed
    return flag, tot_iter1, tot_iter2, newf, f_diff, step_diff, alpha


def exact_linesearch_(xnew, x, n, d, f, fval, retract_method, cval, c_, param, work):
    """src/linesearch.jl:107-339 (same rotation of the four work vectors as the reference)."""
    phi1 = (3 - math.sqrt(5)) / 2
    phi2 = (math.sqrt(5) - 1) / 2
    phi3 = (math.sqrt(5) + 1) / 2
    Delta = param.alpha
    flag = 0
    tot = [0, 0]
    f_a = f_b = f_c = f_d = 0.0
    a_a = a_b = a_c = a_d = 0.0
    x_a, x_b, x_c, x_d = work.tmp_n1, work.tmp_n2, work.tmp_n3, work.tmp_n4
    step = work.tmp_n1
    do_shrinking = True

    def _retract(pt):
        fl, i1, i2 = retract_(cval, xnew, c_, pt, x, retract_method)
        tot[0] += i1
        tot[1] += i2
        pt.copy_from(xnew)
        return fl

    # Shrinking phase (:176-208): the trial steps a_c*phi1, a_c*phi1^2, ... are a fixed sequence, each retracted from the same x,
    # and in the regime where it runs (the first trial of the search failed) most of them fail after the full iteration limit.
    # The next `ls_batch` of them are retracted together (lfpsqp_retract_nr_batch: one pass over Jct per Newton step for all) and
    # consumed in the reference's order -- same points, flags and counts as one by one; unconsumed look-ahead is not counted.
    nbatch = _batch_width(work, retract_method, c_)
    ahead = {}
    n_shrink_failed = [0]

    def _retract_shrink(pt, a_next):
        """retract x + a_next*d into pt (pt already holds xtilde), looking ahead along a_next*phi1^k"""
        nonlocal nbatch
        if a_next in ahead:
            fl, i1, i2, xb, cb = ahead.pop(a_next)
            xnew.copy_from(xb)
            cval[:] = cb
            pt.copy_from(xnew)
            tot[0] += i1
            tot[1] += i2
            return fl
        if nbatch > 1:
            import numpy as _np
            xts, xns = (vs_[:nbatch] for vs_ in work.batch_vectors(nbatch))      # (the cache may hold more from an earlier, wider batch)
            if len(xts) < nbatch:                              # device memory ran out: as wide as the vectors that exist
                nbatch = len(xts)
        if nbatch > 1:
            alphas = [a_next]
            for _ in range(nbatch - 1):
                alphas.append(alphas[-1] * phi1)          # the reference's own products a_c *= phi1 (:196)
            xts[0].copy_from(pt)
            for a_, xt_ in zip(alphas[1:], xts[1:]):
                waxpby(1.0, x, a_, d, xt_)
            cvs = _np.zeros((nbatch, len(cval)))
            got = retract_nr_batch_(cvs, xns, c_, xts, x, retract_method)
            if got is not None:
                for a_, res, xb, cb in zip(alphas[1:], got[1:], xns[1:], cvs[1:]):
                    ahead[a_] = (res[0], res[1], res[2], xb, cb.copy())
                fl, i1, i2 = got[0]
                xnew.copy_from(xns[0])
                cval[:] = cvs[0]
                pt.copy_from(xnew)
                tot[0] += i1
                tot[1] += i2
                return fl
            nbatch = 1                                     # this configuration cannot batch
        return _retract(pt)

    x_d.copy_from(x)
    f_d = fval
    while True:
        x_b, x_c, x_d = x_c, x_d, x_b
        f_b, f_c = f_c, f_d
        a_b, a_c = a_c, a_d
        waxpby(1.0, x, a_d + Delta, d, x_d)
        flag = _retract(x_d)
        a_d += Delta
        if flag > 0 or a_d > 1.0:
            f_d = math.inf
            break
        f_d = f(x_d)
        if f_d > f_c:
            break
        do_shrinking = False
        Delta *= phi3
    if do_shrinking:
        f_b = fval
        a_b = 0.0
        x_b.copy_from(x)
        f_c = math.inf
        a_c = Delta
        x_d, x_c = x_c, x_d
        while True:
            x_d, x_c = x_c, x_d
            f_d = f_c
            a_d = a_c
            waxpby(1.0, x, phi1 * a_c, d, x_c)
            flag = _retract_shrink(x_c, phi1 * a_c)
            n_shrink_failed[0] += 1 if flag > 0 else 0
            a_c *= phi1
            f_c = math.inf if (flag > 0 or a_c > 1.0) else f(x_c)
            if f_c <= fval or a_c < 1e-100:
                break
    f_a, f_b = f_b, f_c
    a_a, a_b = a_b, a_c
    x_a, x_b, x_c = x_b, x_c, x_a
    a_c = a_a + phi2 * (a_d - a_a)
    waxpby(1.0, x, a_c, d, x_c)
    flag = _retract(x_c)
    f_c = math.inf if (flag > 0 or a_c > 1.0) else f(x_c)
    nd = nrm2(d)
    while (a_c - a_b) > 1e-6 * nd:
        if f_b < f_c or math.isinf(f_c):
            x_d, x_c, x_b = x_c, x_b, x_d
            f_d, f_c = f_c, f_b
            a_d, a_c = a_c, a_b
            a_b = a_a + phi1 * (a_d - a_a)
            waxpby(1.0, x, a_b, d, x_b)
            flag = _retract(x_b)
            f_b = f(x_b)
        else:
            x_a, x_b, x_c = x_b, x_c, x_a
            f_a, f_b = f_b, f_c
            a_a, a_b = a_b, a_c
            a_c = a_a + phi2 * (a_d - a_a)
            waxpby(1.0, x, a_c, d, x_c)
            flag = _retract(x_c)
            f_c = math.inf if (flag > 0 or a_c > 1.0) else f(x_c)
    if f_b < f_c:
        xnew.copy_from(x_b)
        newf, alpha = f_b, a_b
    else:
        xnew.copy_from(x_c)
        newf, alpha = f_c, a_c
    waxpby(1.0, xnew, -1.0, x, step)
    step_diff = _step_norm(step, n)
    f_diff = abs(newf - fval)
    work.prev_failures = n_shrink_failed[0]
    return flag, tot[0], tot[1], newf, f_diff, step_diff, alpha
