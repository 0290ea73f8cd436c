"""TEST INFRASTRUCTURE: the ProjPenalty retraction statement by statement in Python on the device primitives -- a readable
mirror of src/retractions.jl:265-441 used as a cross-check of the C implementation (lfpsqp_retract_pp).  Not part of the
product package."""
import math

import numpy as np

from lfpsqp_jl_amd.device import axpby, dot, waxpby
from lfpsqp_jl_amd.inequality import calculate_h_, inequality_gradient_
from lfpsqp_jl_amd.projpenalty import ProjPenalty, _call_c, _JacPlain, _JacStacked, no_precondition, pcg_


def retract_pp_reference_loop(cval, xnew, c_, xtilde, x, method: ProjPenalty):
    """The same retraction statement by statement in Python on the device primitives (kept as a readable
    mirror of src/retractions.jl:265-441 and as a cross-check of the C implementation in the tests)."""
    w = method.work
    idecomp, idata = method.idecomp, method.idata
    Jct = idecomp.Jct
    r, p, z, dx, g = w.r, w.p, w.z, w.dx, w.g
    ineq = method.ineq
    n = idecomp.N
    m = len(method.Sigma)
    J = _JacStacked(idecomp, w) if ineq else _JacPlain(Jct, w)
    mu0, tol, maxiter, maxiter_pcg = method.mu0, method.tol, method.maxiter, method.maxiter_pcg
    flag = 0
    xnew.copy_from(xtilde)                                   # :329
    mu = mu0
    i = 0
    pcg_iter_count = 0
    hh = 0.0
    while i < maxiter:
        method.jac_(Jct, cval, xnew)                         # :340 (device Jct == transpose!(idecomp.Jct, J), :347)
        curtol = float(np.max(np.abs(cval), initial=0.0))
        if ineq:                                             # :343-353
            inequality_gradient_(idecomp, xnew, idata)
            J.refresh()
            hmax = calculate_h_(w.h, xnew, idata)
            curtol = hmax if (math.isnan(hmax) or hmax > curtol) else curtol     # Julia max propagates NaN
            hh = dot(w.h, w.h)
        if curtol < tol:                                     # :359
            break
        waxpby(1.0, xnew, -1.0, xtilde, g)                   # :364
        cc = float(np.dot(cval, cval))
        prev_obj_val = (hh + cc) + mu * dot(g, g)            # :366
        w.cval_dev.upload(cval)
        J.apply_t(g, 1.0, mu, from_cval=True)                # :369  g = fulljac' cvalaug + mu g
        dx.fill(0.0)
        r.copy_from(g)
        pcg_flag, pcg_i = pcg_(mu, J, no_precondition, dx, r, p, z, None, tol, maxiter_pcg)   # :375
        pcg_iter_count += pcg_i
        if pcg_flag > 0:                                     # :377-381
            flag = 2
            break
        p.copy_from(xnew)                                    # :384
        ar_dot = -dot(g, dx)                                 # :385
        alpha = 1.0
        axpby(-alpha, dx, 1.0, xnew)                         # :389
        waxpby(1.0, xnew, -1.0, xtilde, g)
        dist2 = dot(g, g)
        _call_c(c_, cval, xnew, n)                           # :392
        if ineq:
            calculate_h_(w.h, xnew, idata, want_max=False)
            hh = dot(w.h, w.h)
        cc = float(np.dot(cval, cval))                       # :399 cvalaug[end-m+1:end] = cval
        armijo_count = 0
        while (hh + cc) + mu * dist2 > prev_obj_val + 1e-4 * alpha * ar_dot:   # :403
            alpha /= 2
            waxpby(1.0, p, -alpha, dx, xnew)
            waxpby(1.0, xnew, -1.0, xtilde, g)
            dist2 = dot(g, g)
            # BUG-COMPAT :410-417: c! is evaluated into cvalaug and then overwritten by the stale
            # full-step cval, so only the bound part h and dist2 change; the (discarded) c! call is skipped.
            if ineq:
                calculate_h_(w.h, xnew, idata, want_max=False)
                hh = dot(w.h, w.h)
            armijo_count += 1
            if armijo_count == 100:                          # :422-425 (leaves only the inner loop)
                flag = 3
                break
        i += 1
        mu = min(mu * 0.1, math.sqrt(hh + cc))               # :431
    if i == maxiter:                                         # :435-437
        flag = 1
    return flag, i, pcg_iter_count
