"""armijo! / exact_linesearch! (reference src/linesearch.jl): host control flow, every trial
point retracted on the device.  Vectors are device vectors; ``f`` maps a device vector to a float."""
from __future__ import annotations

import math

from .device import DeviceVector, dot, nrm2, waxpby
from .retractions import retract_


class ArmijoWork:  # src/linesearch.jl:1-5
    def __init__(self, like: DeviceVector):
        self.xtilde = like.__class__(like.ctx, like.N) if hasattr(like, "N") else DeviceVector(like.ctx, like.n)


class ExactLinesearchWork:  # :7-14
    def __init__(self, like: DeviceVector):
        mk = (lambda: like.__class__(like.ctx, like.N)) if hasattr(like, "N") else (lambda: DeviceVector(like.ctx, like.n))
        self.tmp_n1, self.tmp_n2, self.tmp_n3, self.tmp_n4 = mk(), mk(), mk(), mk()


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
    while step_diff > param.eps_x:
        waxpby(1.0, x, alpha, d, xtilde)                 # xtilde = x + alpha d
        flag, iter1, iter2 = retract_(cval, xnew, c_, xtilde, x, retract_method)
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
            flag = _retract(x_c)
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
    return flag, tot[0], tot[1], newf, f_diff, step_diff, alpha
