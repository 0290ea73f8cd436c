"""The drop-in boundary used from plain C (tests/c_consumer/consumer.c, compiled with gcc -std=c99 against
include/lfpsqp_hip.h and linked to the shared library): same numbers as the Python host layer over the same library."""
import os
import subprocess

import numpy as np
import pytest

import lfpsqp_jl_amd as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_consumer", "consumer.c")


def _build_and_run(libpath, n, m, tol, maxit, tmp_path, src=SRC):
    exe = str(tmp_path / os.path.basename(src)[:-2])
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L" + libdir, "-l:" + libname, "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, str(n), str(m), repr(tol), str(maxit)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    return dict(line.split("=", 1) for line in out.stdout.strip().splitlines())


def _python_side(ctx, n, m, tol, maxit):
    J = ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
    Z = ctx.matrix(n, m)
    S, Vt, rank = L.ksvd_(J, Z)
    x, lam = ctx.vector(n), ctx.vector(m)
    it, nr = L.projcg_(x, lam, L.DiagOperator(0.0, ctx.vector(n).hash_fill(3, 0, 4.0, 5.0)), L.DeviceBasis(Z), ctx.vector(n).hash_fill(4), None,
                       tol=tol, maxit=maxit)
    return rank, it, nr, L.nrm2(x), L.nrm2(lam), float(S[0])


@pytest.mark.parametrize("n,m", [(5000, 12), (6011, 33)])
def test_c_consumer_matches_host_layer(dev_ctx, n, m, tmp_path):
    tol, maxit = 1e-9, 200
    got = _build_and_run(dev_ctx.L.path, n, m, tol, maxit, tmp_path)
    rank, it, nr, xn, ln, s0 = _python_side(dev_ctx, n, m, tol, maxit)
    assert got["device"] == dev_ctx.device_name
    assert int(got["rank"]) == rank == m and int(got["iters"]) == it
    # the same library, the same calls: identical bits
    assert float.fromhex(got["nr"]) == nr and float.fromhex(got["xnorm"]) == xn and float.fromhex(got["lnorm"]) == ln
    assert float.fromhex(got["sigma0"]) == s0


def test_c_consumer_with_an_operator_callback(dev_ctx, tmp_path):
    """lfpsqp_projcg_op from plain C: a tridiagonal A as a callback made of queued library primitives, against the ORACLE's
    projcg! with the same operator on the same orthonormal basis (counts equal, iterate and multipliers to 1e-10), and the
    operator applied once per iteration plus start and lambda."""
    from oracle import lfpsqp_ref as R
    from oracle import synth
    n, m, tol, maxit = 6000, 9, 1e-10, 300
    got = _build_and_run(dev_ctx.L.path, n, m, tol, maxit, tmp_path, src=os.path.join(ROOT, "tests", "c_consumer", "consumer_op.c"))
    # the same basis through the Python layer (same library, same calls: same bits), downloaded for the oracle
    J = dev_ctx.matrix(n, m).hash_fill(1, 0, n, 1.0)
    Z = dev_ctx.matrix(n, m)
    L.ksvd_(J, Z)
    Uh = Z.download()
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    e = 0.8 * synth.hash_vector(15, n)[:n - 1]
    bh = synth.hash_vector(4, n)

    class Tri:
        def mul_(self, dest, v, al=None, be=None):
            t = a * v
            t[:-1] += e * v[1:]
            t[1:] += e * v[:-1]
            dest[:] = t if al is None else al * t + be * dest
            return dest

        def adjoint(self):
            return self
    x0, l0 = np.zeros(n), np.zeros(m)
    i0, nr0 = R.projcg_(x0, l0, Tri(), Uh, bh, np.zeros(m), tol=tol, maxit=maxit)
    assert int(got["rank"]) == m and int(got["iters"]) == i0
    assert float.fromhex(got["nr"]) == pytest.approx(nr0, rel=1e-5)
    assert float.fromhex(got["xnorm"]) == pytest.approx(np.linalg.norm(x0), rel=1e-10)
    assert float.fromhex(got["lnorm"]) == pytest.approx(np.linalg.norm(l0), rel=1e-9)
    assert float.fromhex(got["x0"]) == pytest.approx(x0[0], rel=1e-8, abs=1e-12) and float.fromhex(got["xlast"]) == pytest.approx(x0[-1], rel=1e-8, abs=1e-12)
    assert i0 + 2 <= int(got["calls"]) <= i0 + 5
