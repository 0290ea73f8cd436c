"""The drop-in boundary used from plain C (tests/c_consumer/consumer.c, compiled with gcc -std=c99 against
include/lfpsqp_hip.h and linked to the shared library): same numbers as the Python host layer over the same library."""
import os
import subprocess

import numpy as np
import pytest

import lfpsqp_jl_amd as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_consumer", "consumer.c")


def _build_and_run(libpath, n, m, tol, maxit, tmp_path):
    exe = str(tmp_path / "consumer")
    libdir, libname = os.path.dirname(libpath), os.path.basename(libpath)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
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


@pytest.mark.parametrize("n,m", [(5000, 12), (20011, 33)])
def test_c_consumer_matches_host_layer(dev_ctx, n, m, tmp_path):
    tol, maxit = 1e-9, 200
    got = _build_and_run(dev_ctx.L.path, n, m, tol, maxit, tmp_path)
    rank, it, nr, xn, ln, s0 = _python_side(dev_ctx, n, m, tol, maxit)
    assert got["device"] == dev_ctx.device_name
    assert int(got["rank"]) == rank == m and int(got["iters"]) == it
    # the same library, the same calls: identical bits
    assert float.fromhex(got["nr"]) == nr and float.fromhex(got["xnorm"]) == xn and float.fromhex(got["lnorm"]) == ln
    assert float.fromhex(got["sigma0"]) == s0
