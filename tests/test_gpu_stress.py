"""Regression guard for the stale-scalar-cache hazard (FINDINGS.md §6): the failure mode was
per PROCESS (about one process in three returned wrong iterates on tiny problems, where the
kernels are a few microseconds long), so the stress loop runs in several fresh processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("proc", range(4))
def test_many_tiny_solves_in_a_fresh_process(proc):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_small.py"), "6"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "bad 0 of" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("fuzz_factorize.py", ["30", "7"]), ("fuzz_onepass.py", ["40", "5"]), ("fuzz_operators.py", ["30", "3"])])
def test_fuzz_random_shapes(script, args):
    """tools/fuzz_*.py on the GPU: random (n, m) through the tangent-setup kernels (Gram plain / weighted / leading columns, all three rmul
    kernels, lfpsqp_factorize against numpy: 5e-12), through the one-pass kernels against the two-pass ones, and through the one-pass projected CG
    with tridiagonal / diagonal + low-rank Hessians against the callback path (+ the Gram pass with extra right-hand columns against numpy)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
