import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "liblfpsqp_emu.so")

# The oracle allocates its work arrays uninitialised where the reference does (`Array{Float64}(undef, ...)`).  In a long test process
# np.empty hands back recycled memory, so a read before the first write would make a test depend on what ran before it (one such read --
# beta * y with beta = 0 on an uninitialised y -- once sent the oracle into an endless Armijo loop in the middle of a suite run).  Under
# the tests those arrays are NaN-filled instead, which turns any such read into a deterministic failure (oracle/lfpsqp_ref.py::_uninit).
os.environ.setdefault("ORACLE_POISON_UNINIT", "1")


def _usable_cpus() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a GPU box can show 256 hardware threads
    to a container that is allowed 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError, IndexError):
        pass
    return n


# The oracle is numpy on OpenBLAS, which sizes its thread pool by the hardware threads it sees: on a quota-limited box that
# oversubscribes the allowed CPUs many times over and the (CPU-side) checker crawls.  Child processes inherit the variables;
# the pool of this process, if numpy is loaded already, is limited in pytest_sessionstart.
_NT = str(max(1, min(_usable_cpus(), 8)))
for _var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, _NT)
# the CPU HIP emulator runs the workgroups of a launch on a few OS threads (4 unless told otherwise): use the CPUs this box allows
os.environ.setdefault("HIPEMU_THREADS", _NT)
_BLAS_LIMIT = None


def pytest_sessionstart(session):
    global _BLAS_LIMIT
    try:
        from threadpoolctl import threadpool_limits
        _BLAS_LIMIT = threadpool_limits(limits=int(_NT), user_api="blas")
    except Exception:       # threadpoolctl missing: the environment variables above still cover a fresh numpy import
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The product's C-ABI sources compiled against the CPU HIP emulator (tests/emu) --
    test infrastructure only; the product never loads it."""
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "emu")])
    import lfpsqp_jl_amd as L
    return L.load_library(EMU_LIB)


@pytest.fixture(scope="session")
def gpu_lib():
    import lfpsqp_jl_amd as L
    import __graft_entry__ as entry
    if not os.path.exists(entry.LIB):      # a checkout without built artefacts: compile it here (hipcc is in the image)
        entry.build()
    return L.load_library()   # raises loudly if the HIP extension is not built


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev_ctx(request):
    import lfpsqp_jl_amd as L
    if request.param == "emu":
        lib = request.getfixturevalue("emu_lib")
    else:
        lib = request.getfixturevalue("gpu_lib")
    ctx = L.Context(0, lib)
    if request.param == "gpu":
        assert "emulator" not in ctx.device_name
    yield ctx
    ctx.close()
