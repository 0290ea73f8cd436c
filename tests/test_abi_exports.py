"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads
without a GPU, exports every function include/lfpsqp_hip.h declares, and refuses to run
(loudly) when no device is visible -- there is no CPU fallback in the product."""
import ctypes
import os

import pytest

import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return L.load_library()


def test_every_declared_symbol_is_exported(built):
    names = L.header_functions()
    assert len(names) >= 35
    for name in names:
        assert hasattr(built.lib, name), f"{name} declared in include/lfpsqp_hip.h but not exported"
    # and the ctypes table binds exactly the declared surface
    assert set(_capi._SIGS) | {"lfpsqp_last_error", "lfpsqp_vec_len", "lfpsqp_half_stride"} == set(names)


def test_code_object_targets_gfx950(built):
    blob = open(built.path, "rb").read()
    assert b"gfx950" in blob


def test_shard_range_partitions_rows(built):
    for n in (0, 1, 1023, 1024, 10_000_000, 40_000_000 + 7):
        for nranks in (1, 2, 4, 8):
            prev = 0
            for r in range(nranks):
                r0, r1 = ctypes.c_int64(), ctypes.c_int64()
                assert built.lfpsqp_shard_range(n, r, nranks, ctypes.byref(r0), ctypes.byref(r1)) == 0
                assert r0.value == prev and r1.value >= r0.value
                assert r0.value % 2048 == 0 or r0.value == n      # shard boundaries on whole tiles
                prev = r1.value
            assert prev == n
    r0, r1 = ctypes.c_int64(), ctypes.c_int64()
    assert built.lfpsqp_shard_range(10, 3, 2, ctypes.byref(r0), ctypes.byref(r1)) < 0


def test_product_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(L.LfpsqpError):
        L.Context(0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(L.LfpsqpError):
        L.load_library(str(tmp_path / "nope.so"))
