#!/bin/bash
# GEMV-N (y = alpha M t + beta y) through the one-pass kernel's persistent grid and staged stores (main) against the two-pass form
# (gemvn2 = -DLFPSQP_GEMVN_ONEPASS=0), same buffers, interleaved
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/gemvn_ab.txt
import os, sys, math
sys.path.insert(0, os.getcwd())
import lfpsqp_jl_amd as L
from lfpsqp_jl_amd import _capi
libB = L.load_library("lfpsqp.jl_amd/lib/variants/liblfpsqp_gemvn2.so")
ctxA, ctxB = L.Context(0), L.Context(0, libB)
for n, m in ((10_000_000, 128), (10_000_000, 64), (5_000_000, 512), (10_000_000, 32)):
    res = {"one-pass": [], "two-pass": []}
    Ms = [ctxA.matrix(n, m).hash_fill(1, 0, n, 1e-3) for _ in range(2)]
    t = ctxA.vector(m).hash_fill(6)
    ys = [ctxA.vector(n).hash_fill(7) for _ in range(2)]
    for rnd in range(3):
        for M in Ms:
            for y in ys:
                for tag, ctx in (("one-pass", ctxA), ("two-pass", ctxB)) if rnd % 2 == 0 else (("two-pass", ctxB), ("one-pass", ctxA)):
                    def call():
                        ctx.check(ctx.L.lfpsqp_gemv_n(ctx.h, M.h, m, 1.0, t.h, 0.5, y.h))
                    call(); ctx.sync()
                    ctx.timer_begin()
                    for _ in range(6): call()
                    res[tag].append(ctx.timer_end() / 6)
    a, b = res["one-pass"], res["two-pass"]
    by = (8.0 * n * m + 16.0 * n) / 1e9
    print(f"n={n} m={m}: one-pass min/mean {min(a):.4f}/{sum(a)/len(a):.4f} ms ({by/min(a)*1e3/8000:.3f} of peak)   two-pass {min(b):.4f}/{sum(b)/len(b):.4f} ms ({by/min(b)*1e3/8000:.3f})")
    for M in Ms: M.free()
    for y in ys: y.free()
PY
python -m pytest tests/test_capi_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "gemv or linear or orthonormal" 2>&1 | tail -2
