#!/bin/bash
# stacked fused projected-CG iteration: staged stores (main) against the unstaged build (base = -DLFPSQP_OP_STAGE=0), interleaved
mkdir -p gpurun_out
{
for rep in 1 2; do
  python tools/time_stacked.py 1e7 128 --lib lfpsqp.jl_amd/lib/variants/liblfpsqp_base.so | sed 's/^/base: /'
  python tools/time_stacked.py 1e7 128 | sed 's/^/main: /'
done
python -m pytest tests/test_staged_stores.py tests/test_capi_inequalities.py -m gpu -x -q 2>&1 | tail -2
} > gpurun_out/stacked_ab.txt 2>&1
cat gpurun_out/stacked_ab.txt
