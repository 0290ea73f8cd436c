#!/bin/bash
# staged stores of the fused projected-CG kernel: variants against the unstaged build on the same buffers
#   gpurun -- bash tools/gpu_stage_ab.sh
mkdir -p gpurun_out
{
AB_ROWS=5e6 AB_COLS=512 python tools/ab_same_buffers.py nowide main 3 2
AB_ROWS=5e6 AB_COLS=384 python tools/ab_same_buffers.py nowide main 2 2
AB_ROWS=1e7 AB_COLS=256 python tools/ab_same_buffers.py nowide main 2 2
timeout 900 python tools/fuzz_onepass.py 2>&1 | tail -2
python -m pytest tests/test_staged_stores.py -m gpu -x -q 2>&1 | tail -2
} > gpurun_out/stage_ab_wide.txt 2>&1
tail -50 gpurun_out/stage_ab_wide.txt
