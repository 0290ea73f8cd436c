#!/bin/bash
# staged stores of the fused projected-CG kernel: variants against the unstaged build on the same buffers
#   gpurun -- bash tools/gpu_stage_ab.sh
mkdir -p gpurun_out
{
python tools/ab_same_buffers.py base main 3 2
AB_COLS=120 python tools/ab_same_buffers.py base main 2 2
AB_COLS=96 python tools/ab_same_buffers.py base c8 2 2
AB_COLS=64 python tools/ab_same_buffers.py base c8 2 2
AB_COLS=32 python tools/ab_same_buffers.py base c8 2 2
timeout 900 python tools/fuzz_onepass.py 2>&1 | tail -2
} > gpurun_out/stage_ab.txt 2>&1
tail -50 gpurun_out/stage_ab.txt
