#!/bin/bash
# streaming stores in K1 (the vector kernel between two launches of the fused kernel): does its write-back stop falling into F's read stream?
mkdir -p gpurun_out
{
python tools/ab_same_buffers.py main k1nt1 3 2
python tools/ab_same_buffers.py main k1nt2 3 2
} > gpurun_out/k1nt_ab.txt 2>&1
cat gpurun_out/k1nt_ab.txt
