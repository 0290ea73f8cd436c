#!/bin/bash
# Does the COLUMN STRIDE of the basis decide the two speeds of the fused kernel?  The product rounds the leading dimension up to 2048 rows, so every
# column starts on the same 16 KB phase; LFPSQP_LD_SKEW (rows, even) replaces the default of 16.
#   gpurun -- bash tools/gpu_ldskew_probe.sh

for rep in 1 2; do
  for sk in 0 16 32 96 544 1040; do
    echo "== rep $rep: leading dimension + $sk rows"; LFPSQP_LD_SKEW=$sk timeout 150 python tools/placement_matrix_probe.py 3 4 2>&1 | grep "^round 1"
  done
done | tee gpurun_out/ldskew_probe.txt
