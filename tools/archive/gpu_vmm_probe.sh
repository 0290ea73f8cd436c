#!/bin/bash
# (ARCHIVED: the -DLFPSQP_VMM_EXPERIMENT variant of csrc/context.hip this probe needs was removed from the sources in round 5; git show a86abc9:lfpsqp.jl_amd/csrc/context.hip has it)
# Does the virtual-memory API (hipMemCreate / hipMemMap, fixed-size physical chunks) change which allocations the fused projected-CG
# kernel is slow on?  tools/placement_matrix_probe.py (3 bases x 4 work sets, F per pair) with hipMalloc and with a variant library built with
# -DLFPSQP_VMM_EXPERIMENT (csrc/context.hip), interleaved:   gpurun -- bash tools/gpu_vmm_probe.sh [reps]
mkdir -p gpurun_out
V=lfpsqp.jl_amd/lib/variants/liblfpsqp_vmm.so
P="timeout 150 python tools/placement_matrix_probe.py 3 4"
for rep in $(seq 1 ${1:-2}); do
  echo "== rep $rep: hipMalloc";                                $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: VMM, one handle per buffer";               LFPSQP_LIB=$V LFPSQP_VMM=1 $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: VMM, 2 MB chunks";                         LFPSQP_LIB=$V LFPSQP_VMM=2 $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: VMM, 64 MB chunks";                        LFPSQP_LIB=$V LFPSQP_VMM=64 $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: VMM, 64 MB chunks mapped in reverse";      LFPSQP_LIB=$V LFPSQP_VMM=64 LFPSQP_VMM_REV=1 $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: hipMalloc matrix, VMM vectors in 2 MB chunks";   LFPSQP_LIB=$V LFPSQP_VMM=2 LFPSQP_VMM_VEC_ONLY=1 $P 2>&1 | grep -v "^round 0"
  echo "== rep $rep: hipMalloc matrix, VMM vectors in 64 MB chunks";  LFPSQP_LIB=$V LFPSQP_VMM=64 LFPSQP_VMM_VEC_ONLY=1 $P 2>&1 | grep -v "^round 0"
done | tee -a gpurun_out/vmm_probe.txt
