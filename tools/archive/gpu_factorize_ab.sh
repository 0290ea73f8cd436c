#!/bin/bash
# tangent-setup timings (tools/time_factorize.py) of the product library and of variants, twice, interleaved, in one call:
#   gpurun -- bash tools/gpu_factorize_ab.sh main tagA ...
mkdir -p gpurun_out
for rep in 1 2; do
  for tag in "$@"; do
    lib="lfpsqp.jl_amd/lib/liblfpsqp_hip.so"; [ "$tag" != "main" ] && lib="lfpsqp.jl_amd/lib/variants/liblfpsqp_$tag.so"
    for shape in "1e7 128" "5e6 512" "1e7 129"; do
      echo "$tag rep$rep: $(LFPSQP_LIB=$lib python tools/time_factorize.py $shape 2>&1 | tail -2 | tr '\n' ' ')"
    done
  done
done | tee gpurun_out/factorize_ab.txt
