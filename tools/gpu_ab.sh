#!/bin/bash
# A/B of library variants (tools/build_variant.py) on one box: each variant's bench line twice, interleaved.
#   gpurun -- bash tools/gpu_ab.sh "<extra bench args>" tagA tagB ...     ("main" = the product library)
mkdir -p gpurun_out/ab
ARGS="$1"; shift
for rep in 1 2; do
  for tag in "$@"; do
    lib=""; [ "$tag" != "main" ] && lib="--lib lfpsqp.jl_amd/lib/variants/liblfpsqp_$tag.so"
    python bench.py --no-cpu-baseline --no-extras --steps 40 $ARGS $lib > gpurun_out/ab/${tag}_$rep.json 2> gpurun_out/ab/${tag}_$rep.err
    python - "$tag" $rep <<'PY'
import json, sys
tag, rep = sys.argv[1], sys.argv[2]
try:
    o = json.load(open(f"gpurun_out/ab/{tag}_{rep}.json"))
    print(f"{tag:8s} rep{rep}: {o['value']:7.1f} it/s  step {o['ms_per_step']:.3f} ms  F {o['roofline']['avg_launch_ms']:.3f} ms ({o['roofline']['frac']:.3f})  K1 {o['kernels']['K1_dir_dAd']['ms']:.3f}  gemv_t {o['matvec']['gemv_t']['ms']:.3f} gemv_n {o['matvec']['gemv_n']['ms']:.3f}  x_norm {o['check']['x_norm']:.15e}")
except Exception as e:
    print(tag, rep, "FAILED", e, open(f"gpurun_out/ab/{tag}_{rep}.err").read()[-400:])
PY
  done
done
