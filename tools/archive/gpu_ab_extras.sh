#!/bin/bash
# A/B of library variants on one box with the bench's extra timings (Newton step, pcg! iteration): each variant twice, interleaved.
#   gpurun -- bash tools/gpu_ab_extras.sh tagA tagB ...     ("main" = the product library)
mkdir -p gpurun_out/abx
for rep in 1 2; do
  for tag in "$@"; do
    lib=""; [ "$tag" != "main" ] && lib="--lib lfpsqp.jl_amd/lib/variants/liblfpsqp_$tag.so"
    python bench.py --no-cpu-baseline --steps 40 $lib > gpurun_out/abx/${tag}_$rep.json 2> gpurun_out/abx/${tag}_$rep.err
    python - "$tag" $rep <<'PY'
import json, sys
tag, rep = sys.argv[1], sys.argv[2]
try:
    o = json.load(open(f"gpurun_out/abx/{tag}_{rep}.json"))
    e = o["extras"] if "extras" in o else o
    def f(k):
        for d in (o, o.get("extras", {}), o.get("retraction", {}), o.get("kernels", {})):
            if isinstance(d, dict) and k in d: return d[k]
        return float("nan")
    print(f"{tag:6s} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.3f} ({o['roofline']['frac']:.3f})  nr_step {f('nr_step_ms'):.3f}  two_streams {f('nr_step_two_streams_ms'):.3f}  batch4 {f('nr_batch4_step_ms'):.3f}  pcg_iter {f('pcg_iter_ms'):.3f}  x_norm {o['check']['x_norm']:.15e}")
except Exception as ex:
    print(tag, rep, "FAILED", ex, open(f"gpurun_out/abx/{tag}_{rep}.err").read()[-400:])
PY
  done
done
