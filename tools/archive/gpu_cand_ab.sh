#!/bin/bash
# bench line with and without the placement candidates (basis x3, work vectors x4), interleaved, N times:  gpurun -- bash tools/gpu_cand_ab.sh [N]
mkdir -p gpurun_out/cand
for rep in $(seq 1 ${1:-4}); do
  for k in "3 4" "1 1"; do
    set -- $k
    python bench.py --no-cpu-baseline --no-extras --steps 40 --basis-candidates $1 --work-candidates $2 > gpurun_out/cand/k_$rep.json 2> gpurun_out/cand/k_$rep.err
    python - "$1x$2" $rep <<'PY'
import json, sys
k, rep = sys.argv[1], sys.argv[2]
try:
    o = json.load(open(f"gpurun_out/cand/k_{rep}.json"))
    c = o["config"]
    print(f"candidates {k} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.3f} ms ({o['roofline']['frac']:.3f})  gemv_n {o['matvec']['gemv_n']['ms']:.3f}  basis trials {c['basis_placement'].get('trial_F_ms')} -> {c['basis_placement'].get('chosen')}  work trials {c['work_placement'].get('trial_F_ms')} -> {c['work_placement'].get('chosen')}")
except Exception as e:
    print(k, rep, "FAILED", e, open(f"gpurun_out/cand/k_{rep}.err").read()[-300:])
PY
  done
done | tee gpurun_out/cand_ab2.txt
