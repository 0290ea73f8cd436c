#!/bin/bash
# bench line with and without the work-placement candidates, interleaved, N times:  gpurun -- bash tools/gpu_cand_ab.sh [N]
mkdir -p gpurun_out/cand
for rep in $(seq 1 ${1:-4}); do
  for k in 4 1; do
    python bench.py --no-cpu-baseline --no-extras --steps 40 --work-candidates $k > gpurun_out/cand/k${k}_$rep.json 2> gpurun_out/cand/k${k}_$rep.err
    python - $k $rep <<'PY'
import json, sys
k, rep = sys.argv[1], sys.argv[2]
try:
    o = json.load(open(f"gpurun_out/cand/k{k}_{rep}.json"))
    print(f"candidates={k} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.3f} ms ({o['roofline']['frac']:.3f})  gemv_n {o['matvec']['gemv_n']['ms']:.3f}  trials {o['config']['work_placement'].get('trial_F_ms')} -> {o['config']['work_placement'].get('chosen')}")
except Exception as e:
    print(k, rep, "FAILED", e, open(f"gpurun_out/cand/k{k}_{rep}.err").read()[-300:])
PY
  done
done | tee gpurun_out/cand_ab.txt
