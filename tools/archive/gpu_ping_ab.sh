#!/bin/bash
# A/B of the alternating residual buffers of the fused projected-CG kernel (LFPSQP_GPING=1; default 0: one buffer), interleaved
mkdir -p gpurun_out/ping
for rep in 1 2 3; do
  for ping in 1 0; do
    LFPSQP_GPING=$ping python bench.py --no-cpu-baseline --no-extras --steps 40 $1 > gpurun_out/ping/p${ping}_$rep.json 2> gpurun_out/ping/p${ping}_$rep.err
    python - $ping $rep <<'PY'
import json, sys
p, rep = sys.argv[1], sys.argv[2]
o = json.load(open(f"gpurun_out/ping/p{p}_{rep}.json"))
print(f"ping={p} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.3f} ms ({o['roofline']['frac']:.3f})  gemv_t {o['matvec']['gemv_t']['ms']:.3f} gemv_n {o['matvec']['gemv_n']['ms']:.3f}  x_norm {o['check']['x_norm']:.15e}")
PY
  done
done
