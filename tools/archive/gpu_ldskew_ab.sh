#!/bin/bash
# bench line (placement grid, no extras) and the tangent setup with the leading dimension of every matrix skewed by LFPSQP_LD_SKEW rows,
# interleaved:  gpurun -- bash tools/gpu_ldskew_ab.sh [reps] [skews...]
reps=${1:-3}; shift; skews=${@:-0 16 48}
mkdir -p gpurun_out/ldskew
for rep in $(seq 1 $reps); do
  for sk in $skews; do
    LFPSQP_LD_SKEW=$sk python bench.py --no-cpu-baseline --no-extras --steps 40 > gpurun_out/ldskew/b.json 2> gpurun_out/ldskew/b.err
    python - $sk $rep <<'PY'
import json, sys
sk, rep = sys.argv[1], sys.argv[2]
try:
    o = json.loads(open("gpurun_out/ldskew/b.json").read().strip().splitlines()[-1])
    g = o["config"]["basis_placement"]["trial_grid_F_ms"]
    flat = [v for r in g for v in r]
    print(f"skew {sk:>4} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.3f} ms ({o['roofline']['frac']:.3f})  gemv_t {o['matvec']['gemv_t']['ms']:.3f} gemv_n {o['matvec']['gemv_n']['ms']:.3f}  grid min/max {min(flat):.3f}/{max(flat):.3f}")
except Exception as e:
    print(sk, rep, "FAILED", e, open("gpurun_out/ldskew/b.err").read()[-300:])
PY
    echo "        factorize: $(LFPSQP_LD_SKEW=$sk python tools/time_factorize.py 1e7 128 2>&1 | tail -2 | head -1)"
  done
done | tee -a gpurun_out/ldskew_ab.txt
