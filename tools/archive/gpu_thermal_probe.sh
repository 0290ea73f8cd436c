#!/bin/bash
# is the slow regime of the fused kernel thermal?  F before / after idling / after sustained load, with the temperatures rocm-smi reports
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
q() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --basis-candidates 2 --work-candidates 2 --prewarm-seconds ${2:-2} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['config']['basis_placement']['trial_grid_F_ms']; print('$1:', round(d['value'],1), 'it/s F', round(d['roofline']['avg_launch_ms'],3), 'gemv_t', round(d['matvec']['gemv_t']['ms'],3), 'gemv_n', round(d['matvec']['gemv_n']['ms'],3), 'grid', g)"; rocm-smi --showtemp --showpower 2>/dev/null | grep -i "temperature\|power" | tr '\n' ' ' | cut -c1-400; echo; }
{
q "start"
sleep 90
q "after 90 s idle"
sleep 180
q "after 180 s more idle"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prewarm-seconds 60 --basis-candidates 1 --work-candidates 1 > /dev/null 2>&1
q "after 60 s of sustained load"
} | tee gpurun_out/thermal_probe.txt
