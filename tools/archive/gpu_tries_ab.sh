#!/bin/bash
# candidate allocations per placed buffer: 3 (default) against 5, fresh processes, interleaved
mkdir -p gpurun_out
for rep in 1 2 3 4; do for tr in 3 5; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --placement-tries $tr 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config']['placement']
print('tries $tr rep $rep:', round(d['value'],1), 'it/s  F', round(d['roofline']['avg_launch_ms'],4), ' trials min/median/max', p['probe_min_median_max_F_ms'], ' first', round(d['first_allocation']['F_ms'],3))"
done; done 2>&1 | tee gpurun_out/tries_ab.txt
