#!/bin/bash
# one gpurun call: build, smoke, GPU parity tests, bench, rocprof kernel trace
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx9" | head -4
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 900 python bench.py --steps 30 --warmup 3 --basis scaled-hash > gpurun_out/bench_r1a.json 2> gpurun_out/bench_r1a.err
tail -3 gpurun_out/bench_r1a.err; cat gpurun_out/bench_r1a.json
# load-order check: torch first, then our library
python -c "
import torch, __graft_entry__ as g
g.smoke()
print('torch-first OK', torch.__version__)" 2>&1 | tail -3
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1a -- python3 $R/bench.py --steps 10 --warmup 2 --basis scaled-hash --no-cpu-baseline > $R/gpurun_out/prof_r1a.log 2>&1
cd $R; tail -2 gpurun_out/prof_r1a.log
find gpurun_out/prof_r1a -name "*stats*" | head; 
f=$(find gpurun_out/prof_r1a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
