#!/bin/bash
# kernel times of SpMV-T / SpMV-N at (1e7, 128, K = 4) under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- python3 tools/time_spmv.py > /tmp/sp.log 2>&1
tail -3 /tmp/sp.log
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
