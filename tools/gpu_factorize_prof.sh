#!/bin/bash
# kernel statistics of the tangent setup (tools/time_factorize.py N M) -> gpurun_out/factorize_prof_<N>_<M>.txt
N=${1:-1e7}; M=${2:-128}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fprof_${N}_${M}; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/time_factorize.py $N $M > $O/run.log 2>&1
cd $R
f=$(find $O -name "*kernel_stats.csv" | head -1)
{ tail -2 $O/run.log; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY
} | tee $R/gpurun_out/factorize_prof_${N}_${M}.txt
find $O -name "*kernel_trace.csv" -delete
