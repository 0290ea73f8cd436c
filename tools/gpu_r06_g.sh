# round 6: rocprofv3 kernel statistics of the tangent step (tools/time_tangent.py) and of the exact batch (tools/time_nrbatch.py, bounds), final library;
# timeline of one outer iteration (streamed class); steady outer iteration with bounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g.txt; : > $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06g_tan -- python3 $R/tools/time_tangent.py > $R/gpurun_out/r06g_tan.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06g_nrb -- python3 $R/tools/time_nrbatch.py 1e7 128 --bounds 1 --nbs 2,4 --iters 40 > $R/gpurun_out/r06g_nrb.log 2>&1
cd $R
grep "n=" gpurun_out/r06g_tan.log | tee -a $O
for d in r06g_tan r06g_nrb; do
  f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1)
  echo "== rocprofv3 --kernel-trace --stats: $d" | tee -a $O
  python - "$f" <<'PY' | tee -a $O
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{int(r['Calls']):6d} calls  avg {float(r['AverageNs']) / 1e6:8.4f} ms  min {float(r['MinNs']) / 1e6:8.4f}  {r['Name'].replace('void lfpsqp::', '').replace('lfpsqp::', '')[:120]}")
PY
  find gpurun_out/$d -name "*kernel_trace.csv" -delete
done
grep "nb=" gpurun_out/r06g_nrb.log | tee -a $O
bash tools/gpu_outer_trace.sh stream 2>&1 | tail -34 | tee -a $O
timeout 600 python tools/time_outer_bounds.py 2>&1 | tail -3 | tee -a $O
rm -rf gpurun_out/r06g_tan gpurun_out/r06g_nrb
