# kernel times of the tridiagonal one-pass iteration at (1e7, 128): rocprofv3 --kernel-trace --stats over tools/time_tridiag.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05n; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05n/prof -- python3 $R/tools/time_tridiag.py 1e7 128 > $R/gpurun_out/r05n/run.txt 2>&1
cd $R; tail -4 gpurun_out/r05n/run.txt | cut -c1-200
python - <<'PY' | tee gpurun_out/r05n/tridiag_kernel_stats.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/r05n/prof/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
rm -rf gpurun_out/r05n/prof
