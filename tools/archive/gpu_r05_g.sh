# where does config 4 with the reference's DEFAULT retraction (ProjPenalty) + exact preconditioner spend its time?  two outer iterations under the kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05g/prof -- python3 $R/tools/run_config.py 4 --pp --precond --max-outer=2 > $R/gpurun_out/r05g/run.txt 2>&1
cd $R; tail -12 gpurun_out/r05g/run.txt
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/r05g/prof/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e9:.2f} s")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{float(r['TotalDurationNs'])/1e9:8.3f} s  {int(r['Calls']):7d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
PY
rm -rf gpurun_out/r05g/prof
