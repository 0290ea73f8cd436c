# kernel statistics of config 4 with the DEFAULT (projection-penalty) retraction, n = 1e6 (a few outer iterations, time-boxed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pp4 -- python3 $R/tools/run_config.py 4 1e6 128 --pp --max-outer=4 > $R/gpurun_out/prof_pp4.log 2>&1
cd $R; tail -4 gpurun_out/prof_pp4.log | cut -c1-150
python - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_pp4/**/*kernel_stats.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel s", round(tot/1e9,2))
for r in rows[:12]:
    print(r["Calls"], round(float(r["TotalDurationNs"])/1e9,2),"s", round(float(r["AverageNs"])/1e3,1),"us", r["Percentage"], r["Name"][:110])
PY
rm -rf gpurun_out/prof_pp4
