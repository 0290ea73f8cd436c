cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1500 python tools/run_config.py 4 --pp > gpurun_out/pp4.log 2>&1; tail -5 gpurun_out/pp4.log
cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pp4 -- python3 $R/tools/run_config.py 4 --pp > $R/gpurun_out/prof_pp4.log 2>&1
cd $R; python - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_pp4/**/*kernel_stats.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel s", tot/1e9)
for r in rows[:22]:
    print(r["Calls"], round(float(r["TotalDurationNs"])/1e6,1),"ms", round(float(r["AverageNs"])/1e3,1),"us", r["Percentage"], r["Name"][:90])
PY
rm -rf gpurun_out/prof_pp4
