# kernel statistics of BASELINE config 4 at full size (batched Newton retractions)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/tools/run_config.py 4 > $R/gpurun_out/prof_c4.log 2>&1
cd $R; tail -2 gpurun_out/prof_c4.log
python - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_c4/**/*kernel_stats.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel s", round(tot/1e9,2))
for r in rows[:10]:
    print(r["Calls"], round(float(r["TotalDurationNs"])/1e9,2),"s", round(float(r["AverageNs"])/1e3,1),"us", r["Percentage"], r["Name"][:100])
PY
rm -rf gpurun_out/prof_c4
