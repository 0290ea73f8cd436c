cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_capi_retractions.py -m gpu -q -x 2>&1 | tail -4
for cw in 4 2; do
  LFPSQP_NR_ONEPASS=$cw python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/nr1_$cw.json 2> gpurun_out/nr1_$cw.err; tail -2 gpurun_out/nr1_$cw.err
  python -c "
import json; d=json.loads(open('gpurun_out/nr1_$cw.json').read().strip().splitlines()[-1]); e=d['extras']; print('cw=$cw', {k:e[k] for k in e if k.startswith('nr_')})"
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_nr1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace_nr1.log 2>&1
cd $R; python - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/trace_nr1/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("gemv_nt","nr_small","nr_onepass")): print(r["Calls"], float(r["AverageNs"])/1e3, n[:80])
PY
rm -rf gpurun_out/trace_nr1
