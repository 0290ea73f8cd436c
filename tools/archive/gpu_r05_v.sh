# HBM traffic of the tridiagonal one-pass iteration's kernels at (1e7, 128): FETCH_SIZE / WRITE_SIZE passes over tools/time_tridiag.py
# (separate --pmc passes with --kernel-trace only; corrections as tools/pmc_summary.py states them)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05v; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/time_tridiag.py 1e7 128 > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/time_tridiag.py 1e7 128 > $O/write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json > /dev/null 2>&1
python - <<'PY' | tee gpurun_out/r05v/tridiag_traffic.txt
import json
d = json.load(open("gpurun_out/r05v/pmc_summary.json"))["kernels"]
for k, v in d.items():
    if any(t in k for t in ("PcgFuseTri<false>", "TriPrepF<false>", "PcgFuseE<false, false>", "PcgDirG", "TriMulF", "gram_kernel")):
        print(f"{v['launches']:5d} launches  fetch {v['fetch_GB']:8.4f} GB  write {v['write_GB']:8.4f} GB  total {v['traffic_GB']:8.4f} GB  {k[:100]}")
PY
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/pmc_fetch $O/pmc_write
