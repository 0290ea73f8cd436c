# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only) of the kernels a timing tool launches:
#   bash tools/gpu_pmc_tool.sh <tag> <tool.py> [tool args...]      -> gpurun_out/<tag>/traffic.txt  (per launch, corrections of tools/pmc_summary.py)
cd $GRAFT_REPO_ROOT; TAG=$1; shift; TOOL=$1; shift
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/$TOOL "$@" > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/$TOOL "$@" > $O/write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json > /dev/null 2>&1
python - $TAG <<'PY' | tee gpurun_out/$TAG/traffic.txt
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}/pmc_summary.json"))["kernels"]
rows = sorted(d.items(), key=lambda kv: -kv[1]["traffic_GB"] * kv[1]["launches"])[:12]
for k, v in rows:
    print(f"{v['launches']:5d} launches  fetch {v['fetch_GB']:8.4f} GB  write {v['write_GB']:8.4f} GB  total {v['traffic_GB']:8.4f} GB  {k[:110]}")
PY
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/pmc_fetch $O/pmc_write
