#!/bin/bash
# kernel times of the sparse Gram at K = 4 / 12 / 16 (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for k in 4 12 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/spg_${k} -- python3 $R/tools/time_spgram.py 1e7 128 $k > /tmp/spg_$k.log 2>&1
  f=$(find /tmp/spg_${k} -name "*kernel_stats.csv" | head -1)
  echo "== K=$k"; tail -1 /tmp/spg_$k.log; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done > $R/gpurun_out/spgram_prof.txt 2>&1
cat $R/gpurun_out/spgram_prof.txt
