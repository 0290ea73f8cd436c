#!/bin/bash
# F (live HIP events) as a function of the untimed warm-up length, with and without rocprofv3 attached -- one box, one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:34s} {d['value']:7.1f} it/s  F {d['roofline']['avg_launch_ms']:.4f} ms  gemv_t {d['matvec']['gemv_t']['ms']:.3f}  gemv_n {d['matvec']['gemv_n']['ms']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for pw in 0 2 8 0 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prewarm-seconds $pw > gpurun_out/pw.json 2>/dev/null; show "plain, prewarm $pw s" gpurun_out/pw.json
done
cd /tmp
for pw in 0 2 0 2; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pwprof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prewarm-seconds $pw > $R/gpurun_out/pw.json 2>/dev/null
  show "rocprofv3 --kernel-trace, prewarm $pw s" $R/gpurun_out/pw.json
  f=$(find $R/gpurun_out/pwprof -name "*kernel_stats.csv" | head -1); grep "PcgFuseE<false, false>" "$f" | awk -F'","' '{printf "      rocprof: %s calls avg %.4f ms\n", $2, $4/1e6}'
  rm -rf $R/gpurun_out/pwprof
done
cd $R
for pw in 0 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prewarm-seconds $pw > gpurun_out/pw.json 2>/dev/null; show "plain again, prewarm $pw s" gpurun_out/pw.json
done
