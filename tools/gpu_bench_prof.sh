cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python tools/ab_variants.py 2>&1 | tail -6
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_r1c.json 2> gpurun_out/bench_r1c.err; tail -2 gpurun_out/bench_r1c.err; cat gpurun_out/bench_r1c.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1c -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r1c.log 2>&1
cd $R; f=$(find gpurun_out/prof_r1c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
