# kernel timeline of the timed projected-CG iterations of bench.py (per-kernel durations + gaps): gpurun_out/iter_trace.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/iter_trace; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS > $O/run.log 2>&1
cd $R
python tools/trace_seq.py $O pcg_post_kernel 110 | tee gpurun_out/iter_trace.txt
find $O -name "*kernel_trace.csv" -delete
