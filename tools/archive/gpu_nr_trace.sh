cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_nr1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trace_nr1.log 2>&1
cd $R; python tools/trace_nr.py gpurun_out/trace_nr1 nr_onepass; rm -rf gpurun_out/trace_nr1
