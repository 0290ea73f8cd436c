cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_pcg -- python3 $R/bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/trace_pcg.log 2>&1
cd $R; python tools/trace_seq.py gpurun_out/trace_pcg PcgFuseE 200; rm -rf gpurun_out/trace_pcg
