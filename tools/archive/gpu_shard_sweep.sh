# per-GPU shard sizes of the strong-scaling runs (n=1e7 over 2/4/8 ranks) on one GPU: where does the iteration time go?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rows in 10000000 5000000 2500000 1250000; do
  python bench.py --rows $rows --steps 200 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/sweep_$rows.json 2>gpurun_out/sweep_$rows.err
  python -c "
import json; d=json.loads(open('gpurun_out/sweep_$rows.json').read().strip().splitlines()[-1]); print($rows, round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v['ms'],4) if isinstance(v,dict) else round(v,2)) for k,v in d['kernels'].items()})"
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_small -- python3 $R/bench.py --rows 1250000 --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/trace_small.log 2>&1
cd $R; python tools/trace_seq.py gpurun_out/trace_small PcgFuseE 300; rm -rf gpurun_out/trace_small
