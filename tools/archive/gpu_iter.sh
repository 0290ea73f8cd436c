cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_quick_bench.sh
for rows in 1250000; do
  python bench.py --rows $rows --steps 200 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/sweep_$rows.json 2>gpurun_out/sweep_$rows.err
  python -c "
import json; d=json.loads(open('gpurun_out/sweep_$rows.json').read().strip().splitlines()[-1]); print($rows, round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v['ms'],4) if isinstance(v,dict) else round(v,2)) for k,v in d['kernels'].items()})"
done
python bench.py --rows 5000000 --cols 512 --steps 30 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/c5.json 2>gpurun_out/c5.err; python -c "
import json; d=json.loads(open('gpurun_out/c5.json').read().strip().splitlines()[-1]); print('c5 shard', round(d['value'],1), round(d['ms_per_step'],4), d['check'])"
