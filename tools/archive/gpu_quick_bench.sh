cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/qb.json 2> gpurun_out/qb.err; tail -2 gpurun_out/qb.err
python -c "
import json; d=json.loads(open('gpurun_out/qb.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v['ms'],4) if isinstance(v,dict) else v) for k,v in d['kernels'].items()}, d['check']['nr'], {k:round(v,3) for k,v in d['extras'].items() if k.startswith('nr_step')})"
done
