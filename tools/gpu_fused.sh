cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err; tail -2 gpurun_out/bench_fused.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_fused.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['roofline']); print(d['kernels']); print(d['check']); print(d['cpu_baseline']['value'])"
LFPSQP_ONEPASS=-1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench_2pass.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_2pass.json').read().strip().splitlines()[-1])
print('two-pass', d['value'], d['ms_per_step'], d['check'])"
