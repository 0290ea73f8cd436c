cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cw in 8 4 8 4; do
  LFPSQP_NR_ONEPASS=$cw python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/nr1_$cw.json 2> gpurun_out/nr1_$cw.err; tail -2 gpurun_out/nr1_$cw.err
  python -c "
import json; d=json.loads(open('gpurun_out/nr1_$cw.json').read().strip().splitlines()[-1]); e=d['extras']; print('cw=$cw', {k:round(e[k],3) for k in e if k.startswith('nr_')})"
done
