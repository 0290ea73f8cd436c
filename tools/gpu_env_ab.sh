# A/B of one environment switch of the library in interleaved fresh bench processes:  bash tools/gpu_env_ab.sh VAR valueA valueB [reps]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; V=$1; A=$2; B=$3; R=${4:-2}; O=gpurun_out/env_ab_$V.txt; : > $O
q() { env $V=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d.get('extras',{}); k=d['kernels']
print('$V=$1', 'it/s %.1f' % d['value'], 'F %.4f' % k['F_fused_projection']['ms'], 'K1 %.4f' % k['K1_dir_dAd']['ms'], 'pcg_iter %.4f' % e.get('pcg_iter_ms',0), 'nr_step %.4f' % e.get('nr_step_ms',0), 'copy %.0f triad %.0f' % (e['stream_rates']['copy_GBs'], e['stream_rates']['triad_GBs']), 'fact %.3f' % e.get('factorize_factored_basis_ms',0))"; }
for r in $(seq $R); do q $A >> $O; q $B >> $O; done
cat $O
