cd $GRAFT_REPO_ROOT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "
import sys; sys.path.insert(0,'.')
from oracle import port; print('usable cpus', port.usable_cpus())"
for rep in 1 2; do
for v in base nt ks4 ks1 tc8 tc2 ntks4; do
  python bench.py --steps 20 --warmup 2 --no-cpu-baseline --basis scaled-hash --lib tools/variants/lib_$v.so 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']; mv = d['matvec']
print('$v', 'it/s %.1f' % d['value'], 'K1 %.3f K2 %.3f K3 %.3f ms' % (k['K1_dir_dAd']['ms'], k['K2_step_UTrp']['ms'], k['K3_proj_dots']['ms']), 'gemv_t %.3f gemv_n %.3f ms' % (mv['gemv_t']['ms'], mv['gemv_n']['ms']))"
done; done
