# The library's placement policy against plain first allocations and against the round-2 Python grid, interleaved fresh processes on one box.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pp
REPS=${1:-3}
for rep in $(seq 1 $REPS); do
  for mode in library first grid; do
    python bench.py --placement $mode $EXTRA --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/pp/${mode}_$rep.json 2> gpurun_out/pp/${mode}_$rep.err
    python - $mode $rep <<'PY'
import json, sys
mode, rep = sys.argv[1], sys.argv[2]
try:
    o = json.load(open(f"gpurun_out/pp/{mode}_{rep}.json"))
    pl = o["config"]["placement"]
    fa = o.get("first_allocation", {})
    extra = ""
    if mode == "library":
        extra = f" pair probes {pl['probe_F_ms']} kept {pl['kept_pair']}"
    if mode == "grid":
        extra = f" grid min/median/max {pl['grid_min_median_max_F_ms']}"
    print(f"{mode:8s} rep{rep}: {o['value']:7.1f} it/s  F {o['roofline']['avg_launch_ms']:.4f} ms ({o['roofline']['frac']:.3f})  single_call {o['single_call']['value']:.1f}  first_alloc F {fa.get('F_ms', float('nan')):.4f}{extra}")
except Exception as e:
    print(mode, rep, "FAILED", e, open(f"gpurun_out/pp/{mode}_{rep}.err").read()[-600:])
PY
  done
done
