# kernel timeline of ONE steady outer iteration of `optimize` on the nonlinear class (tools/time_elementwise.py <mode>): the launches between the last two
# Gram kernels, with durations and gaps -- which passes over the matrix an outer iteration is made of
#   bash tools/gpu_outer_trace.sh [stream|dense|nostream]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; M=${1:-stream}
cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/otrace -- python3 $R/tools/time_elementwise.py $M > /dev/null 2> $R/gpurun_out/otrace.err
cd $R; python - $M <<'PY' | tee gpurun_out/outer_trace_$1.txt
import csv, glob, sys
f = sorted(glob.glob("gpurun_out/otrace/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
g = [i for i, r in enumerate(rows) if "gram_kernel" in r["Kernel_Name"]]
# the last optimize run is the ProjPenalty one; take the Newton runs' steady iteration: the pair of Gram launches 9 and 8 factorisations before the end of the
# first half is fragile -- simply take the last two Gram launches (ProjPenalty run, zero retraction iterations in steady state: same passes + jac!)
a, b = g[-2], g[-1]
seq = rows[a:b]
def short(n):
    n = n.replace("void lfpsqp::", "").replace("lfpsqp::", "")
    return n[:110]
tot = 0.0
print(f"time_elementwise.py {sys.argv[1]}: one steady outer iteration = {len(seq)} launches, {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")
prev_end = None
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    d = (e - s) / 1e3
    tot += d
    if d > 20 or gap > 50:
        print(f"  {d:9.1f} us  (gap before {gap:7.1f})  {short(r['Kernel_Name'])}")
    prev_end = e
print(f"kernel time {tot:.1f} us")
PY
rm -rf gpurun_out/otrace
