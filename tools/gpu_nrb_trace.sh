# kernel timeline of the BATCHED Newton-retraction loop (tools/time_nrbatch.py, 16 trials on the matrix cores): durations and gaps
#   bash tools/gpu_nrb_trace.sh [bounds 0|1] [nb]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=${1:-1}; NB=${2:-16}
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_nrb -- python3 $R/tools/time_nrbatch.py 1e7 128 --bounds $B --nbs $NB --iters 30 > $R/gpurun_out/trace_nrb.log 2>&1
cd $R; tail -2 gpurun_out/trace_nrb.log; python - <<'PY' | tee gpurun_out/nrb_trace.txt
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/trace_nrb/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nrb_mfma_kernel" in r["Kernel_Name"] or "NRStepBatchRow" in r["Kernel_Name"]]
# steady state of the batched loop: the middle third of the step kernels found (whatever --iters / --nbs produced them; the first launches
# are the warm-up repetition, the last ones run with finished trials)
if len(idx) < 9:
    raise SystemExit(f"only {len(idx)} batched step kernels in the trace: nothing to average")
lo, hi = len(idx) // 3, 2 * len(idx) // 3
seq = rows[idx[lo]:idx[hi] + 1]
def short(n):
    n = n.replace("void lfpsqp::", "").replace("lfpsqp::", "")
    return n[:60]
dur = collections.defaultdict(list); gaps = []
for a, b in zip(seq, seq[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
steps = hi - lo
span = (int(seq[-1]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
print(f"{steps} batched steps, {span/steps:.1f} us per step; kernels per step: {len(seq)/steps:.1f}; gaps per step {sum(gaps)/steps/1e3:.1f} us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:60s} n/step {len(v)/steps:4.1f}  avg {sum(v)/len(v)/1e3:9.2f} us  per step {sum(v)/steps/1e3:9.2f} us")
PY
rm -rf gpurun_out/trace_nrb
