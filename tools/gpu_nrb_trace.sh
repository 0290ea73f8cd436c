# kernel timeline of the BATCHED Newton-retraction loop (lfpsqp_retract_nr_batch inside bench.py's extras): durations and gaps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_nrb -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --prewarm-seconds 0 > $R/gpurun_out/trace_nrb.log 2>&1
cd $R; python - <<'PY' | tee gpurun_out/nrb_trace.txt
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/trace_nrb/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "NRStepBatchRow" in r["Kernel_Name"]]
seq = rows[idx[2]:idx[-2] + 1]          # steady state of the batched loop
def short(n):
    n = n.replace("void lfpsqp::", "").replace("lfpsqp::", "")
    return n[:60]
dur = collections.defaultdict(list); gaps = []
for a, b in zip(seq, seq[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
steps = len(idx) - 4
span = (int(seq[-1]["Start_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
print(f"{steps} batched steps, {span/steps:.1f} us per step; kernels per step: {len(seq)/steps:.1f}; gaps per step {sum(gaps)/steps/1e3:.1f} us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:60s} n/step {len(v)/steps:4.1f}  avg {sum(v)/len(v)/1e3:9.2f} us  per step {sum(v)/steps/1e3:9.2f} us")
PY
rm -rf gpurun_out/trace_nrb
