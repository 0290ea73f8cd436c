"""Kernel timeline of the projcg loop from a rocprofv3 --kernel-trace CSV: durations and inter-kernel gaps."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("PcgDirF", "PcgStepV", "PcgProjE", "PcgPost1", "PcgPost3", "NoPost", "InitState", "FlushXF", "ResidualV"):
        if k in n: return ("reduce<%s>" % k) if "reduce_rows" in n else k
    return n[:30]
# take the LAST projcg call: find last InitState
idx = max(i for i, r in enumerate(rows) if "InitState" in r["Kernel_Name"])
seq = rows[idx:idx + 400]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(seq, seq[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
print("durations (us):"); [print(f"  {k:24s} n={len(v):3d} avg {sum(v)/len(v)/1e3:8.2f}") for k, v in dur.items()]
print("gaps (us):"); [print(f"  {k:44s} n={len(v):3d} avg {sum(v)/len(v)/1e3:8.2f}") for k, v in gap.items() if len(v) > 3]
