"""Generic kernel timeline from a rocprofv3 --kernel-trace CSV: per-kernel-name durations and gaps between consecutive kernels
inside the window of the last `count` launches before the final occurrence of the kernel named argv[2]."""
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
key = sys.argv[2]; count = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"lfpsqp::", "", n)
    m = re.match(r"(?:void )?(\w+)(?:<([\w:]+))?", n)
    return (m.group(1) + ("<" + m.group(2) + ">" if m.group(2) else ""))[:40] if m else n[:40]
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
last = idx[-1]; seq = rows[max(0, last - count):last + 1]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(seq, seq[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
nkey = sum(1 for r in seq if key in r["Kernel_Name"])
span = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e6
print(f"window: {len(seq)} kernels, {nkey} x {key}, span {span:.3f} ms -> {span / max(nkey - 1, 1):.4f} ms per occurrence")
print("durations (us):"); [print(f"  {k:42s} n={len(v):4d} avg {sum(v)/len(v)/1e3:9.2f}") for k, v in dur.items()]
print("gaps (us):"); [print(f"  {k:84s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.2f}") for k, v in gap.items() if len(v) > 2]
