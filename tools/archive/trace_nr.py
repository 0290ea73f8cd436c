"""Kernel timeline of the Newton-retraction loop from a rocprofv3 --kernel-trace CSV: per-kernel durations and the gaps
between consecutive kernels inside the last retraction that used the kernel named in argv[2] (default nr_onepass)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
key = sys.argv[2] if len(sys.argv) > 2 else "nr_onepass"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("nr_onepass", "nr_small", "gemv_nt", "reduce_rows", "post_kernel", "gemv_t_kernel", "vec_kernel"):
        if k in n: return k
    return n[:24]
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
last = idx[-1]; first = last
while first - 1 >= 0 and last - first < 400 and any(k in rows[first - 1]["Kernel_Name"] for k in ("nr_", "reduce_rows", "gemv_nt")): first -= 1
seq = rows[first:last + 3]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(seq, seq[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
print("kernels in window:", len(seq), " span ms:", (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e6)
print("durations (us):"); [print(f"  {k:24s} n={len(v):3d} avg {sum(v)/len(v)/1e3:9.2f}") for k, v in dur.items()]
print("gaps (us):"); [print(f"  {k:44s} n={len(v):3d} avg {sum(v)/len(v)/1e3:9.2f}  max {max(v)/1e3:9.2f}") for k, v in gap.items()]
