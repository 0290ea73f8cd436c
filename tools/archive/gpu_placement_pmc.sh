# Which hardware counter separates a slow (matrix, work-vector) allocation pair from a fast one?  
# One probe process per counter set; F dispatches attributed to pairs by launch order.  Separate --pmc passes, kernel-trace only.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_avail.txt 2>&1
python3 $R/tools/placement_pmc_probe.py 4 4 20 > $R/gpurun_out/placement_nopmc.log 2>&1
# the counters only matter on a box that shows BOTH speeds in one process (rocprofv3 --pmc also fixes the clocks: absolute times shift)
if ! python3 - $R/gpurun_out/placement_nopmc.log <<'PY'
import json, sys
p = [q["F_ms"] for line in open(sys.argv[1]) if line.startswith("PAIRS ") for q in json.loads(line[6:])["pairs"]]
print("no-pmc F ms: min %.4f max %.4f" % (min(p), max(p)))
sys.exit(0 if max(p) > 1.05 * min(p) or "FORCE" in sys.argv else 1)
PY
then echo "uniform box: no PMC passes" ; [ -z "$FORCE_PMC" ] && exit 0; fi
WISH="${WISH_OVERRIDE:-TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE MemUnitStalled WriteUnitStalled L2CacheHit}"
python3 - "$R/gpurun_out/counters_avail.txt" $WISH > /tmp/pmc_sets.txt <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
have = [c for c in sys.argv[2:] if re.search(r"\b" + re.escape(c) + r"\b", txt)]
miss = [c for c in sys.argv[2:] if c not in have]
sys.stderr.write("missing: " + " ".join(miss) + "\n")
# TCC / TCP / TA blocks have few counter slots each: 4 per pass, same block grouped
for i in range(0, len(have), 4): print(" ".join(have[i:i + 4]))
PY
i=0
while read -r set; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/ppmc$i -- python3 $R/tools/placement_pmc_probe.py 3 3 4 > $R/gpurun_out/ppmc$i.log 2>&1
  echo "set $i: $set rc=$?" >> $R/gpurun_out/ppmc_sets.txt
done < /tmp/pmc_sets.txt
cd $R
python3 - <<'PY' | tee gpurun_out/placement_pmc.txt
import csv, glob, json, collections, os
for d in sorted((d for d in glob.glob("gpurun_out/ppmc[0-9]*") if os.path.isdir(d)), key=lambda s: int(s.split("ppmc")[1])):
    log = d + ".log"
    pairs = None
    for line in open(log):
        if line.startswith("PAIRS "): pairs = json.loads(line[6:])
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not pairs or not files:
        print(d, "no data"); continue
    rows = collections.defaultdict(dict)     # dispatch -> counter -> value
    for r in csv.DictReader(open(files[0])):
        if "PcgFuseE<false, false>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = [rows[k] for k in sorted(rows)]
    disp = disp[pairs["warm_F"]:]
    per = pairs["per_pair_F"]
    names = sorted({c for r in disp for c in r})
    print(f"== {d}: {len(disp)} F dispatches after warm-up, expected {per * len(pairs['pairs'])}")
    print("pair        F_ms   " + "  ".join(f"{c:>34s}" for c in names))
    for k, p in enumerate(pairs["pairs"]):
        chunk = disp[k * per + 3:(k + 1) * per]
        if not chunk: continue
        avg = {c: sum(r.get(c, 0.0) for r in chunk) / len(chunk) for c in names}
        print(f"r{p['round']} Z{p['iz']} W{p['iw']}  {p['F_ms']:.4f}  " + "  ".join(f"{avg[c]:34.1f}" for c in names))
PY
rm -rf gpurun_out/ppmc[0-9]*/
