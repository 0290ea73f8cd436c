# SQ issue/wait breakdown of the fused projected-CG kernel (separate --pmc passes, kernel-trace only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq$i -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_sq$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/sq_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "F" if "PcgFuseE" in n else ("K1" if ("PcgDirG" in n or "PcgDirF" in n) else None)
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:26s} n={len(v):3d} avg {sum(v)/len(v):16.0f}")
PY
rm -rf gpurun_out/pmc_sq*
