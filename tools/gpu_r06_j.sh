# round 6: how busy is the vector pipe?  SQ counters (separate --pmc passes, kernel trace only) of the tangent step (tools/time_tangent.py) and of the
# exact batch of Newton steps (tools/time_nrbatch.py, bounds, 4 trials), with F beside them (the tools run a few F launches for the placement)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_j_tan$i -- python3 $R/tools/time_tangent.py --reps 6 > $R/gpurun_out/pmc_j_tan$i.log 2>&1
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_j_nrb$i -- python3 $R/tools/time_nrbatch.py 1e7 128 --bounds 1 --nbs 4 --iters 12 > $R/gpurun_out/pmc_j_nrb$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r06j.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_j_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = None
        for tag, pat in (("tangent step, nonlinear class (TangentStepE<2, true>)", "TangentStepE<2, true>"), ("tangent step, linear class (TangentStepE<0, true>)", "TangentStepE<0, true>"),
                         ("exact batch, 4 trials with bounds (NRStepBatchRow<true, 4>)", "NRStepBatchRow<true, 4>"), ("Newton step, one trial with bounds (NRStepRow<true>)", "onepass_kernel<lfpsqp::NRStepRow<true>"),
                         ("F (PcgFuseE<false, false>)", "onepass_kernel<lfpsqp::PcgFuseE<false, false>")):
            if pat in n: key = tag
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters per launch (n = 1e7, m = 128), averaged over the launches of each kernel; VALU busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES x (4 SIMDs share a CU's busy cycles: / 4)")
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:26s} n={len(v):3d} avg {sum(v)/len(v):16.0f}")
    if "SQ_ACTIVE_INST_VALU" in d and "SQ_BUSY_CYCLES" in d:
        av = sum(d["SQ_ACTIVE_INST_VALU"]) / len(d["SQ_ACTIVE_INST_VALU"]); bc = sum(d["SQ_BUSY_CYCLES"]) / len(d["SQ_BUSY_CYCLES"])
        print(f"   => SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {av / bc:.2f}")
PY
rm -rf gpurun_out/pmc_j_*
