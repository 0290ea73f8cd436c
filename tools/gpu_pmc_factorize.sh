# SQ issue / wait / LDS / MFMA counters of the tangent-setup kernels (tools/time_factorize.py N M), separate --pmc passes
N=${1:-5e6}; M=${2:-512}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f$i -- python3 $R/tools/time_factorize.py $N $M > $R/gpurun_out/pmc_f$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/factorize_sq_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_f*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("gram_kernel<true", "gram_kernel<false", "rmul_resident", "rmul_kernel", "jacobi_round") if k in n), None)
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:30s} n={len(v):4d} avg {sum(v)/len(v):16.0f}")
PY
tail -3 gpurun_out/pmc_f4.log
rm -rf gpurun_out/pmc_f*
