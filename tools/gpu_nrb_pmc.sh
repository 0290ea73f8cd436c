# matrix-core batched Newton step (nrb_mfma_kernel): SQ / LDS / MFMA counters and HBM traffic, separate --pmc passes over tools/time_nrbatch.py
#   bash tools/gpu_nrb_pmc.sh [bounds 0|1] [nb]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=${1:-1}; NB=${2:-16}
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nrb$i -- python3 $R/tools/time_nrbatch.py 1e7 128 --bounds $B --nbs $NB --iters 12 > $R/gpurun_out/pmc_nrb$i.log 2>&1
done
cd $R
python - $B $NB <<'PY' | tee gpurun_out/nrb_counters_b${B}_nb${NB}.txt
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_nrb*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nrb_mfma_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"nrb_mfma_kernel at n = 1e7, m = 128, bounds = {sys.argv[1]}, {sys.argv[2]} trials: counter averages per launch")
for c, v in sorted(acc.items()): print(f"   {c:30s} n={len(v):4d} avg {sum(v)/len(v):18.0f}")
a = lambda c: sum(acc[c]) / len(acc[c]) if acc.get(c) else float('nan')
nsimd = 1024
print(f"derived: MFMA pipe busy cycles / launch {a('SQ_VALU_MFMA_BUSY_CYCLES'):.3g} = {a('SQ_INSTS_MFMA'):.3g} v_mfma_f64_16x16x4 x 64 cycles; share of the SIMD cycles of the launch "
      f"(SQ_BUSY_CYCLES is summed over the 32 shader engines: cycles = SQ_BUSY_CYCLES / 32) = {a('SQ_VALU_MFMA_BUSY_CYCLES') / (nsimd * a('SQ_BUSY_CYCLES') / 32):.3f}")
print(f"         LDS bank-conflict cycles / LDS active cycles = {a('SQ_LDS_BANK_CONFLICT') / a('SQ_LDS_IDX_ACTIVE'):.3f}; VALU instructions per MFMA = {a('SQ_INSTS_VALU') / a('SQ_INSTS_MFMA'):.1f}")
print(f"         HBM read {2 * a('FETCH_SIZE') * 1e3 / 1e9:.3f} GB (FETCH_SIZE in KB, x 2: gfx950 counts wide streaming reads at half), written {a('WRITE_SIZE') * 1e3 / 1e9:.3f} GB")
PY
tail -2 gpurun_out/pmc_nrb4.log
rm -rf gpurun_out/pmc_nrb*
