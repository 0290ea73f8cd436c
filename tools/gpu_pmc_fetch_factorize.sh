# HBM read traffic (FETCH_SIZE) of the tangent-setup kernels: tools/time_factorize.py N M under rocprofv3 --pmc
N=${1:-5e6}; M=${2:-512}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ff -- python3 $R/tools/time_factorize.py $N $M > $R/gpurun_out/pmc_ff.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fw -- python3 $R/tools/time_factorize.py $N $M > $R/gpurun_out/pmc_fw.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/factorize_traffic.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_f[fw]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = next((k for k in ("gram_kernel<true", "gram_kernel<false", "rmul_resident", "rmul_kernel") if k in n), None)
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("raw counter averages per launch (FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950: wide streaming reads count half, see tools/pmc_summary.py)")
for k, d in acc.items():
    for c, v in sorted(d.items()): print(f"   {k:22s} {c:12s} n={len(v):3d} avg {sum(v)/len(v)/1e6:10.3f} GB(raw, KB units)")
PY
rm -rf gpurun_out/pmc_ff gpurun_out/pmc_fw
