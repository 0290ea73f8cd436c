# round-end evidence: GPU tests, bench line, rocprofv3 kernel stats (+ that process's own bench line), PMC traffic passes,
# SQ counters of the fused kernel, box description -> gpurun_out/final_*
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
bash tools/gpu_boxinfo.sh > /dev/null 2>&1; cp gpurun_out/boxinfo.txt $O/boxinfo.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench.json | cut -c1-600
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json 2> $O/stats.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/segprobe $R/tools/micro/segprobe.hip 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_calib -- /tmp/segprobe > $O/pmc_calib.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json $O/pmc_calib | head -12
find $O -name "*kernel_trace.csv" -delete
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-180
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/final/bench_profiled.json") if l.startswith("{")][-1])
print("profiled process: live F avg_launch_ms", d["roofline"]["avg_launch_ms"], "value", d["value"])
PY
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/sq_counters.txt $O/sq_counters.txt; cat $O/sq_counters.txt
# tangent setup: kernel statistics at the two headline shapes, SQ / LDS / MFMA counters and HBM traffic at (5e6, 512), the fp64 MFMA ceiling
bash tools/gpu_factorize_prof.sh 1e7 128 > /dev/null 2>&1; bash tools/gpu_factorize_prof.sh 5e6 512 > /dev/null 2>&1
cp gpurun_out/factorize_prof_1e7_128.txt gpurun_out/factorize_prof_5e6_512.txt $O/; cat $O/factorize_prof_5e6_512.txt | head -8
bash tools/gpu_pmc_factorize.sh 5e6 512 > /dev/null 2>&1; cp gpurun_out/factorize_sq_counters.txt $O/
bash tools/gpu_pmc_fetch_factorize.sh 5e6 512 > /dev/null 2>&1; cp gpurun_out/factorize_traffic.txt $O/; cat $O/factorize_traffic.txt
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_peak tools/micro/mfma_f64_peak.hip && /tmp/mfma_peak > $O/mfma_f64_peak.txt; tail -4 $O/mfma_f64_peak.txt
rm -rf gpurun_out/fprof_*
