#!/bin/bash
# copy the evidence bundle of tools/gpu_final.sh (gpurun_out/final) into profiles/<tag>_*:  bash tools/collect_profiles.sh r02b
T=$1; O=gpurun_out/final; P=profiles      # (delete gpurun_out/final before the gpurun call: merged results of earlier calls stay there otherwise)
cp $O/bench.json $P/${T}_bench.json; cp $O/bench_profiled.json $P/${T}_bench_profiled.json; cp $O/boxinfo.txt $P/${T}_boxinfo.txt
cp $O/pytest_gpu.txt $P/${T}_pytest_gpu.txt; cp $O/sq_counters.txt $P/${T}_sq_counters.txt; cp $O/pmc_summary.json $P/${T}_pmc_summary.json
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" $P/${T}_bench_kernel_stats.csv
for k in fetch write calib; do cp "$(find $O/pmc_$k -name '*counter_collection.csv' | head -1)" $P/${T}_pmc_${k}_counter_collection.csv; done
for f in factorize_prof_1e7_128.txt factorize_prof_5e6_512.txt factorize_sq_counters.txt factorize_traffic.txt mfma_f64_peak.txt; do [ -f $O/$f ] && cp $O/$f $P/${T}_$f; done
ls -la $P | grep " ${T}_"
