# streamed gradients of the nonlinear class (row-scaled view of A) against materialised ones at n = 1e7, m = 128: the GPU tests, tools/time_elementwise.py in
# both modes, and the per-kernel durations of both runs (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_streamed_gradients.py tests/test_elementwise.py -x -q -m gpu > gpurun_out/ew_stream_tests.log 2>&1; tail -3 gpurun_out/ew_stream_tests.log
for mode in dense dense-materialised stream nostream; do
  timeout 900 python tools/time_elementwise.py $mode > gpurun_out/elementwise_$mode.json 2> gpurun_out/elementwise_$mode.err || tail -5 gpurun_out/elementwise_$mode.err
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ewprof_$mode -- python3 $R/tools/time_elementwise.py $mode > /dev/null 2> $R/gpurun_out/ewprof_$mode.err)
  python - $mode <<'PY' > gpurun_out/elementwise_${mode}_kernels.txt
import csv, glob, sys
f = sorted(glob.glob(f"gpurun_out/ewprof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1]))) if f else []
print(f"time_elementwise.py {sys.argv[1]}: kernels by total time (rocprofv3 --kernel-trace --stats)")
for r in rows[:22]:
    print(f"  {float(r['Percentage']):6.2f} %  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:10.1f} us   {r['Name'][:150]}")
PY
  rm -rf gpurun_out/ewprof_$mode
done
python - <<'PY'
import json
for pair in (("dense", "dense-materialised"), ("stream", "nostream")):
  a, b = (json.load(open(f"gpurun_out/elementwise_{m}.json")) for m in pair)
  print(pair[0], "against", pair[1], "(streamed gradients:", a["streamed_gradients"], b["streamed_gradients"], ")")
  for k in ("c_ms", "jac_ms", "hess_diag_ms", "tangent_setup_ms", "tangent_setup_factored_ms", "nr_iteration_ms_with_generator", "nr_iteration_ms_basis_only"):
    print(f"{k:34s} streamed {a[k]:9.3f}   materialised {b[k]:9.3f}")
  for k in ("optimize_newton", "optimize_projpenalty"):
    print(f"{k:34s} streamed {a[k]['seconds_per_outer_iteration']*1e3:9.1f} ms / outer iteration ({a[k]['outer_iterations']} it, tn {a[k]['tn_iterations']})   materialised {b[k]['seconds_per_outer_iteration']*1e3:9.1f} ({b[k]['outer_iterations']} it, tn {b[k]['tn_iterations']})")
PY
head -16 gpurun_out/elementwise_dense_kernels.txt; head -12 gpurun_out/elementwise_dense-materialised_kernels.txt
