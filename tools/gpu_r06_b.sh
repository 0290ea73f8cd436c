# round 6, second GPU check: the exact batch with its running sums in LDS (bit identity again, step time at full size), the tangent step with
# staged stores (timeline of one outer iteration), config 4 at full size under the default options
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06b.txt; : > $O
timeout 1500 python -m pytest tests/test_exact_batch.py tests/test_bounds_only.py tests/test_staged_stores.py tests/test_tangent_step.py -m gpu -x -q 2>&1 | tail -5 | tee -a $O
for b in 0 1; do
  echo "== exact batch (default), bounds=$b" | tee -a $O
  timeout 600 python tools/time_nrbatch.py 1e7 128 --bounds $b --nbs 2,4 --iters 40 2>&1 | grep "nb=" | tee -a $O
  echo "== matrix-core batch (opt-in), bounds=$b" | tee -a $O
  LFPSQP_NRB_MFMA=0 timeout 600 python tools/time_nrbatch.py 1e7 128 --bounds $b --nbs 4,16 --iters 40 2>&1 | grep "nb=" | tee -a $O
done
bash tools/gpu_outer_trace.sh stream 2>&1 | tail -30 | tee -a $O
echo "== tools/run_config.py 4 (default: exact batch)" | tee -a $O; timeout 900 python tools/run_config.py 4 2>&1 | tail -6 | tee -a $O
