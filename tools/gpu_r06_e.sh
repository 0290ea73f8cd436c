# round 6: the exact batch without its scratch array (FINDINGS.md 12.7): bit identity again, ms per step, config 4 at full size under the default options
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06e.txt; : > $O
timeout 900 python -m pytest tests/test_exact_batch.py tests/test_capi_retractions.py -m gpu -x -q -k "exact or config4 or batched or armijo" 2>&1 | tail -3 | tee -a $O
for b in 0 1; do
  echo "== exact batch (default), bounds=$b" | tee -a $O
  timeout 600 python tools/time_nrbatch.py 1e7 128 --bounds $b --nbs 2,4 --iters 40 2>&1 | grep "nb=" | tee -a $O
done
echo "== tools/run_config.py 4 (default: exact batch)" | tee -a $O; timeout 900 python tools/run_config.py 4 2>&1 | tail -6 | tee -a $O
echo "== exact batch at config 5's column count with bounds (wide VALU form), 4 trials" | tee -a $O
timeout 900 python tools/time_nrbatch.py 1e7 512 --bounds 1 --nbs 4 --iters 12 2>&1 | grep "nb=" | tee -a $O
