# round 6, first GPU check: the exact batch on the hardware (bit identity, config 4's chaotic searches under default options), the
# bounds-only driver, and config 4 at full size under the three settings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06a.txt; : > $O
timeout 1500 python -m pytest tests/test_exact_batch.py tests/test_bounds_only.py -m gpu -x -q -rA 2>&1 | tail -40 | tee -a $O
timeout 1500 python -m pytest tests/test_capi_retractions.py -m gpu -x -q -rA -k "config4 or batched or armijo or exact_linesearch" 2>&1 | grep -v "^PASSED\|^$" | tail -60 | tee -a $O
for a in "" "--matrix-cores" "--ls-batch=1"; do echo "== tools/run_config.py 4 $a" | tee -a $O; timeout 900 python tools/run_config.py 4 $a 2>&1 | tail -12 | tee -a $O; done
