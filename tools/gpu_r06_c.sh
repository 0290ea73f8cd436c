# round 6, third GPU check: the new parity sites at the protocol's sizes (config 4 at 1e6 x 128 to convergence, at 1e7 x 128 for one outer
# iteration; config 3 at full size with the default retraction), with their times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06c.txt; : > $O
timeout 600 python -m pytest tests/test_bounds_only.py -m gpu -x -q 2>&1 | grep -B30 "^E " | head -80 | tee -a $O
timeout 900 python -m pytest tests/test_staged_stores.py tests/test_tangent_step.py -m gpu -x -q 2>&1 | tail -4 | tee -a $O
timeout 2400 python -m pytest tests/test_gpu_parity_1e7.py -m gpu -x -q -s --durations=0 -k "config4 or config3" > gpurun_out/r06c_1e7.log 2>&1
grep -v "^$" gpurun_out/r06c_1e7.log | head -30 | cut -c1-600 | tee -a $O
grep -v "^$" gpurun_out/r06c_1e7.log | tail -15 | cut -c1-600 | tee -a $O
