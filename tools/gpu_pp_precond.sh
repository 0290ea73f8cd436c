# ProjPenalty with the exact preconditioner: GPU parity tests, then configs 3 and 4 at full size with the reference's DEFAULT retraction
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/pp_precond.txt; : > $O
timeout 900 python -m pytest tests/test_pcg_precondition.py -m gpu -q -s 2>&1 | tail -6 | tee -a $O
echo "== config 3, ProjPenalty, live path" | tee -a $O; timeout 600 python tools/run_config.py 3 --pp 2>&1 | tail -12 | tee -a $O
echo "== config 3, ProjPenalty + exact preconditioner" | tee -a $O; timeout 600 python tools/run_config.py 3 --pp --precond 2>&1 | tail -12 | tee -a $O
echo "== config 4, ProjPenalty + exact preconditioner" | tee -a $O; timeout 1500 python tools/run_config.py 4 --pp --precond 2>&1 | tail -36 | tee -a $O
