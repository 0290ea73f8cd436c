# config 4 at full size, first 8 outer iterations: the round-3 tree (_r3/) against this tree, ls_batch = 4 and automatic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c4_ab.txt; : > $O
echo "== round 3 tree, ls_batch 4" | tee -a $O; (cd _r3 && timeout 600 python tools/run_config.py 4 --ls-batch=4 --max-outer=8 2>&1 | tail -14) | tee -a $O
echo "== this tree, ls_batch 4" | tee -a $O; timeout 600 python tools/run_config.py 4 --ls-batch=4 --max-outer=8 2>&1 | tail -14 | tee -a $O
echo "== this tree, automatic" | tee -a $O; timeout 600 python tools/run_config.py 4 --max-outer=8 2>&1 | tail -14 | tee -a $O
