# config 4 at full size (ball + four-way bounds, Newton retraction): automatic batch width (matrix cores) [and ls_batch = 4 with "both"]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/c4_full.txt; : > $O
for b in 0 ${1:+4}; do echo "== tools/run_config.py 4 --ls-batch=$b" | tee -a $O; timeout 900 python tools/run_config.py 4 --ls-batch=$b 2>&1 | tail -32 | tee -a $O; done
