# round 6: the whole GPU suite with durations (which tests take the time, which one fails)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06f.txt; : > $O
timeout 1800 python -m pytest tests -m gpu -q --durations=25 -rfEs > gpurun_out/r06f_full.log 2>&1
tail -80 gpurun_out/r06f_full.log | cut -c1-400 | tee -a $O
