# round 6, final library: steady outer iteration with bounds, config 4 at full size (default / matrix-core opt-in), the streamed class's timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06i.txt; : > $O
timeout 600 python tools/time_outer_bounds.py 2>&1 | tail -3 | tee -a $O
for a in "" "--matrix-cores"; do echo "== tools/run_config.py 4 $a" | tee -a $O; timeout 900 python tools/run_config.py 4 $a 2>&1 | tail -7 | tee -a $O; done
timeout 300 python tools/time_gram_w.py 2>&1 | tail -1 | tee -a $O
