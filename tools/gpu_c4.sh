cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/run_config.py 4 > gpurun_out/c4.log 2>&1; tail -4 gpurun_out/c4.log
LFPSQP_ONEPASS=-1 timeout 900 python tools/run_config.py 4 > gpurun_out/c4_two.log 2>&1; tail -3 gpurun_out/c4_two.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
