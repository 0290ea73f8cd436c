# round 5: the tangent step with BOUNDS in one pass: GPU parity (incl. config 4 trajectories), outer-iteration time, warm tangent setup with the host loops shared over threads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
timeout 1200 python -m pytest tests/test_tangent_step.py tests/test_capi_inequalities.py tests/test_capi_retractions.py tests/test_gpu_parity_1e6.py tests/test_factored_basis.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python tools/time_outer_bounds.py 2>&1 | tee $O/outer_bounds.txt
LFPSQP_TRACE_FACTORIZE=1 timeout 300 python tools/time_gram.py 128 2> $O/factorize_trace.txt | head -3 | tee $O/time_gram.txt; tail -12 $O/factorize_trace.txt
