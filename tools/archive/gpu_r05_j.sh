# round 5: the tangent step as projcg!'s initial projection (one more pass less): GPU parity of everything that runs through optimize, then the timelines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
timeout 1500 python -m pytest tests/test_tangent_step.py tests/test_factored_basis.py tests/test_streamed_gradients.py tests/test_elementwise.py tests/test_capi_inequalities.py tests/test_capi_retractions.py tests/test_gpu_parity_1e6.py tests/test_warm_factorize.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
bash tools/gpu_outer_trace.sh stream > /dev/null 2>&1; cp gpurun_out/outer_trace_stream.txt $O/; head -14 $O/outer_trace_stream.txt | cut -c1-150
timeout 600 python tools/time_outer_bounds.py 2>&1 | tail -1 | tee $O/outer_bounds.txt
