# round 5, first GPU call: the tangent-step fusion (tests, Gram timings with extra columns, outer-iteration timeline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a; O=gpurun_out/r05a
timeout 900 python -m pytest tests/test_tangent_step.py tests/test_factored_basis.py tests/test_streamed_gradients.py tests/test_warm_factorize.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 300 python tools/time_gram.py 128 2>&1 | tee $O/time_gram_128.txt
bash tools/gpu_outer_trace.sh stream > /dev/null 2>&1; cp gpurun_out/outer_trace_stream.txt $O/ 2>/dev/null; tail -50 gpurun_out/outer_trace_stream.txt
