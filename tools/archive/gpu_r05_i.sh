# round 5: timelines after the border column rides with the Gram pass and the warm-start loops run on the host threads
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
timeout 900 python -m pytest tests/test_tangent_step.py tests/test_warm_factorize.py tests/test_capi_parity.py -m gpu -q -x -k "gram or factoriz or ksvd or warm or svd or tangent" 2>&1 | tail -3 | tee $O/pytest.txt
bash tools/gpu_outer_trace.sh stream > /dev/null 2>&1; cp gpurun_out/outer_trace_stream.txt $O/; head -12 $O/outer_trace_stream.txt | cut -c1-150
timeout 600 python tools/time_outer_bounds.py 2>&1 | tail -1 | tee $O/outer_bounds.txt
LFPSQP_TRACE_FACTORIZE=1 timeout 300 python tools/time_gram.py 128 2> $O/factorize_trace.txt | head -2 | tee $O/time_gram.txt; grep "warm" $O/factorize_trace.txt | tail -3
