cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
timeout 1200 python -m pytest tests/test_tangent_step.py tests/test_capi_inequalities.py tests/test_capi_retractions.py tests/test_gpu_parity_1e6.py tests/test_factored_basis.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
