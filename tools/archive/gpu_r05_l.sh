# round 5: the tridiagonal one-pass iteration -- tests on the device, iteration and set-up times at (1e7, 128)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_projcg_tridiag.py tests/test_capi_parity.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r05l/pytest_tridiag.txt
cat gpurun_out/r05l/pytest_tridiag.txt
timeout 600 python3 tools/time_tridiag.py 1e7 128 2>&1 | tee gpurun_out/r05l/time_tridiag.txt
timeout 600 python3 tools/time_tridiag.py 1e7 128 --factored 2>&1 | tee -a gpurun_out/r05l/time_tridiag.txt
timeout 600 python3 tools/time_tridiag.py 4e6 300 2>&1 | tee -a gpurun_out/r05l/time_tridiag.txt
