# batched Newton step: GPU parity tests, then timings at (1e7, 128) without / with bounds -- VALU form (nb <= 4) and matrix cores
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/nrbatch.txt; : > $O
timeout 900 python -m pytest tests/test_capi_retractions.py -m gpu -q -k "batched" 2>&1 | tail -4 | tee -a $O
for b in 0 1; do
  timeout 600 python tools/time_nrbatch.py 1e7 128 --bounds $b --nbs 2,4,8,16 --iters 40 2>&1 | grep "nb=" | tee -a $O
  LFPSQP_NRB_MFMA=1 timeout 600 python tools/time_nrbatch.py 1e7 128 --bounds $b --nbs 4 --iters 40 2>&1 | grep "nb=" | sed 's/^/[LFPSQP_NRB_MFMA=1] /' | tee -a $O
done
