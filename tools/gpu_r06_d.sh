# round 6: the tangent step's kernel time, A/B: staged stores (default) / row-by-row stores / row inputs requested behind the matrix loads / shorter bursts
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06d.txt; : > $O
V=lfpsqp.jl_amd/lib/variants
for rep in 1 2; do
echo "== default (staged stores)" | tee -a $O; timeout 600 python tools/time_tangent.py 2>&1 | grep "n=" | tee -a $O
echo "== row-by-row stores (LFPSQP_TANGENT_STAGE=0)" | tee -a $O; timeout 600 python tools/time_tangent.py --lib $V/liblfpsqp_tnostage.so 2>&1 | grep "n=" | tee -a $O
echo "== LFPSQP_OP_ROWLATE=1" | tee -a $O; timeout 600 python tools/time_tangent.py --lib $V/liblfpsqp_trowlate.so 2>&1 | grep "n=" | tee -a $O
done
for cap in 8 16; do echo "== bursts of at most $cap rounds" | tee -a $O; LFPSQP_STAGE_ROUNDS=$cap timeout 600 python tools/time_tangent.py 2>&1 | grep "n=" | tee -a $O; done
echo "== outer iteration with bounds (config 4's class), staged tangent step (default)" | tee -a $O; timeout 600 python tools/time_outer_bounds.py 2>&1 | tail -3 | tee -a $O
echo "== ... row-by-row stores" | tee -a $O; timeout 600 python tools/time_outer_bounds.py 1e7 128 --lib $V/liblfpsqp_tnostage.so 2>&1 | tail -3 | tee -a $O
timeout 600 python -m pytest tests/test_bounds_only.py tests/test_staged_stores.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O
