# round 5: one-pass projcg! with a diagonal + low-rank Hessian: GPU parity, then iteration times at (1e7, 128)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
timeout 600 python -m pytest tests/test_tangent_step.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
timeout 600 python tools/time_lowrank.py 1e7 128 1,2,4,8 2>&1 | tee $O/time_lowrank_1e7_128.txt
