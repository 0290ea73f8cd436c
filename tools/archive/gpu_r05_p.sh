# round 5: optimize on the chain-objective class at full size, one-pass tridiagonal solver against the callback path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05p
timeout 900 python3 tools/time_chain.py 1e7 128 8 2>&1 | tee gpurun_out/r05p/time_chain.txt
