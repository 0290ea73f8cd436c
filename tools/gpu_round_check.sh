# GPU tests + one bench line -> gpurun_out/check/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/check; O=gpurun_out/check
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-700 $O/bench.json
