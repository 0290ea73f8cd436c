# round 5: the whole GPU suite with durations (new: full-size trajectory parity, 8-rank bench, 1e-10 preconditioner site), then one bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
( time timeout 2400 python -m pytest tests -m gpu -q -x -s --durations=25 ) > $O/pytest_gpu_full.txt 2>&1
grep -E "^\[parity|^\[pcg_pre|^\[bench --gpus 8|passed|failed|error" $O/pytest_gpu_full.txt | tail -40
grep -A 30 "slowest" $O/pytest_gpu_full.txt | head -32
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-900 $O/bench.json
