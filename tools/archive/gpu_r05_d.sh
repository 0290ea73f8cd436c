# round 5: the WIDE matrix-core batched Newton step (133 .. 528 columns): GPU parity tests, then timings at (1e7, 512) with / without bounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d/nrbatch_wide.txt; : > $O
timeout 900 python -m pytest tests/test_capi_retractions.py -m gpu -q -x -k "batched" 2>&1 | tail -4 | tee -a $O
for b in 1 0; do
  timeout 900 python tools/time_nrbatch.py 1e7 512 --bounds $b --nbs 4,8 --iters 24 2>&1 | grep "nb=" | tee -a $O
  LFPSQP_NRB_MFMA=-1 timeout 900 python tools/time_nrbatch.py 1e7 512 --bounds $b --nbs 4 --iters 24 2>&1 | grep "nb=" | sed 's/^/[LFPSQP_NRB_MFMA=-1: VALU wide form] /' | tee -a $O
done
timeout 600 python tools/time_nrbatch.py 1e7 256 --bounds 1 --nbs 8 --iters 24 2>&1 | grep "nb=" | tee -a $O
# the wide forms of the one-pass kernels after the scalar-base fix (no readfirstlane loops around the tile's loads): F at the 8-GPU shard shape and at config 5's full size
python bench.py --rows 5e6 --cols 512 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r05d/bench_5e6_512.json 2> gpurun_out/r05d/bench_5e6_512.err; cut -c1-400 gpurun_out/r05d/bench_5e6_512.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_5e6_512.json")); print("shard shape (5e6, 512):", d["value"], "it/s, F", d["roofline"]["avg_launch_ms"], "ms, frac", d["roofline"]["frac"])
PY
python bench.py --rows 4e7 --cols 512 --basis scaled-hash --steps 10 --warmup 3 --no-cpu-baseline --no-extras --prewarm-seconds 0.5 > gpurun_out/r05d/bench_4e7_512.json 2> gpurun_out/r05d/bench_4e7_512.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_4e7_512.json")); print("config 5 full size (4e7, 512) on one GPU:", d["value"], "it/s, F", d["roofline"]["avg_launch_ms"], "ms, frac", d["roofline"]["frac"])
PY
