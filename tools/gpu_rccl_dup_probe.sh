# What does the library's RCCL bring-up do when the ranks SHARE a GPU (the 1-GPU functional runs)?  Bounded by `timeout`: a refusal is expected.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
for W in 2 8; do
  ( time timeout 150 python3 bench.py --gpus $W --comm rccl --device 0 --rows 2e6 --cols 128 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --prewarm-seconds 0.2 --watchdog-seconds 100 ) > $O/rccl_shared_$W.out 2> $O/rccl_shared_$W.err
  echo "world $W: rc=$?" | tee -a $O/summary.txt
  grep -v "^\s*$" $O/rccl_shared_$W.err | tail -12
done
( time timeout 300 python3 bench.py --gpus 8 --comm p2p --device 0 --rows 2e6 --cols 128 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --prewarm-seconds 0.2 ) > $O/p2p_8.out 2> $O/p2p_8.err
echo "p2p 8: rc=$?" | tee -a $O/summary.txt; cut -c1-600 $O/p2p_8.out; tail -5 $O/p2p_8.err
