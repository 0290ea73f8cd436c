#!/bin/bash
# the 2-ranks-on-one-GPU bench flow of tests/test_gpu_comm.py, repeated: looks for the intermittent stall seen once in round 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in $(seq 1 ${1:-8}); do
  port=$((20000 + RANDOM % 20000)); t0=$(date +%s.%N)
  timeout -s USR1 ${2:-120} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --comm host-gloo --device 0 --steps 6 --warmup 1 --rows 600000 --cols 24 --no-cpu-baseline > gpurun_out/two_$i.out 2> gpurun_out/two_$i.err
  rc=$?; t1=$(date +%s.%N)
  echo "run $i rc=$rc $(echo "$t1 - $t0" | bc) s  $(grep -c '^{' gpurun_out/two_$i.out) json line(s)"
  [ $rc -ne 0 ] && tail -40 gpurun_out/two_$i.err
done | tee gpurun_out/two_rank_loop.txt
