cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/layoutprobe tools/micro/layoutprobe.hip && timeout 400 /tmp/layoutprobe 3 3 > gpurun_out/layoutprobe.txt 2>&1
