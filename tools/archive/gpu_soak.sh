cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python tools/stress_small.py 40 | tail -1; done
echo "== config 4, default ProjPenalty retraction, n=1e6"; timeout 900 python tools/run_config.py 4 1e6 128 --pp 2>&1 | tail -12
echo "== config 3 shape at m=512, n=5e6 (NR)"; timeout 600 python tools/run_config.py 3 5e6 512 2>&1 | tail -4
echo "== config 3 shape at m=512, n=5e6 (PP)"; timeout 600 python tools/run_config.py 3 5e6 512 --pp 2>&1 | tail -4
