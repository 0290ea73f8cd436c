# config 5's shape (4e7 x 512, 164 GB) with ball and bounds on ONE GPU, two outer iterations (r04p: 146 s with trial retractions 4 a pass on the VALU wide form), and config 4 end to end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
( time timeout 1500 python tools/run_config.py 4 4e7 512 --max-outer=2 ) > $O/c4_shape_4e7_512.txt 2>&1; tail -16 $O/c4_shape_4e7_512.txt
( time timeout 600 python tools/run_config.py 4 ) > $O/c4_1e7_128.txt 2>&1; tail -8 $O/c4_1e7_128.txt
( time timeout 600 python tools/run_config.py 3 ) > $O/c3_1e7_128.txt 2>&1; tail -6 $O/c3_1e7_128.txt
