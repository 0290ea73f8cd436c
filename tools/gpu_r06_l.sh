# round 6: the Gram kernel with the number of extra right-hand columns as a template parameter (no run-time null tests if-converted into FMAs and selects):
# the library of the r06z bundle against the new build, two processes each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06l.txt; : > $O
V=lfpsqp.jl_amd/lib/variants
for rep in 1 2; do
timeout 300 python tools/time_gram_w.py 2>&1 | tail -1 | tee -a $O
timeout 300 python tools/time_gram_w.py --lib $V/liblfpsqp_gnx.so 2>&1 | tail -1 | tee -a $O
done
