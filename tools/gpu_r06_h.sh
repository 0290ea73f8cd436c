# round 6: the Gram kernel without compares / selects in its loop (pad rows of the extra columns and of the staged weights are zeros by construction):
# default library (masks in the loop) against the variant, two processes each; then the GPU tests of the tangent setup on the variant's sources
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r06h2.txt; : > $O
V=lfpsqp.jl_amd/lib/variants
for rep in 1 2; do
timeout 300 python tools/time_gram_w.py 2>&1 | tail -1 | tee -a $O
timeout 300 python tools/time_gram_w.py --lib $V/liblfpsqp_gnomask.so 2>&1 | tail -1 | tee -a $O
done
