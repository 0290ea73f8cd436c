"""profiles/<tag>_streamed_gradients_1e7_128.txt from the files tools/gpu_ew_stream.sh left in gpurun_out/.   python tools/ew_stream_summary.py r04g"""
import json, shutil, sys
tag = sys.argv[1]
out = open(f"profiles/{tag}_streamed_gradients_1e7_128.txt", "w")
out.write("""Streamed gradients of the nonlinear constraint class (Jct = a view diag(phi'(x)) A + 2 x qw' of the constant A, lfpsqp_mat_view) against a materialised Jct
at n = 1e7, m = 128 on one MI355X: tools/gpu_ew_stream.sh = tools/time_elementwise.py in four modes, each also under rocprofv3 --kernel-trace --stats.
  dense / dense-materialised: mixed kinds + the common quadratic term (the system of profiles/r03p_elementwise_dense_1e7_128.json)
  stream / nostream: the same without the quadratic term (row scales only)
Times in ms; jac! includes its c! (1.6 ms; the driver calls it with cval == NULL from the second outer iteration on, which skips that).  optimize: SIX outer
iterations (tolerances off), each timed on its own by a callback that synchronises the device; the first (allocation by trial, uploads: "first") apart from the
steady ones (their median is the headline).  Both sides run the warm-started tangent setup and skip the redundant c!.

""")
for pair in (("dense", "dense-materialised"), ("stream", "nostream")):
    a, b = (json.load(open(f"gpurun_out/elementwise_{m}.json")) for m in pair)
    out.write(f"{pair[0]} against {pair[1]}\n")
    for k in ("c_ms", "jac_ms", "hess_diag_ms", "tangent_setup_ms", "tangent_setup_factored_ms", "nr_iteration_ms_with_generator", "nr_iteration_ms_basis_only"):
        out.write(f"  {k:34s} streamed {a[k]:9.3f}   materialised {b[k]:9.3f}\n")
    for k in ("optimize_newton", "optimize_projpenalty"):
        f = lambda d: "first %.0f, then " % d[k]["ms_first_iteration_with_setup"] + " ".join("%.1f" % v for v in d[k]["ms_per_outer_iteration"])
        out.write(f"  {k:34s} streamed {a[k]['seconds_per_outer_iteration']*1e3:9.1f}   materialised {b[k]['seconds_per_outer_iteration']*1e3:9.1f}   ms / outer iteration (median)\n")
        out.write(f"      streamed:     {f(a)}   (projcg! iterations {a[k]['tn_iterations'][:-1]}, retraction iterations {a[k]['retraction_iterations'][:-1]})\n")
        out.write(f"      materialised: {f(b)}\n")
    out.write("\n")
out.write("""Reading: jac! 5.6 -> 1.6 ms (what is left is its c!, which the driver no longer asks for); the fused projected-CG kernel over the view runs at the speed of the
plain one (16 more bytes per row next to 1056; the two one-workgroup launches around it are hidden behind it); the tangent setup in factored form costs 0.4 ms more
with row scales (the weighted Gram kernel, tools/time_gram.py: 3.6 against 3.2 ms) and 2 ms more with the rank-one term (one GEMV-T pass for A'(D W2 u)); the setup
WITH a materialised basis pays a second pass over the output (view_rows_kernel, 4.1 ms) -- off the fast path (the driver keeps the basis of a view in factored form).
Memory: one n x m matrix less (10.2 GB).

""")
for m in ("dense", "dense-materialised"):
    out.write(open(f"gpurun_out/elementwise_{m}_kernels.txt").read() + "\n")
out.close()
shutil.copy("gpurun_out/elementwise_dense.json", f"profiles/{tag}_elementwise_dense_streamed_1e7_128.json")
shutil.copy("gpurun_out/elementwise_dense-materialised.json", f"profiles/{tag}_elementwise_dense_materialised_1e7_128.json")
