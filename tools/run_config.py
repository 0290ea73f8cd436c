"""Run a BASELINE config end to end on the GPU at full size:  python tools/run_config.py {2|3|4} [n] [m] [--pp [--precond]] [--exact] [--ls-batch=K] [--matrix-cores] [--max-outer=K] [--banded | --banded-dense]"""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L

cfg = int(sys.argv[1]); args = [a for a in sys.argv[2:] if not a.startswith('--')]
pp = '--pp' in sys.argv
batch = int([a for a in sys.argv if a.startswith('--ls-batch=')][0].split('=')[1]) if any(a.startswith('--ls-batch=') for a in sys.argv) else 0      # (0 = automatic: DeviceOptions.ls_batch)
ctx = L.Context(0)
t0 = time.perf_counter()
if cfg == 2:
    n = int(float(args[0])) if args else 1_000_000
    J = ctx.matrix(n, 1); col = np.zeros((n, 1), order='F'); col[0, 0] = 1.0; J.upload(col)
    P = L.QuadLinearBallBox(ctx, n, 1, J, np.array([0.75])); x0 = np.ones(n)
elif cfg == 3 and ('--banded' in sys.argv or '--banded-dense' in sys.argv):
    # config 3's shape with BANDED equalities (4 nonzeros per row): --banded hands the solver the sparse twin, --banded-dense only the dense matrix
    n = int(float(args[0])) if args else 10_000_000; m = int(args[1]) if len(args) > 1 else 128
    ii = np.arange(n, dtype=np.int64); k = 4
    rows = np.repeat(ii, k); cols = ((((ii * m) // n)[:, None] + np.arange(k)[None, :]) % m).ravel()
    vals = (np.random.default_rng(5).standard_normal((n, k)) + 2.0 * (np.arange(k) == 0)).ravel()
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    Jct = S.to_dense()
    xs = ctx.vector(n).hash_fill(2); b = ctx.vector(m); L.spmv_t(S, xs, b)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download(), Jsp=S if '--banded' in sys.argv else None); x0 = np.ones(n)
elif cfg == 3:
    n = int(float(args[0])) if args else 10_000_000; m = int(args[1]) if len(args) > 1 else 128
    Jct = ctx.matrix(n, m, placed=True).hash_fill(1)
    xs = ctx.vector(n).hash_fill(2); b = ctx.vector(m); L.gemv_t(Jct, xs, b)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()); x0 = np.ones(n)
elif '--banded' in sys.argv or '--banded-dense' in sys.argv:
    # config 4's shape (ball + four-way bounds, slack row) with BANDED equalities
    n = int(float(args[0])) if args else 10_000_000; m = int(args[1]) if len(args) > 1 else 128
    ii = np.arange(n, dtype=np.int64); k = 4
    rows = np.repeat(ii, k); cols = ((((ii * m) // n)[:, None] + np.arange(k)[None, :]) % m).ravel()
    vals = (np.random.default_rng(5).standard_normal((n, k)) + 2.0 * (np.arange(k) == 0)).ravel()
    S = L.SparseMatrix(ctx, n + 1, m, rows, cols, vals)
    Jct = ctx.matrix(n + 1, m + 1, placed=True); S.to_dense(Jct)
    xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0); ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
    b = ctx.vector(m + 1); L.spmv_t(S, xs, b)
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu, Jsp=S if '--banded' in sys.argv else None); x0 = 0.5 * np.ones(n)
else:
    n = int(float(args[0])) if args else 10_000_000; m = int(args[1]) if len(args) > 1 else 128
    Jct = ctx.matrix(n + 1, m + 1, placed=True).hash_fill(1, 0, n, 1.0, n, m)
    xs = ctx.vector(n + 1).hash_fill(2, 0, 1.0, 0.0); ctx.check(ctx.L.lfpsqp_vec_fill_range(ctx.h, xs.h, n, 1, 0.0))
    b = ctx.vector(m + 1); L.gemv_t(Jct, xs, b, ncols=m)
    i = np.arange(n)
    xl = np.where((i % 4 == 1) | (i % 4 == 3), -1.0, -np.inf); xu = np.where((i % 4 == 2) | (i % 4 == 3), 1.0, np.inf)
    P = L.QuadLinearBallBox(ctx, n, m, Jct, b.download()[:m], R2=n / 2.0, xl=xl, xu=xu); x0 = 0.5 * np.ones(n)
ctx.sync(); print(f"setup {time.perf_counter()-t0:.2f}s  device={ctx.device_name}", flush=True)
t0 = time.perf_counter()
mo = [a for a in sys.argv if a.startswith('--max-outer=')]
ctx.options.ls_batch = batch
ctx.options.ls_batch_matrix_cores = '--matrix-cores' in sys.argv      # the opt-in matrix-core batch of trial retractions (default: the exact batch)
ctx.options.pp_precondition = '--precond' in sys.argv      # exact preconditioner of ProjPenalty's inner solves (lfpsqp_pcg_pre)
par = L.LFPSQPParams(do_project_retract=pp)
if '--exact' in sys.argv: par.linesearch = L.LinesearchOption.exact
if mo: par.maxiter = int(mo[0].split('=')[1])
x, obj, lam, ti = P.optimize(x0, par)
dt = time.perf_counter() - t0
print(ti); print(f"optimize wall {dt:.2f}s  f={obj[-1]:.6e}  |lam|max={np.abs(lam).max():.3e}")
pt = ctx.placement_info()
if pt[0]:
    print(f"placement (last placed allocation: basis + ProjCGWork pairs): {pt[0]} trials, kept {pt[1]}, fused-kernel trial ms min {min(pt[2]):.4f} / first {pt[2][0]:.4f} / max {max(pt[2]):.4f}")
