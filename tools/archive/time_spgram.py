"""Gram matrix of a banded sparse block from its nonzeros (lfpsqp_spmat_gram) vs the dense MFMA Gram of its dense copy:
python tools/time_spgram.py [n] [m] [k] [--scatter]     (--scatter: the rows in random order, so that the column set changes at nearly every row; under rocprofv3 --kernel-trace --stats for the kernel times)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfpsqp_jl_amd as L
scatter = '--scatter' in sys.argv
argv = [a for a in sys.argv if not a.startswith('--')]
n = int(float(argv[1])) if len(argv) > 1 else 10_000_000
m = int(argv[2]) if len(argv) > 2 else 128
k = int(argv[3]) if len(argv) > 3 else 4
ctx = L.Context(0)
rows = np.repeat(np.random.default_rng(3).permutation(n) if scatter else np.arange(n), k)
cols = (((np.arange(n) * m) // n)[:, None] + np.arange(k)[None, :]) % m
vals = (np.random.default_rng(5).standard_normal((n, k)) + 2.0 * (np.arange(k) == 0)).ravel()
S = L.SparseMatrix(ctx, n, m, rows, cols.ravel(), vals)
Jd = S.to_dense()
out = {}
for tag, fn in (("from_nonzeros", lambda: S.gram()), ("dense", lambda: L.gram(Jd))):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(5):
        G = fn()
    ctx.sync(); out[tag] = ((time.perf_counter() - t0) * 1e3 / 5, G)
print(f"n={n} m={m} k={k}{' scattered rows' if scatter else ''}: Gram from the nonzeros {out['from_nonzeros'][0]:.3f} ms, dense {out['dense'][0]:.3f} ms, "
      f"max rel diff {np.abs(out['from_nonzeros'][1] - out['dense'][1]).max() / np.abs(out['dense'][1]).max():.2e}")
