"""Sparse constraint gradients (lfpsqp_spmat; SURVEY §8 f4, the reference's README.md:80 to-do): the two sparse products against
numpy, the dense path they replace inside pcg! / ProjPenalty / c!, and an end-to-end optimize run -- same counts, same iterates."""
import numpy as np
import pytest
import scipy.sparse as sp

import lfpsqp_jl_amd as L
from lfpsqp_jl_amd.projpenalty import _JacPlain, _JacStacked
from oracle import lfpsqp_ref as R
from oracle import synth


def _is_emu(ctx):
    return "emulator" in ctx.device_name


def banded(n, m, k=3, seed=5):
    """n x m sparse Jct: row i touches k constraints around column i*m/n (a banded / block-structured equality system)."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), k)
    base = (np.arange(n) * m) // n
    cols = (base[:, None] + np.arange(k)[None, :]) % m
    vals = rng.standard_normal((n, k)) + 2.0 * (np.arange(k) == 0)
    return rows, cols.ravel(), vals.ravel()


@pytest.mark.parametrize("n,m,k", [(3000, 40, 3), (5000, 7, 2), (2500, 300, 5)])
def test_sparse_products_match_numpy(dev_ctx, n, m, k):
    ctx = dev_ctx
    rows, cols, vals = banded(n, m, k)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).tocsr()
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    assert S.nnz == A.nnz and S.ell_width == k
    rng = np.random.default_rng(1)
    vh, th, yh = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(n)
    t = L.spmv_t(S, ctx.vector(n, vh), ctx.vector(m))
    np.testing.assert_allclose(t.download(), A.T @ vh, rtol=0, atol=1e-12 * np.sqrt(n))
    y = ctx.vector(n, yh)
    L.spmv_n(S, ctx.vector(m, th), y, 1.5, -0.5)
    np.testing.assert_allclose(y.download(), 1.5 * (A @ th) - 0.5 * yh, atol=1e-13)
    L.spmv_n(S, ctx.vector(m, th), y, 1.0, 0.0)
    np.testing.assert_allclose(y.download(), A @ th, atol=1e-13)
    np.testing.assert_array_equal(S.to_dense().download(), A.toarray())
    # the sparse products are the dense ones on the same entries
    M = S.to_dense()
    t2, y2 = ctx.vector(m), ctx.vector(n)
    L.gemv_t(M, ctx.vector(n, vh), t2)
    L.gemv_n(M, ctx.vector(m, th), y2)
    np.testing.assert_allclose(t.download(), t2.download(), atol=1e-12 * np.sqrt(n))
    np.testing.assert_allclose(y.download(), y2.download(), atol=1e-13)


def test_sparse_edge_cases(dev_ctx):
    """duplicates add up, empty columns / rows, unsorted input, nothing at all; a dense row is refused."""
    ctx = dev_ctx
    n, m = 1000, 6
    rows = np.array([5, 5, 999, 0, 5, 700, 700])
    cols = np.array([2, 2, 5, 0, 0, 2, 2])
    vals = np.array([1.0, 2.5, -1.0, 4.0, 0.5, 1.0, -1.0])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).tocsr()          # scipy sums duplicates too
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    assert S.nnz == 5                                                      # 7 triplets, two duplicated positions
    v = np.cos(np.arange(n))
    np.testing.assert_allclose(L.spmv_t(S, ctx.vector(n, v), ctx.vector(m)).download(), A.T @ v, atol=1e-14)
    np.testing.assert_array_equal(S.to_dense().download(), A.toarray())
    E = L.SparseMatrix(ctx, n, m, [], [], [])
    assert E.nnz == 0
    np.testing.assert_array_equal(L.spmv_t(E, ctx.vector(n, v), ctx.vector(m).fill(7.0)).download(), np.zeros(m))
    y = ctx.vector(n, v)
    L.spmv_n(E, ctx.vector(m).fill(1.0), y, 1.0, 2.0)
    np.testing.assert_allclose(y.download(), 2.0 * v)
    with pytest.raises(L.LfpsqpError):
        L.SparseMatrix(ctx, 4, 300, np.zeros(300, dtype=int), np.arange(300), np.ones(300))       # one row with 300 nonzeros
    with pytest.raises(L.LfpsqpError):
        L.SparseMatrix(ctx, 4, 3, [4], [0], [1.0])                                                # row out of range


@pytest.mark.parametrize("bounds", [False, True])
def test_pcg_with_a_sparse_operator_is_the_dense_solve(dev_ctx, bounds):
    """pcg! (src/retractions.jl:179-246) with lfpsqp_basis.S set: two sparse products per iteration instead of a dense pass --
    the same iteration count and flag, iterates equal to rounding, the system solved."""
    ctx = dev_ctx
    n, m = (2400, 24) if _is_emu(ctx) else (20000, 64)
    rows, cols, vals = banded(n, m, 3)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).tocsr()
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    J = S.to_dense()
    mu, tol = 0.05, 1e-9
    rng = np.random.default_rng(2)
    res = {}
    for label, Jsp in (("dense", None), ("sparse", S)):
        w = L.ProjPenaltyWork(ctx, m, n, bounds)
        if bounds:
            from lfpsqp_jl_amd.inequality import InequalityData, InequalityDecomp, StackedVector, generate_initial_y_, inequality_gradient_
            i = np.arange(n)
            xl = np.where(i % 3 == 1, -1.0, -np.inf)
            xu = np.where(i % 3 == 2, 1.0, np.inf)
            idata = InequalityData(ctx, xl, xu)
            xa = StackedVector(ctx, n)
            xa.upload(0.5 * synth.hash_vector(2, n), 0)
            generate_initial_y_(xa, idata)
            dec = InequalityDecomp(ctx, n, m, J)
            inequality_gradient_(dec, xa, idata)
            op = _JacStacked(dec, w, Jsp)
            op.refresh()
            x, r = StackedVector(ctx, n), StackedVector(ctx, n)
            r.upload2(np.random.default_rng(3).standard_normal(2 * n))
        else:
            op = _JacPlain(J, w, Jsp)
            x, r = ctx.vector(n), ctx.vector(n, np.random.default_rng(3).standard_normal(n))
        flag, it = L.pcg_(mu, op, L.no_precondition, x, r, w.p, w.z, None, tol, 400)
        res[label] = (flag, it, x.download2() if bounds else x.download())
    # (the stopping test norm(r) > tol can flip by an iteration when the residual lands within rounding of tol: the two paths sum in
    # different orders -- the same slack the dense-vs-oracle comparisons allow)
    assert res["sparse"][0] == res["dense"][0] == 0 and res["dense"][1] > 3 and abs(res["sparse"][1] - res["dense"][1]) <= 2
    exact = res["sparse"][1] == res["dense"][1]
    np.testing.assert_allclose(res["sparse"][2], res["dense"][2], rtol=0, atol=(1e-11 if exact else 1e-7) * np.abs(res["dense"][2]).max())
    if not bounds:       # (J'J + mu I) x = b
        b = np.random.default_rng(3).standard_normal(n)
        xs = res["sparse"][2]
        assert np.linalg.norm(A @ (A.T @ xs) + mu * xs - b) <= 2 * tol


@pytest.mark.parametrize("newton", [False, True])
@pytest.mark.parametrize("bounds", [False, True])
def test_optimize_with_sparse_equalities_matches_the_dense_run_and_the_oracle(dev_ctx, bounds, newton):
    """End to end: f = ||x||^2 subject to banded sparse equalities (and bounds): the reference's DEFAULT retraction (ProjPenalty: c!,
    jac!, pcg! all on the sparse block) and the Newton retraction (do_project_retract = false: every step on the nonzeros, also on the
    bound manifold) -- the trajectory of the dense run and of the oracle."""
    ctx = dev_ctx
    n, m = (600, 8) if _is_emu(ctx) else (6000, 24)
    rows, cols, vals = banded(n, m, 3, seed=9)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).toarray()
    xs = synth.hash_vector(2, n)
    prob0 = synth.QuadLinearProblem(np.asfortranarray(A), A.T @ xs)
    x0 = xs + 0.05 * synth.hash_vector(6, n)
    xl = xu = None
    if bounds:
        i = np.arange(n)
        xl = np.where(i % 3 == 1, -2.0, -np.inf)
        xu = np.where(i % 3 == 2, 2.0, np.inf)
    maxiter = 3
    tr0 = []
    xr, objr, lamr, tir = R.optimize(prob0.f, prob0.grad_, prob0.c_, prob0.jac_, prob0.hess_lag_vec_, x0, xl, xu, m,
                                     R.LFPSQPParams(disp=R.DisplayOption.off, maxiter=maxiter, do_project_retract=not newton), trace=tr0)
    out = {}
    for label in ("dense", "sparse"):
        S = L.SparseMatrix(ctx, n, m, rows, cols, vals) if label == "sparse" else None
        P = L.QuadLinearBallBox(ctx, n, m, ctx.matrix(n, m, np.asfortranarray(A)), prob0.b, xl=xl, xu=xu, Jsp=S)
        tr = []
        x, obj, lam, ti = P.optimize(x0, L.LFPSQPParams(disp=L.DisplayOption.off, maxiter=maxiter, do_project_retract=not newton), trace=tr)
        out[label] = (tr, x, obj, ti)
    trd, xd, objd, tid = out["dense"]
    trs, xsp, objs, tis = out["sparse"]
    assert tis.iter == tid.iter == tir.iter
    for a, b, c in zip(trs, trd, tr0):
        for k in ("tn_iter", "steptype", "mtype", "retract_iter1", "alpha", "ls_flag", "rank"):
            assert a.get(k) == b.get(k) == c.get(k), (a["iter"], k, a.get(k), b.get(k), c.get(k))
        assert abs((a.get("retract_iter2") or 0) - (c.get("retract_iter2") or 0)) <= 2
        assert np.linalg.norm(a["x"] - c["x"]) <= 1e-9 * np.linalg.norm(c["x"])
        assert np.linalg.norm(a["x"] - b["x"]) <= 1e-10 * np.linalg.norm(b["x"])
    np.testing.assert_allclose(objs, objr, rtol=1e-10)


def _exact_gram(A, w=None):
    """A' diag(w) A with every product and sum exact (Python fractions), rounded once."""
    from fractions import Fraction
    n, m = A.shape
    G = np.zeros((m, m))
    nz = [[(j, Fraction(float(A[i, j]))) for j in range(m) if A[i, j] != 0.0] for i in range(n)]
    acc = {}
    for i in range(n):
        wi = Fraction(float(w[i])) if w is not None else Fraction(1)
        for a, (ja, va) in enumerate(nz[i]):
            for jb, vb in nz[i][a:]:
                acc[(ja, jb)] = acc.get((ja, jb), Fraction(0)) + wi * va * vb
    for (ja, jb), v in acc.items():
        G[ja, jb] = G[jb, ja] = float(v)
    return G


@pytest.mark.parametrize("case", ["plain", "weighted", "ball_weighted", "two_tiles", "wide_range", "k6", "k11", "scattered", "slack_weight",
                                  "k12", "k16", "k16_weighted", "k13_scattered", "k20", "k12_boxes", "k5_boxes"])
def test_gram_from_the_nonzeros_is_exact_and_order_independent(dev_ctx, case):
    """lfpsqp_spmat_gram: the scattered accumulation in two fixed-point limbs.  The device forms each term w_i v_a v_b in floating point (two
    roundings) and then sums EXACTLY: against exact rational arithmetic on the same rounded terms the result is the correctly rounded sum up
    to the dropped third limb; a permutation of the rows changes nothing, bit for bit; the dense extra columns (SpMV-T, dot products) and
    the weights agree with numpy."""
    from fractions import Fraction
    ctx = dev_ctx
    n, m, k = (900, 150, 3) if case == "two_tiles" else (1300, 14, {"k6": 6, "k11": 11}.get(case, 3))
    if case in ("k12", "k16", "k16_weighted", "k13_scattered", "k20"):       # 9..16 nonzeros: a group of lanes shares a row (sp_gram_split_kernel); 20: per term
        n, m, k = 1100, 70, {"k12": 12, "k13_scattered": 13, "k20": 20}.get(case, 16)
    if case in ("k12_boxes", "k5_boxes"):            # several column boxes of 4096 rows, three tiles of G: most (box, tile) pairs are skipped
        n, m, k = 9000, 150, 12 if case == "k12_boxes" else 5
    rows, cols, vals = banded(n, m, k, seed=4)
    if case in ("scattered", "k13_scattered"):                          # no two consecutive rows with the same columns, rows of 1..3 nonzeros
        rng0 = np.random.default_rng(11)
        cols = np.stack([rng0.permutation(m)[:k] for _ in range(n)]).ravel()
        keep = rng0.random(cols.size) < 0.8
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
    if case == "wide_range":
        vals = vals * np.logspace(0, -9, m)[cols] * np.where(np.arange(vals.size) % 7 == 0, 1e3, 1.0)
    rng = np.random.default_rng(3)
    w = rng.random(n) + 0.2 if case in ("weighted", "ball_weighted", "slack_weight", "k16_weighted") else None
    if case == "slack_weight":                       # a row without nonzeros (the slack row of a ball constraint) with a weight 2^40 times the others:
        keep = rows != n - 1                         # the bound on the terms is measured over the rows that have nonzeros, so nothing is lost
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        w[n - 1] = 2.0 ** 40
    extra = 2 if case == "ball_weighted" else 0
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).toarray()
    X = rng.standard_normal((n, extra))
    Ad = np.asfortranarray(np.hstack([A, X]))
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    Jd = ctx.matrix(n, m + extra, Ad) if extra else None
    wv = ctx.vector(n, w) if w is not None else None
    G = S.gram(Jd, wv)
    # the terms as the device forms them: (w_i * v_a) * v_b in floating point, then exact sums
    acc = {}
    A0 = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).tocsr()
    for i in range(n):
        lo, hi = A0.indptr[i], A0.indptr[i + 1]
        cs, vs = A0.indices[lo:hi], A0.data[lo:hi]
        order = np.argsort(cs)
        cs, vs = cs[order], vs[order]
        wi = w[i] if w is not None else 1.0
        for a in range(len(cs)):
            wa = wi * vs[a]
            for b in range(a, len(cs)):
                acc[(cs[a], cs[b])] = acc.get((cs[a], cs[b]), Fraction(0)) + Fraction(float(wa * vs[b]))
    Gx = np.zeros((m, m))
    for (ja, jb), v in acc.items():
        Gx[ja, jb] = Gx[jb, ja] = float(v)
    Gs = G[:m, :m]
    # one rounding + the dropped third limb, ENTRYWISE relative to the magnitudes of the two columns (power-of-two column scaling)
    cmax = np.array([np.abs(vals[cols == j]).max() if np.any(cols == j) else 0.0 for j in range(m)])
    wmax = w[:n - 1].max() if case == "slack_weight" else (w.max() if w is not None else 1.0)
    assert np.all(np.abs(Gs - Gx) <= 2.0 ** -52 * np.abs(Gx) * 0.51 + 4.0 * wmax * np.outer(cmax, cmax) * 2.0 ** -70 * n)
    if case != "wide_range":
        np.testing.assert_array_equal(Gs, Gx)
    np.testing.assert_array_equal(Gs, Gs.T)
    Gn = Ad.T @ ((w[:, None] if w is not None else 1.0) * Ad)
    np.testing.assert_allclose(G, Gn, rtol=0, atol=1e-13 * np.abs(Gn).max())
    np.testing.assert_array_equal(G, G.T)
    # the rows in another order: the same matrix, bit for bit
    perm = rng.permutation(n)
    inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
    S2 = L.SparseMatrix(ctx, n, m, inv[rows], cols, vals)
    Jd2 = ctx.matrix(n, m + extra, np.asfortranarray(Ad[perm])) if extra else None
    wv2 = ctx.vector(n, w[perm]) if w is not None else None
    G2 = S2.gram(Jd2, wv2)
    np.testing.assert_array_equal(G2[:m, :m], Gs)
    np.testing.assert_allclose(G2, G, rtol=0, atol=1e-13 * np.abs(Gn).max())
    # refused, not approximated: rows wider than 32 nonzeros
    r3, c3, v3 = banded(200, 40, 33, seed=1)
    with pytest.raises(L.LfpsqpError):
        L.SparseMatrix(ctx, 200, 40, r3, c3, v3).gram()


def test_gram_from_the_nonzeros_with_many_row_slices(dev_ctx, monkeypatch):
    """More than 64 slices of the rows per tile of G: their integer partial sums are folded before the final reduction
    (sp_gram_fold_kernel) -- same matrix as with few slices, bit for bit, and equal to numpy's to rounding."""
    ctx = dev_ctx
    n, m, k = 4096 * 70 + 17, 40, 9
    rows, cols, vals = banded(n, m, k, seed=6)
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    G1 = S.gram()
    monkeypatch.setenv("LFPSQP_SPGRAM_SLICES", "64")
    G2 = S.gram()
    monkeypatch.delenv("LFPSQP_SPGRAM_SLICES")
    np.testing.assert_array_equal(G1, G2)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).tocsr()
    Gn = (A.T @ A).toarray()
    np.testing.assert_allclose(G1, Gn, rtol=0, atol=1e-12 * np.abs(Gn).max())


@pytest.mark.parametrize("case", ["plain", "weighted", "ball_column", "no_dense_twin", "ill_conditioned", "wide_k", "two_panels", "very_wide_k",
                                  "wide_k_scattered"])
def test_factorize_from_the_nonzeros_is_the_dense_factorisation(dev_ctx, case):
    """lfpsqp_factorize_sp (Gram matrix from the nonzeros, exactly accumulated; basis-forming products Z = [S | extra columns] * W from the
    nonzeros) against lfpsqp_factorize on the dense matrix: the same factorisation up to the rounding of the Gram matrix and of the product;
    Z = A * W and Z' diag(w2) Z = I checked directly.  With the Gram matrix taken from a dense copy (context setting LFPSQP_SPGRAM=-1, and
    always when a row has more than 16 nonzeros: very_wide_k; wide_k = 9 nonzeros per row goes through the lane-group kernel) Sigma / Vt / W are
    those of the dense factorisation bit for bit; likewise with 9 .. 16 nonzeros per row in scattered columns (wide_k_scattered)."""
    import os
    ctx = dev_ctx
    n, m, k = (1000 if _is_emu(ctx) else 26000, 200, 4) if case == "two_panels" else (3100, 12 if case != "very_wide_k" else 40,
                                                                                     {"wide_k": 9, "very_wide_k": 34, "wide_k_scattered": 9}.get(case, 3))
    rows, cols, vals = banded(n, m, k, seed=8)
    if case == "wide_k_scattered":                   # 9 nonzeros per row in columns that change from row to row: the dense twin's Gram matrix
        cols = np.stack([np.random.default_rng(100 + i).permutation(m)[:k] for i in range(n)]).ravel()
    if case == "ill_conditioned":                    # refinement rounds: several basis-forming products from the nonzeros
        vals = vals * np.logspace(0, -7, m)[cols]
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, m)).toarray()
    extra = 1 if case == "ball_column" else 0
    rng = np.random.default_rng(2)
    Ad = np.asfortranarray(np.hstack([A, rng.standard_normal((n, extra))])) if extra else np.asfortranarray(A)
    M = m + extra
    wh = rng.random(n) + 0.2 if case == "weighted" else None
    res = {}
    for label in ("exact", "dense_copy"):
        if label == "dense_copy":
            os.environ["LFPSQP_SPGRAM"] = "-1"
        try:
            c = ctx if label == "exact" else L.Context(0, ctx.L)
        finally:
            os.environ.pop("LFPSQP_SPGRAM", None)
        S = L.SparseMatrix(c, n, m, rows, cols, vals)
        Jd = c.matrix(n, M, Ad)
        w = c.vector(n, wh) if wh is not None else None
        Z0, Z1 = c.matrix(n, M), c.matrix(n, M)
        W0, W1 = np.zeros((M, M), order='F'), np.zeros((M, M), order='F')
        S0, Vt0, r0 = L.ksvd_(Jd, Z0, w2=w, W=W0)
        S1, Vt1, r1 = L.ksvd_(None if case == "no_dense_twin" else Jd, Z1, w2=w, W=W1, Jsp=S)
        res[label] = (S0, Vt0, W0, r0, Z0.download(), S1, Vt1, W1, r1, Z1.download())
        if c is not ctx:
            c.close()
    for label, (S0, Vt0, W0, r0, Zh0, S1, Vt1, W1, r1, Zh1) in res.items():
        assert r1 == r0 == M
        fast = S0[0] ** 2 <= 10.0 * S0[-1] ** 2           # lfpsqp_factorize's one-Gram-one-product path
        if fast and (label == "dense_copy" or k > 16 or case == "wide_k_scattered"):    # same Gram matrix, same replicated small step: bit for bit
            np.testing.assert_array_equal(S1, S0)
            np.testing.assert_array_equal(Vt1, Vt0)
            np.testing.assert_array_equal(W1, W0)
        else:                                            # Gram matrices that differ in their last bits / refinement rounds on another product
            np.testing.assert_allclose(S1, S0, rtol=1e-11 if not fast else 1e-13)
        scale = np.abs(Zh0).max()
        tol = 1e-9 if case == "ill_conditioned" else 1e-11        # (eigenvectors of a matrix perturbed in its last bits: rounding / gap)
        assert np.abs(Zh1 - Zh0).max() <= tol * scale * (M ** 0.5)
        assert np.abs(Zh1 - Ad @ W1).max() <= (1e-9 if case == "ill_conditioned" else 2e-14 if fast else 1e-12) * scale * (M ** 0.5)
        wv = wh if wh is not None else np.ones(n)
        if case != "ill_conditioned":
            assert np.abs(Zh1.T @ (wv[:, None] * Zh1) - np.eye(M)).max() <= (1e-12 if fast else 1e-10)
        np.testing.assert_allclose(S1, np.linalg.svd(np.sqrt(wv)[:, None] * Ad, compute_uv=False), rtol=1e-9 if case != "ill_conditioned" else 1e-6,
                                   atol=1e-13 * S1[0])


@pytest.mark.gpu
def test_sparse_products_at_1e6_rows_against_the_dense_kernels():
    """The §8 f4 bar: a banded Jct at n = 1e6, m = 128 (4 nonzeros per row): both sparse products agree with the dense GEMV
    kernels on the same entries, and run at a multiple of their speed (48 MB of nonzeros against a 1 GB matrix)."""
    ctx = L.Context(0)
    n, m, k = 1_000_000, 128, 4
    rows, cols, vals = banded(n, m, k)
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    M = S.to_dense()
    v, t = ctx.vector(n).hash_fill(5), ctx.vector(m).hash_fill(6)
    ts, td, ys, yd = ctx.vector(m), ctx.vector(m), ctx.vector(n), ctx.vector(n)
    L.spmv_t(S, v, ts); L.gemv_t(M, v, td)
    L.spmv_n(S, t, ys); L.gemv_n(M, t, yd)
    np.testing.assert_allclose(ts.download(), td.download(), rtol=0, atol=1e-12 * np.abs(td.download()).max())
    L.axpby(1.0, ys, -1.0, yd)
    assert L.nrm2(yd) <= 1e-13 * L.nrm2(ys)
    ms = {}
    for name, fn in (("spmv_t", lambda: L.spmv_t(S, v, ts)), ("gemv_t", lambda: L.gemv_t(M, v, td)), ("spmv_n", lambda: L.spmv_n(S, t, ys)),
                     ("gemv_n", lambda: L.gemv_n(M, t, yd))):
        fn()
        ctx.timer_begin()
        for _ in range(20):
            fn()
        ms[name] = ctx.timer_end() / 20
    print("[sparse] ms per product:", {k_: round(v_, 4) for k_, v_ in ms.items()}, "nnz bytes", S.nnz * 12)
    assert ms["spmv_t"] < ms["gemv_t"] and ms["spmv_n"] < ms["gemv_n"]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [4, 7])
def test_gram_from_the_nonzeros_at_1e6_rows(k):
    """lfpsqp_spmat_gram at n = 1e6, m = 128 (both register-resident kernels: rows of 4 and of 7 nonzeros), weighted: equal to the MFMA Gram of
    the dense copy to rounding, bit-identical under a permutation of the rows (which changes every run of equal column sets, every
    workgroup's slice and the order of all atomics), and faster than the dense kernel."""
    ctx = L.Context(0)
    n, m = 1_000_000, 128
    rows, cols, vals = banded(n, m, k)
    rng = np.random.default_rng(12)
    wh = rng.random(n) + 0.25
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    w = ctx.vector(n, wh)
    G = S.gram(w2=w)
    Gd = L.gram(S.to_dense(), w2=w)
    np.testing.assert_allclose(G, Gd, rtol=0, atol=1e-13 * np.abs(Gd).max())
    np.testing.assert_array_equal(G, G.T)
    perm = rng.permutation(n)
    inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
    S2 = L.SparseMatrix(ctx, n, m, inv[rows], cols, vals)
    G2 = S2.gram(w2=ctx.vector(n, wh[perm]))
    np.testing.assert_array_equal(G2, G)
    ms = {}
    Jd = S.to_dense()
    for name, fn in (("from_nonzeros", lambda: S.gram(w2=w)), ("scattered_rows", lambda: S2.gram(w2=w)), ("dense", lambda: L.gram(Jd, w2=w))):
        fn()
        ctx.timer_begin()
        for _ in range(5):
            fn()
        ms[name] = ctx.timer_end() / 5
    print("[sparse gram] ms:", {a: round(b, 4) for a, b in ms.items()})
    assert ms["from_nonzeros"] < ms["dense"]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ball", [False, True])
def test_factorize_from_the_nonzeros_at_1e6_rows(ball):
    """Tangent setup of a banded Jct at n = 1e6, m = 128 (4 nonzeros per row; with a dense ball column: M = 129): lfpsqp_factorize_sp
    against lfpsqp_factorize on the dense twin -- identical Sigma / Vt (same Gram matrix, or rounding level when refinement rounds run),
    the same basis to rounding, Z = A W and Z'Z = I by device products, and faster."""
    import time
    ctx = L.Context(0)
    n, m, k = 1_000_000, 128, 4
    rows, cols, vals = banded(n, m, k)
    S = L.SparseMatrix(ctx, n, m, rows, cols, vals)
    M = m + (1 if ball else 0)
    Jd = ctx.matrix(n, M)
    S.to_dense(Jd)
    if ball:
        Jd.upload(synth.hash_vector(9, n)[:, None], col0=m)          # the dense extra column (2x of a ball constraint, say)
    Z0, Z1 = ctx.matrix(n, M), ctx.matrix(n, M)
    W1 = np.zeros((M, M), order='F')
    S0, Vt0, r0 = L.ksvd_(Jd, Z0)
    S1, Vt1, r1 = L.ksvd_(Jd, Z1, W=W1, Jsp=S)
    assert r0 == r1 == M
    np.testing.assert_allclose(S1, S0, rtol=1e-12)
    G = L.gram(Z1)
    assert np.abs(G - np.eye(M)).max() <= 1e-11
    Zc = ctx.matrix(n, M)
    L.rmul(Jd, W1, Zc)                                  # Z = A W with the dense MFMA product
    for j in range(0, M, 37):
        a, b = Z1.download(j, 1), Zc.download(j, 1)
        assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max() * 10
    t = {}
    for tag, kw in (("dense", {}), ("from_nonzeros", {"Jsp": S})):
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(5):
            L.ksvd_(Jd, Z1, **kw)
        ctx.sync(); t[tag] = (time.perf_counter() - t0) / 5 * 1e3
    print("[sparse] factorize ms:", {k_: round(v_, 3) for k_, v_ in t.items()}, "cond", S1[0] / S1[-1])
    assert t["from_nonzeros"] < t["dense"]
    ctx.close()


@pytest.mark.parametrize("has_ball", [False, True])
def test_newton_retraction_on_the_nonzeros_is_the_dense_retraction(dev_ctx, has_ball):
    """retract!(::NR) with a sparse twin of the linear block and the basis generator known: U*ddelta = Jct*(W*ddelta), so a Newton step is
    one row pass over the ELL entries (+ the dense ball column), the same row update, and c! = S'x by the sparse product -- no dense
    matrix is read.  Same (flag, iterations), iterate and cval as the dense one-stream step on the same data, plain and with a ball
    constraint (one dense extra column); the bound manifold is covered through optimize above."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m_lin, k = (2400 if emu else 400_000), 10, 3
    m = m_lin + (1 if has_ball else 0)
    N = n + (1 if has_ball else 0)
    rows, cols, vals = banded(n, m_lin, k, seed=21)
    Jh = np.zeros((N, m), order='F')
    np.add.at(Jh, (rows, cols), vals)
    xs_h = np.concatenate([0.6 * synth.hash_vector(2, n), [0.0]])[:N]
    if has_ball:
        xs_h[n] = xs_h[:n] @ xs_h[:n] - 0.4 * n
    b = Jh[:, :m_lin].T @ xs_h
    pert = 5e-3 * np.random.default_rng(4).standard_normal(N)
    out = {}
    for mode in ("dense", "sparse"):
        Jct = ctx.matrix(N, m, Jh)
        Ssp = L.SparseMatrix(ctx, N, m_lin, rows, cols, vals) if mode == "sparse" else None
        cons = L.DeviceConstraints(Jct, m_lin, b, has_ball, 0.4 * n, n, n if has_ball else -1, Jsp=Ssp)
        xs_dev = ctx.vector(N, xs_h)
        cv = np.zeros(m)
        cons.jac_(Jct, cv, xs_dev)                            # (ball column of Jct at xs)
        Z, W = ctx.matrix(N, m), np.zeros((m, m), order='F')
        S, Vt, rank = L.ksvd_(Jct, Z, W=W, Jsp=Ssp)
        assert rank == m
        nr = L.NR(L.DeviceBasis(Z, generator=(Jct, W)), S, Vt, 1e-10, 60, L.NRWork(m), False, None)
        xt, xnew = ctx.vector(N, xs_h + pert), ctx.vector(N)
        cval = np.zeros(m)
        flag, it, _ = L.retract_(cval, xnew, cons, xt, xs_dev, nr)
        out[mode] = (flag, it, xnew.download(), cval.copy())
    (f0, i0, x0, c0), (f1, i1, x1, c1) = out["dense"], out["sparse"]
    assert (f1, i1) == (f0, i0) and f0 == 0 and i0 >= (2 if has_ball else 1)
    assert np.abs(x1 - x0).max() <= 1e-12 * np.abs(x0).max()
    assert np.abs(c1 - c0).max() <= 1e-11 and np.abs(c1).max() < 1e-9


@pytest.mark.parametrize("case", ["plain", "extra_column", "operator_callback"])
def test_projected_cg_in_factored_form_on_the_nonzeros(dev_ctx, case):
    """projcg! with the basis applied as U t = A (W t), U'v = W'(A'v) on the nonzeros of A (lfpsqp_basis.SA; A = [S | one dense column] in
    the second case, a general operator behind the callback in the third) against the dense basis Z = A W and the oracle: same
    iteration count, same iterate and multipliers."""
    ctx = dev_ctx
    emu = _is_emu(ctx)
    n, m_lin, k = (2600 if emu else 300_000), 9, 3
    extra = 1 if case == "extra_column" else 0
    m = m_lin + extra
    rows, cols, vals = banded(n, m_lin, k, seed=31)
    Ah = np.zeros((n, m), order='F')
    np.add.at(Ah, (rows, cols), vals)
    if extra:
        Ah[:, m_lin] = synth.hash_vector(12, n)
    A = ctx.matrix(n, m, Ah)
    S = L.SparseMatrix(ctx, n, m_lin, rows, cols, vals)
    Z, W = ctx.matrix(n, m), np.zeros((m, m), order='F')
    Sg, Vt, rank = L.ksvd_(A, Z, W=W, Jsp=S)
    assert rank == m
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    bh = synth.hash_vector(4, n)
    Zh = Z.download()
    if case == "operator_callback":
        e = 0.7 * synth.hash_vector(15, n)[:n - 1]
        av, up, dn = ctx.vector(n, a), ctx.vector(n, np.concatenate([e, [0.0]])), ctx.vector(n, np.concatenate([[0.0], e]))
        sh, tt = ctx.vector(n), ctx.vector(n)

        class Tri:
            def mul_(self, dest, v, al=None, be=None):
                L.vmul(av, v, dest)
                sh.fill(0.0); sh.copy_range_from(v, n - 1, 0, 1)
                L.vmul(up, sh, tt); L.axpby(1.0, tt, 1.0, dest)
                sh.fill(0.0); sh.copy_range_from(v, n - 1, 1, 0)
                L.vmul(dn, sh, tt); L.axpby(1.0, tt, 1.0, dest)
                return dest

            def adjoint(self):
                return self

        class TriRef:
            def mul_(self, dest, v, al=None, be=None):
                t = a * v
                t[:-1] += e * v[1:]
                t[1:] += e * v[:-1]
                dest[:] = t if al is None else al * t + be * dest
                return dest

            def adjoint(self):
                return self
        Aop, Aref = Tri(), TriRef()
    else:
        from tests.helpers import DiagOpRef
        Aop, Aref = L.DiagOperator(0.0, ctx.vector(n, a)), DiagOpRef(a)
    x0, l0 = np.zeros(n), np.zeros(m)
    it0, nr0 = R.projcg_(x0, l0, Aref, np.asfortranarray(Zh), bh, np.zeros(m), tol=1e-10, maxit=400)
    res = {}
    for label, basis in (("dense", L.DeviceBasis(Z)), ("nonzeros", L.DeviceBasis(Z, generator=(A, W), sparse=S))):
        x, lam = ctx.vector(n), ctx.vector(m)
        it, nr = L.projcg_(x, lam, Aop, basis, ctx.vector(n, bh), None, tol=1e-10, maxit=400)
        res[label] = (it, nr, x.download(), lam.download())
    for label in res:
        it, nr, xh, lh = res[label]
        assert it == it0, (label, it, it0)
        assert np.linalg.norm(xh - x0) <= 1e-9 * np.linalg.norm(x0), label
        assert np.abs(lh - l0).max() <= 1e-9 * max(np.abs(l0).max(), 1.0), label
    assert np.abs(Zh.T @ res["nonzeros"][2]).max() <= 1e-10 * np.linalg.norm(res["nonzeros"][2])     # the iterate is in the null space of U'


@pytest.mark.parametrize("extra", [0, 1])
def test_factored_basis_products_are_the_dense_ones(dev_ctx, extra):
    """mul!(t, U', v) and mul!(y, U, t, alpha, beta) with U = A W applied on the nonzeros of A (lfpsqp_q_gemv_t / _n with lfpsqp_basis.SA:
    the tangent projection of optimize for sparse constraint gradients) against the products with the dense basis Z."""
    ctx = dev_ctx
    n, m_lin, k = (2200 if _is_emu(ctx) else 200_000), 11, 3
    m = m_lin + extra
    rows, cols, vals = banded(n, m_lin, k, seed=41)
    Ah = np.zeros((n, m), order='F')
    np.add.at(Ah, (rows, cols), vals)
    if extra:
        Ah[:, m_lin] = synth.hash_vector(13, n)
    A = ctx.matrix(n, m, Ah)
    S = L.SparseMatrix(ctx, n, m_lin, rows, cols, vals)
    Z, W = ctx.matrix(n, m), np.zeros((m, m), order='F')
    _, _, rank = L.ksvd_(A, Z, W=W, Jsp=S)
    assert rank == m
    Ud, Uf = L.DeviceBasis(Z), L.DeviceBasis(Z, generator=(A, W), sparse=S)
    v = ctx.vector(n).hash_fill(7)
    td, tf = ctx.vector(m), ctx.vector(m)
    Ud.adjoint().mul_(td, v); Uf.adjoint().mul_(tf, v)
    np.testing.assert_allclose(tf.download(), td.download(), rtol=0, atol=1e-12 * np.abs(td.download()).max())
    yd, yf = ctx.vector(n).hash_fill(8), ctx.vector(n).hash_fill(8)
    Ud.mul_(yd, td, -1.5, 0.5); Uf.mul_(yf, td, -1.5, 0.5)
    assert np.abs(yf.download() - yd.download()).max() <= 1e-13 * np.abs(yd.download()).max() * 10
