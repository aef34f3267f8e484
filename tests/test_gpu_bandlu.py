"""GPU: band LU with partial pivoting (K16f, csrc/sx_bandlu.hip) against LAPACK through scipy on the same
matrices -- solves with one and many right-hand sides, both orientations, pivoting that has to swap, columns
without a usable pivot (replaced by unit vectors and reported).  The reference has no counterpart (its basis
factorisations are inside Gurobi): the check is the linear algebra itself, 1e-9 relative to the solution."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def band_matrix(n, kl, ku, seed, density=0.5, weak_diag=True):
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(n), kl + ku + 1)
    j = i + np.tile(np.arange(-kl, ku + 1), n)
    keep = (j >= 0) & (j < n) & ((rng.random(i.size) < density) | (i == j))    # (the diagonal is always there)
    i, j = i[keep], j[keep]
    v = rng.uniform(-1, 1, i.size)
    if weak_diag:                       # small diagonal entries: partial pivoting has to swap rows
        v[i == j] *= 1e-3
    A = sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()
    A.sum_duplicates()
    return A


def factor(ctx, A, kl, ku, tol=1e-11):
    from smart_crossover.hip.device import BandLU
    C = sp.coo_matrix(A)
    put = lambda v, t: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    lu = BandLU(ctx, A.shape[0], kl, ku, put(C.row, np.int32), put(C.col, np.int32), put(C.data, np.float64))
    rep, piv = lu.factor(tol)
    return lu, rep, piv


def with_replaced_columns(A, rep, piv):
    """The matrix the factors stand for: replaced column j = unit vector of the row that sat on the diagonal at
    step j (the swaps up to j - 1 applied to the identity)."""
    n = A.shape[0]
    rowof = np.arange(n)
    B = sp.lil_matrix(A)
    for j in range(n):
        if rep[j]:
            B[:, j] = 0
            B[rowof[j], j] = 1.0
        else:
            p = piv[j]
            rowof[j], rowof[p] = rowof[p], rowof[j]
    return sp.csc_matrix(B)


@pytest.mark.parametrize("n,kl,ku,seed", [(50, 3, 2, 0), (700, 40, 25, 1), (3000, 130, 90, 2), (2500, 600, 300, 3),
                                          (1300, 1100, 200, 4)])
def test_solves_match_lapack(ctx, n, kl, ku, seed):
    A = band_matrix(n, kl, ku, seed)
    lu, rep, piv = factor(ctx, A, kl, ku)
    assert rep.sum() == 0 and np.any(piv != np.arange(n))
    rng = np.random.default_rng(seed + 10)
    Ad = A.toarray()
    for nrhs in (1, 19):
        B = rng.standard_normal((n, nrhs))
        for trans in (False, True):
            X = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
            lu.solve(X, nrhs, n, trans)
            got = X.download().reshape((n, nrhs), order="F")
            want = np.linalg.solve(Ad.T if trans else Ad, B)
            scale = np.abs(want).max()
            assert np.abs(got - want).max() <= 1e-9 * scale * max(1.0, np.linalg.cond(Ad) * 1e-7), (nrhs, trans)
            resid = (Ad.T if trans else Ad) @ got - B
            assert np.abs(resid).max() <= 1e-9 * (1 + scale * np.abs(Ad).sum(axis=1).max())
    lu.free()


def test_columns_without_a_pivot_are_replaced_and_reported(ctx):
    n, kl, ku = 900, 30, 20
    A = sp.lil_matrix(band_matrix(n, kl, ku, 7, weak_diag=False))
    dead = [5, 6, 300, 431, 899]
    for j in dead:
        A[:, j] = 0                        # empty columns
    A[:, 100] = 0
    A[:, 99] = 0
    rows = np.arange(100 - ku, 99 + kl + 1)  # rows both columns may hold
    vals = np.random.default_rng(1).uniform(0.5, 1.0, rows.size)
    A[rows, 99] = vals
    A[rows, 100] = 2.0 * vals              # a dependent column: its pivot candidates cancel to ~1e-16
    A = sp.csr_matrix(A)
    lu, rep, piv = factor(ctx, A, kl, ku, tol=1e-9)
    assert set(np.flatnonzero(rep)) >= set(dead) and rep.sum() == len(dead) + 1
    assert rep[99] + rep[100] == 1
    B = with_replaced_columns(A, rep, piv).toarray()
    assert np.linalg.matrix_rank(B) == n and np.linalg.cond(B) < 1e10   # the repaired matrix is what was factored: non-singular
    rng = np.random.default_rng(3)
    R = rng.standard_normal((n, 9))
    for trans in (False, True):
        X = ctx.to_device(np.asfortranarray(R).ravel(order="F"))
        lu.solve(X, 9, n, trans)
        got = X.download().reshape((n, 9), order="F")
        want = np.linalg.solve(B.T if trans else B, R)
        assert np.abs(got - want).max() <= 1e-8 * np.abs(want).max()
    lu.free()


def test_entries_outside_the_band_are_refused(ctx):
    from smart_crossover.hip.device import BandLU
    put = lambda v, t: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    with pytest.raises(Exception):
        BandLU(ctx, 10, 1, 1, put([0, 5], np.int32), put([0, 1], np.int32), put([1.0, 2.0], np.float64))


def dominant_band(n, kl, ku, seed, swaps):
    """A band matrix whose inverse decays away from the diagonal (own entry ~1, the others small), as the bases of the
    sparse crossover are; `swaps`: a tenth of the rows change places with their neighbour below, so that the
    factorisation has to swap them back (same conditioning; one more sub- and super-diagonal)."""
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(n), kl + ku + 1)
    j = i + np.tile(np.arange(-kl, ku + 1), n)
    keep = (j >= 0) & (j < n) & ((rng.random(i.size) < min(0.2, 12.0 / (kl + ku))) | (i == j))   # ~12 small entries per row
    i, j = i[keep], j[keep]
    v = rng.uniform(-0.1, 0.1, i.size)
    v[i == j] = rng.uniform(0.8, 1.2, int((i == j).sum()))
    if swaps:
        perm = np.arange(n)
        for r in np.flatnonzero(rng.random(n - 1) < 0.1)[::2]:
            if perm[r] == r and perm[r + 1] == r + 1:
                perm[r], perm[r + 1] = r + 1, r
        i = perm[i]
    return sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()


@pytest.mark.parametrize("swaps", [False, True])
@pytest.mark.parametrize("n,kl,ku", [(6000, 40, 30), (20000, 117, 113)])
def test_sparse_right_hand_sides(ctx, monkeypatch, n, kl, ku, swaps):
    """solve_sparse (panels without entries skipped): with tiny = 0 the result of the plain solve's SEQUENTIAL sweeps bit
    for bit (SX_BANDLU_SEQ: up to 1,024 dense right-hand sides otherwise take the partitioned sweeps, which round
    differently), with the crossover's 1e-60 equal to 1e-50; right-hand sides with entries in one place, in two far apart,
    at both ends, none."""
    monkeypatch.setenv("SX_BANDLU_SEQ", "1")
    A = dominant_band(n, kl, ku, 5, swaps)
    kl, ku = kl + int(swaps), ku + int(swaps)
    lu, rep, piv = factor(ctx, A, kl, ku)
    assert rep.sum() == 0 and np.any(piv != np.arange(n)) == swaps
    rng = np.random.default_rng(6)
    nrhs = 27
    B = np.zeros((n, nrhs))
    for t in range(nrhs - 4):
        at = rng.integers(0, n)
        rows = np.clip(at + rng.integers(-20, 21, 6), 0, n - 1)
        B[rows, t] = rng.uniform(-1, 1, 6)
    B[[3, n - 2], nrhs - 4] = [1.0, -2.0]            # both ends
    B[[n // 5, 4 * n // 5], nrhs - 3] = [0.5, 0.25]  # two places far apart
    B[n - 1, nrhs - 2] = 1.0                         # the last row only; the last right-hand side stays empty
    ref = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
    lu.solve(ref, nrhs, n, False)
    want = ref.download().reshape((n, nrhs), order="F")
    Ad = sp.csc_matrix(A)
    assert np.abs(Ad @ want - B).max() <= 1e-12
    for tiny, tol in ((0.0, 0.0), (1e-60, 1e-50)):
        X = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
        lu.solve_sparse(X, nrhs, n, tiny)
        got = X.download().reshape((n, nrhs), order="F")
        assert np.abs(got - want).max() <= tol, tiny
    assert np.all(want[:, nrhs - 1] == 0.0)
    lu.free()


@pytest.mark.parametrize("swaps", [False, True])
def test_partitioned_sweeps_agree_with_the_sequential_ones(ctx, monkeypatch, swaps):
    """Up to 8 dense right-hand sides take the partitioned sweeps (blocks of panels side by side + a chain of small
    products, csrc/sx_bandlu.hip k_gbp_*); SX_BANDLU_SEQ keeps a handle on the sequential ones.  Same solutions to
    rounding, both orientations, and small residuals."""
    n, kl, ku = 20000, 117, 113
    A = dominant_band(n, kl, ku, 7, swaps)
    kl, ku = kl + int(swaps), ku + int(swaps)
    monkeypatch.setenv("SX_BANDLU_SEQ", "1")
    seq, rep, piv = factor(ctx, A, kl, ku)
    rng = np.random.default_rng(8)
    M = sp.csc_matrix(A)
    want = {}
    for nrhs in (1, 5, 8):
        B = rng.standard_normal((n, nrhs))
        for trans in (False, True):
            X = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
            seq.solve(X, nrhs, n, trans)
            want[(nrhs, trans)] = (B, X.download().reshape((n, nrhs), order="F"))
    monkeypatch.delenv("SX_BANDLU_SEQ")
    par, rep2, piv2 = factor(ctx, A, kl, ku)
    assert np.array_equal(piv, piv2)
    for (nrhs, trans), (B, x_seq) in want.items():
        X = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
        par.solve(X, nrhs, n, trans)
        got = X.download().reshape((n, nrhs), order="F")
        scale = np.abs(x_seq).max()
        assert np.abs(got - x_seq).max() <= 1e-11 * scale, (nrhs, trans)
        assert np.abs((M.T if trans else M) @ got - B).max() <= 1e-11 * (1 + scale), (nrhs, trans)
    seq.free()
    par.free()


@pytest.mark.parametrize("nblocks,real,last,kl,ku", [(4, 640, 500, 40, 25), (7, 1280, 1290, 117, 113), (3, 96, 33, 5, 3)])
def test_blocks_factored_side_by_side_match_lapack(ctx, nblocks, real, last, kl, ku):
    """factor_blocks: a block-diagonal band matrix with identity padding of kl + ku + 32 positions (rounded up to whole
    panels) between its blocks, the blocks' panels factored side by side -- pivoting has to swap rows inside every block;
    the solves (both orientations, one and many right-hand sides) against LAPACK, and the same factors as the sequential
    factorisation of the same matrix bit for bit."""
    from scipy.sparse.linalg import splu
    from smart_crossover.hip.device import BandLU
    pad = (kl + ku + 32 + 31) // 32 * 32
    stride = real + pad
    n = (nblocks - 1) * stride + last
    blocks, rows, cols, vals = [], [], [], []
    for b in range(nblocks):
        nb_ = real if b < nblocks - 1 else last
        Ab = sp.coo_matrix(band_matrix(nb_, min(kl, nb_ - 1), min(ku, nb_ - 1), 100 + b, density=1.0))   # (a full band: no block is singular)
        rows.append(Ab.row + b * stride); cols.append(Ab.col + b * stride); vals.append(Ab.data)
        if b < nblocks - 1:
            idx = np.arange(b * stride + real, (b + 1) * stride)
            rows.append(idx); cols.append(idx); vals.append(np.ones(idx.size))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    C = sp.coo_matrix(A)
    put = lambda v, t: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    lu = BandLU(ctx, n, kl, ku, put(C.row, np.int32), put(C.col, np.int32), put(C.data, np.float64))
    rep, piv = lu.factor_blocks(nblocks, stride, real, last)
    assert rep.sum() == 0 and np.any(piv != np.arange(n))
    seq = BandLU(ctx, n, kl, ku, put(C.row, np.int32), put(C.col, np.int32), put(C.data, np.float64))
    rep2, piv2 = seq.factor()
    assert np.array_equal(piv, piv2) and np.array_equal(rep, rep2)
    ref = splu(sp.csc_matrix(A))
    rng = np.random.default_rng(3)
    for nrhs in (1, 19):
        B = rng.standard_normal((n, nrhs))
        for trans in (False, True):
            want = ref.solve(B, trans="T" if trans else "N")
            for h in (lu, seq):
                X = ctx.to_device(np.asfortranarray(B).ravel(order="F"))
                h.solve(X, nrhs, n, trans)
                got = X.download().reshape((n, nrhs), order="F")
                assert np.abs(got - want).max() <= 1e-9 * (1 + np.abs(want).max())
    lu.free()
    seq.free()
