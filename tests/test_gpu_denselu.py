"""GPU: dense LU with partial pivoting (K16g, csrc/sx_denselu.hip) against LAPACK through numpy on the same matrices:
solutions of both orientations with one and many right-hand sides, sizes that end inside a panel / a block, the
row permutation, and the repair of columns without a usable pivot (replaced by the unit vector of the row on their
diagonal -- what the bordered basis of the sparse crossover relies on).  The reference leaves basis factorisations to its
solvers (solver_caller/gurobi.py:202-210): parity with them is unpinned; LAPACK is the yardstick."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def factor(ctx, A, tol=1e-11):
    from smart_crossover.hip.device import DenseLU
    n = A.shape[0]
    lu = DenseLU(ctx, n, ctx.to_device(np.asfortranarray(A).ravel(order="F")))
    rep, perm = lu.factor(tol)
    return lu, rep, perm


def solve(ctx, lu, R, trans, ldx=None):
    n, k = R.shape
    ldx = ldx or n
    buf = np.zeros((ldx, k), order="F")
    buf[:n] = R
    X = ctx.to_device(buf.ravel(order="F"))
    lu.solve(X, k, ldx, trans)
    return X.download().reshape((ldx, k), order="F")[:n]


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 65, 130, 500, 1037, 2500])
def test_solutions_match_lapack(ctx, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    lu, rep, perm = factor(ctx, A)
    assert rep.sum() == 0 and sorted(perm) == list(range(n))
    for k in (1, 5, 70):
        R = rng.standard_normal((n, k))
        for trans in (False, True):
            got = solve(ctx, lu, R, trans, ldx=n + (3 if k == 5 else 0))
            want = np.linalg.solve(A.T if trans else A, R)
            assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()) * max(1.0, np.linalg.cond(A) / 1e4)
    lu.free()


def test_partial_pivoting_takes_the_largest_entry(ctx):
    """A matrix that needs its rows swapped at every step (tiny diagonal): without pivoting the solution is garbage."""
    n = 300
    rng = np.random.default_rng(2)
    A = rng.standard_normal((n, n))
    A[np.arange(n), np.arange(n)] = 1e-14
    lu, rep, perm = factor(ctx, A)
    assert rep.sum() == 0
    assert np.count_nonzero(perm != np.arange(n)) > n // 2
    R = rng.standard_normal((n, 3))
    got = solve(ctx, lu, R, False)
    assert np.abs(A @ got - R).max() <= 1e-9 * np.abs(got).max()
    lu.free()


def test_columns_without_a_pivot_are_replaced(ctx):
    n = 200
    rng = np.random.default_rng(3)
    A = rng.standard_normal((n, n))
    A[:, 17] = 0.0                              # an empty column
    A[:, 90] = 2.0 * A[:, 40] - A[:, 41]        # a dependent one (its candidates cancel to ~1e-15)
    A[:, 150] = A[:, 149]                       # and a duplicate
    lu, rep, perm = factor(ctx, A, tol=1e-9)
    assert set(np.flatnonzero(rep)) == {17, 90, 150}
    B = A.copy()
    for j in np.flatnonzero(rep):               # the matrix that was factored: column j = unit vector of row perm[j]
        B[:, j] = 0.0
        B[perm[j], j] = 1.0
    assert np.linalg.matrix_rank(B) == n
    R = rng.standard_normal((n, 4))
    for trans in (False, True):
        got = solve(ctx, lu, R, trans)
        want = np.linalg.solve(B.T if trans else B, R)
        assert np.abs(got - want).max() <= 1e-8 * np.abs(want).max()
    lu.free()


def test_deterministic(ctx):
    n = 700
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n))
    R = rng.standard_normal((n, 9))
    outs = []
    for _ in range(2):
        lu, rep, perm = factor(ctx, A)
        outs.append(solve(ctx, lu, R, False).view(np.uint64).copy())
        lu.free()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("n", [6000, 12000])
def test_large_schur_complement_sized_matrices(ctx, n):
    """The size of the border at config-5 size (1e4 linking rows + separators): residual test, no LAPACK run on the host."""
    rng = np.random.default_rng(7)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    A[np.arange(n), rng.permutation(n)] += 2.0            # a dominant entry per column, at a permuted position
    lu, rep, perm = factor(ctx, A)
    assert rep.sum() == 0
    R = rng.standard_normal((n, 3))
    for trans in (False, True):
        got = solve(ctx, lu, R, trans)
        res = (A.T if trans else A) @ got - R
        assert np.abs(res).max() <= 1e-9 * np.abs(got).max() * n ** 0.5
    lu.free()
