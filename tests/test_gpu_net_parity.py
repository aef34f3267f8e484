"""GPU parity: network-side kernels (K7 MCF indicators, K8 OT indicators, K9 ranking, OT pricing)
against the reference goldens and the oracle.  Indicators and reduced costs bit-exact; the ranking
bit-exact against the library's stated tie rule and tie-class equal to the reference's queue."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal, csr_from
from oracle import net_path as N
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def device_mcf_indicators(ctx, A, x, u):
    A = sp.csr_matrix(A).copy()
    A.sum_duplicates()
    A.sort_indices()
    dA = ctx.matrix(A)
    V, E = A.shape
    ind, xhat, f = ctx.empty(E, np.float64), ctx.empty(E, np.float64), ctx.empty(V, np.float64)
    ctx.flow_indicator_mcf(dA, ctx.to_device(x), ctx.to_device(u), ind, xhat, f)
    out = ind.download(), xhat.download(), f.download()
    dA.free()
    return out


def test_mcf_indicators_match_reference_golden(ctx, g3):
    A = csr_from(g3, "A")
    ind, xhat, f = device_mcf_indicators(ctx, A, g3["x"], g3["u"])
    assert bits_equal(ind, g3["ind"])
    _, mid = N.mcf_flow_indicators(A, g3["x"], g3["u"])
    assert bits_equal(xhat, mid["x_hat"]) and bits_equal(f, mid["f"])
    q = ctx.argsort_desc(ctx.to_device(ind)).download()
    assert np.array_equal(q, N.rank_desc(g3["ind"]))
    assert N.same_up_to_ties(g3["ind"], q, g3["queue_ref"])


@pytest.mark.parametrize("V,E,seed", [(4096, 32768, 33), (300, 5000, 1), (2 ** 15, 2 ** 18, 3)])
def test_mcf_indicators_match_oracle(ctx, V, E, seed):
    mi = workloads.mcf(V, E, seed=seed)
    x = mi.x.copy()
    rng = np.random.default_rng(seed)
    x[rng.integers(0, E, 20)] = -0.5                      # outside [0, u] -> x_hat = 0
    x[rng.integers(0, E, 20)] = mi.u[0] * 5000.0
    ind, xhat, f = device_mcf_indicators(ctx, mi.A, x, mi.u)
    want, mid = N.mcf_flow_indicators(mi.A, x, mi.u)
    assert bits_equal(xhat, mid["x_hat"]) and bits_equal(f, mid["f"]) and bits_equal(ind, want)
    assert np.count_nonzero(want) > E // 10


def test_mcf_indicators_general_weights(ctx):
    rng = np.random.default_rng(5)
    V, E = 70, 900
    A = sp.random(V, E, density=0.06, random_state=3, format="csr")
    A.data = rng.choice([-2.0, -1.0, 1.0, 3.0, 0.25], size=A.nnz)
    u = rng.uniform(1, 5, E)
    x = rng.uniform(-0.2, 1.2, E) * u
    ind, xhat, f = device_mcf_indicators(ctx, A, x, u)
    want, mid = N.mcf_flow_indicators(A, x, u)
    assert bits_equal(ind, want) and bits_equal(f, mid["f"])


def test_ot_indicators_and_ranking_match_reference_golden(ctx, g4):
    S, D = g4["M"].shape
    ind = ctx.empty(S * D, np.float64)
    ctx.flow_indicator_ot(S, D, ctx.to_device(g4["x"]), ctx.to_device(g4["s"]), ctx.to_device(g4["d"]), ind)
    got = ind.download()
    assert bits_equal(got, g4["ind"])
    q = ctx.argsort_desc(ind).download()
    assert np.array_equal(q, N.rank_desc(g4["ind"]))
    assert N.same_up_to_ties(g4["ind"], q, g4["queue_ref"])


def test_ot_indicators_config3_size(ctx):
    inst = workloads.config3()
    S = D = 784
    ind = ctx.empty(S * D, np.float64)
    ctx.flow_indicator_ot(S, D, ctx.to_device(inst.x), ctx.to_device(inst.s), ctx.to_device(inst.d), ind)
    want = N.ot_flow_indicators(inst.x, inst.s, inst.d)
    assert bits_equal(ind.download(), want)
    q = ctx.argsort_desc(ind).download()
    assert np.array_equal(q, N.rank_desc(want))


def test_ot_pricing_matches_reference_golden(ctx, g4):
    M1 = g4["M1"]
    S1, D1 = M1.shape
    rc = ctx.empty(S1 * D1, np.float64)
    res = ctx.price_ot(S1, D1, ctx.to_device(M1.ravel()), ctx.to_device(g4["y"]), 1e-6, rc)
    got = rc.download()
    assert bits_equal(got, g4["rc"])
    mn, am, bad = ctx.read_price(res)
    assert mn == got.min() and am == int(np.argmin(got)) and bad == int(np.count_nonzero(~(got >= -1e-6)))
    res = ctx.price_ot(S1, D1, ctx.to_device(M1.ravel()), ctx.to_device(np.zeros_like(g4["y"])), 1e-6, None)
    assert (ctx.read_price(res)[2] == 0) == bool(g4["opt_flag_zero_y"])


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 2048, 2049, 100_003, 1_048_576])
def test_argsort_desc_sizes_and_ties(ctx, n):
    rng = np.random.default_rng(n)
    key = rng.random(n)
    key[rng.random(n) < 0.5] = 0.0                          # the typical case: half of the arcs carry no flow
    key[rng.random(n) < 0.1] = 0.25
    if n > 10:
        key[3], key[7], key[5] = np.inf, -1.5, -0.0
    q = ctx.argsort_desc(ctx.to_device(key)).download()
    assert q.dtype == np.int64
    assert np.array_equal(q, N.rank_desc(key))


def test_argsort_desc_special_values(ctx):
    key = np.array([0.0, -0.0, np.nan, 1e-310, -1e-310, np.inf, -np.inf, 1.0, 1.0, np.nan, -3.0, 2.0 ** -1074])
    q = ctx.argsort_desc(ctx.to_device(key)).download()
    # numpy puts NaN last in ascending order (hence first here) and treats -0.0 == +0.0 as a tie
    assert np.array_equal(q, N.rank_desc(key))
    assert set(q[:2].tolist()) == {2, 9}


# ----------------------------------------------------------------------------- K13 spanning tree
def scipy_max_spanning_tree(S, D, w):
    """The reference's route (tree_BI.py:32-59): minimum spanning tree of the negated weights."""
    from scipy.sparse import csgraph
    W = np.asarray(w, dtype=float).reshape(S, D)
    graph = sp.bmat([[None, sp.csr_matrix(-W)], [sp.csr_matrix((D, S)), None]], format="csr")
    tree = csgraph.minimum_spanning_tree(graph).tocoo()
    return np.sort(tree.row.astype(np.int64) * D + (tree.col.astype(np.int64) - S))


@pytest.mark.parametrize("S,D,seed", [(12, 15, 0), (1, 9, 1), (9, 1, 2), (64, 200, 3), (784, 784, 4)])
def test_spanning_tree_matches_scipy_on_distinct_weights(ctx, S, D, seed):
    rng = np.random.default_rng(seed)
    w = rng.permutation(S * D).astype(np.float64) + 1.0 + rng.random(S * D) * 0.5      # distinct, positive
    w *= 1e-6
    got = ctx.spanning_tree_ot(S, D, ctx.to_device(w))
    assert got.dtype == np.int64 and got.size == S + D - 1
    assert np.array_equal(got, scipy_max_spanning_tree(S, D, w))


def test_spanning_tree_golden_and_sinkhorn_like_plan(ctx, g4):
    S, D = g4["M"].shape
    assert np.array_equal(ctx.spanning_tree_ot(S, D, ctx.to_device(g4["ind"])), g4["tree_edges"])    # reference golden
    inst = workloads.config3()
    ind = N.ot_flow_indicators(inst.x, inst.s, inst.d)
    got = ctx.spanning_tree_ot(784, 784, ctx.to_device(ind))
    want = scipy_max_spanning_tree(784, 784, ind)
    assert got.size == 1567
    # equal weights may be resolved differently: compare the total weight, and the trees when all chosen weights differ
    assert ind[got].sum() == pytest.approx(ind[want].sum(), rel=1e-13)
    if np.unique(ind[want]).size == want.size:
        assert np.array_equal(got, want)


def test_spanning_tree_zero_weights_ties_and_nan(ctx):
    # two components: suppliers {0,1} x demanders {0}, supplier {2} x demanders {1,2}; zeros and NaN are not edges
    w = np.array([[3.0, 0.0, 0.0],
                  [2.0, 0.0, np.nan],
                  [0.0, 5.0, 1.0]])
    got = ctx.spanning_tree_ot(3, 3, ctx.to_device(w.ravel()))
    assert np.array_equal(got, [0, 3, 7, 8])                       # a forest with 4 < S + D - 1 arcs
    # all weights equal: the smaller arc index wins -> first row and first column
    got = ctx.spanning_tree_ot(4, 5, ctx.to_device(np.full(20, 0.25)))
    assert got.size == 8
    tree = np.zeros(20, dtype=bool)
    tree[got] = True
    T = tree.reshape(4, 5)
    assert T[0].all() and T[:, 0].all()
    assert ctx.spanning_tree_ot(3, 0, ctx.to_device(np.zeros(1))).size == 0
