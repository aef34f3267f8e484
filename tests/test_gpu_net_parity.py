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
