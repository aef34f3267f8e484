"""GPU: the network simplex of libsxhip.so (K16n, csrc/sx_netsimplex.hip) behind ``solver="HIP"`` for the re-solves
of the network crossover (reference: network_methods/net_manager.py:211-222 -> solve_mcf with a warm basis).
Checks are solver-independent certificates (primal / dual feasibility, complementary slackness, tree basis) plus
the optimal value against HiGHS at 1e-9."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def big_m_network(V, E, seed, inf_frac=0.3):
    """Random network with an artificial root (node V) and one big-M arc per node: the all-artificial star is a
    primal feasible spanning-tree basis (what MCFManagerStd.extend_by_bigM / set_initial_basis build)."""
    rng = np.random.default_rng(seed)
    tail = rng.integers(0, V, E)
    head = (tail + 1 + rng.integers(0, V - 1, E)) % V
    cost = rng.integers(1, 50, E).astype(float)
    cap = rng.integers(1, 10, E).astype(float)
    cap[rng.random(E) < inf_frac] = np.inf
    b = rng.integers(-5, 6, V).astype(float)
    b[-1] -= b.sum()
    out = b >= 0
    a_tail = np.where(out, np.arange(V), V)
    a_head = np.where(out, V, np.arange(V))
    tail, head = np.concatenate([tail, a_tail]), np.concatenate([head, a_head])
    big = (V + 1) * cost.max()
    cost = np.concatenate([cost, np.full(V, big)])
    cap = np.concatenate([cap, np.full(V, np.inf)])
    n = E + V
    arcs = np.arange(n)
    A = sp.csr_matrix((np.concatenate([np.ones(n), -np.ones(n)]), (np.concatenate([tail, head]), np.concatenate([arcs, arcs]))),
                      shape=(V + 1, n))
    A.sort_indices()
    b = np.concatenate([b, [0.0]])
    vb = np.concatenate([np.full(E, -1), np.zeros(V)]).astype(np.int8)
    cb = np.concatenate([np.full(V, -1), [0]]).astype(np.int8)
    return A, b, cost, cap, tail, head, vb, cb


def run(ctx, A, b, c, u, vb, cb, **kw):
    m, n = A.shape
    dA = ctx.matrix(A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    d_vb, d_cb = ctx.empty(n, np.int8), ctx.empty(m, np.int8)
    res = ctx.net_simplex(dA, put(b), put(c), put(np.zeros(n)), put(u), put(vb, np.int8), put(cb, np.int8), x=d_x, y=d_y,
                          vbasis_out=d_vb, cbasis_out=d_cb, opt_tol=1e-9, **kw)
    out = (res, d_x.download(), d_y.download(), d_vb.download().astype(int), d_cb.download().astype(int))
    dA.free()
    return out


def certificates(A, b, c, u, tail, head, x, y, vb, cb, tol=1e-7):
    m, n = A.shape
    assert np.abs(A @ x - b).max() <= tol * (1 + np.abs(b).max())
    assert x.min() >= -tol and (x - u).max() <= tol
    rc = c - A.T @ y
    basic = vb == 0
    assert np.abs(rc[basic]).max() <= 1e-9 * (1 + np.abs(c).max())          # tree arcs price out exactly
    assert rc[vb == -1].min(initial=0.0) >= -1e-7 and rc[vb == -2].max(initial=0.0) <= 1e-7
    assert np.all(x[vb == -1] == 0.0) and np.all(x[vb == -2] == u[vb == -2])
    # basis = spanning tree + the root row's logical
    assert basic.sum() == m - 1 and (cb == 0).sum() == 1
    g = sp.coo_matrix((np.ones(basic.sum()), (tail[basic], head[basic])), shape=(m, m))
    ncomp, _ = sp.csgraph.connected_components(g, directed=False)
    assert ncomp == 1
    assert y[np.flatnonzero(cb == 0)[0]] == 0.0


@pytest.mark.parametrize("V,E,seed", [(12, 40, 0), (60, 400, 1), (300, 3000, 2), (1500, 12000, 3)])
def test_from_the_artificial_star_to_the_optimum(ctx, V, E, seed):
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, seed)
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
    assert int(res.status) == 0 and int(res.iters) > 0
    certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
    ref = linprog(c, A_eq=A, b_eq=b, bounds=list(zip(np.zeros(c.size), [None if np.isinf(v) else v for v in u])), method="highs")
    assert ref.status == 0
    assert float(res.obj) == pytest.approx(ref.fun, rel=1e-9, abs=1e-9)
    assert float(c @ x) == pytest.approx(float(res.obj), rel=1e-12)
    # the optimal basis handed back is a fixed point: zero pivots from it, same vertex
    res2, x2, y2, vb2, cb2 = run(ctx, A, b, c, u, vbo.astype(np.int8), cbo.astype(np.int8))
    assert int(res2.status) == 0 and int(res2.iters) == 0
    np.testing.assert_allclose(x2, x, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(vb2, vbo)


def test_every_block_size_reaches_the_same_value(ctx):
    A, b, c, u, tail, head, vb, cb = big_m_network(400, 5000, 7)
    vals = []
    for k in (1, 2, 8, 64):
        ctx.set_option("ns_block", k)
        res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
        assert int(res.status) == 0
        certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
        vals.append(float(res.obj))
    ctx.set_option("ns_block", 0)
    assert max(vals) - min(vals) <= 1e-9 * abs(vals[0])


def test_runs_are_reproducible(ctx):
    A, b, c, u, tail, head, vb, cb = big_m_network(500, 6000, 11)
    a = run(ctx, A, b, c, u, vb, cb)
    bb = run(ctx, A, b, c, u, vb, cb)
    assert int(a[0].iters) == int(bb[0].iters)
    assert a[1].tobytes() == bb[1].tobytes() and a[2].tobytes() == bb[2].tobytes()
    np.testing.assert_array_equal(a[3], bb[3])


def test_outside_the_domain_is_reported_not_solved(ctx):
    A, b, c, u, tail, head, vb, cb = big_m_network(40, 200, 5)
    # (a) a basis that is no spanning tree: one artificial arc out, an arbitrary arc in (may close a cycle) -- use a
    #     parallel pair to be sure: two arcs between the same nodes cannot both be in a tree with all the others
    vb_bad = vb.copy()
    vb_bad[0] = 0                       # V + 1 arcs coded basic
    assert int(run(ctx, A, b, c, u, vb_bad, cb)[0].status) == 5
    # (b) two root rows
    cb_bad = cb.copy()
    cb_bad[0] = 0
    assert int(run(ctx, A, b, c, u, vb, cb_bad)[0].status) == 5
    # (c) a tree whose flows violate a bound: artificial arcs turned the wrong way round
    A2 = A.tolil(copy=True)
    n = A.shape[1]
    V = A.shape[0] - 1
    j = n - V                           # the artificial arc of node 0
    A2[:, j] = -A2[:, j]
    res = run(ctx, sp.csr_matrix(A2), b, c, u, vb, cb)[0]
    assert int(res.status) == 5 or b[0] == 0
    # (d) not a network
    A3 = A.tolil(copy=True)
    A3[1, 0] = 2.0
    assert int(run(ctx, sp.csr_matrix(A3), b, c, u, vb, cb)[0].status) == 5


def test_unbounded_is_detected(ctx):
    # a negative cycle of uncapacitated arcs
    V = 3
    tail = np.array([0, 1, 2, 0, 1, 2])
    head = np.array([1, 2, 0, 3, 3, 3])
    c = np.array([-1.0, -1.0, -1.0, 10.0, 10.0, 10.0])
    u = np.full(6, np.inf)
    arcs = np.arange(6)
    A = sp.csr_matrix((np.concatenate([np.ones(6), -np.ones(6)]), (np.concatenate([tail, head]), np.concatenate([arcs, arcs]))),
                      shape=(4, 6))
    b = np.zeros(4)
    vb = np.array([-1, -1, -1, 0, 0, 0], dtype=np.int8)
    cb = np.array([-1, -1, -1, 0], dtype=np.int8)
    assert int(run(ctx, A, b, c, u, vb, cb)[0].status) == 2


@pytest.mark.parametrize("V,E,netdual", [(512, 4096, -1), (512, 4096, 0), (4096, 32768, -1), (4096, 32768, 0)])
def test_cnet_mcf_runs_on_the_network_simplex_and_agrees_with_highs(ctx, V, E, netdual, capsys):
    """The whole CNET_MCF crossover with solver='HIP': every round's re-solve is a network LP with a warm tree
    basis (round 0: the artificial star) -> the dual network simplex K16d, or with "netdual" 0 the primal one
    K16n.  Same optimal cost as the HiGHS-backed run."""
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller import hip as hipmod
    costs = {}
    used = []
    orig = hipmod.HipCaller._solve

    def spy(self):
        orig(self)
        used.append(self.solved_by)

    hipmod.HipCaller._solve = spy
    ctx.set_option("netdual", netdual)
    try:
        for solver in ("HIP", "HGS"):
            inst = workloads.mcf(V, E, 3)
            mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
            out = network_crossover(inst.x.copy(), mcf=mcf, method="cnet_mcf", solver=solver)
            costs[solver] = float(inst.c @ out.x[:E]) if out.x.size >= E else float(out.obj_val)
    finally:
        hipmod.HipCaller._solve = orig
        ctx.set_option("netdual", -1)
    capsys.readouterr()
    assert used and all(how == ("netsimplex" if netdual == 0 else "netdual") for how in used)
    assert costs["HIP"] == pytest.approx(costs["HGS"], rel=1e-9)


def test_composite_backend_keeps_the_network_route(ctx, capsys):
    """'HGS+HIP': ``SplitCaller`` forwards ``read_mcf`` to both backends, so the device still sees a network and the
    re-solves of CNET_MCF go to K16d / K16n -- not to the general simplex with its dense inverse."""
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller import hip as hipmod
    used = []
    orig = hipmod.HipCaller._solve

    def spy(self):
        orig(self)
        used.append(self.solved_by)

    hipmod.HipCaller._solve = spy
    try:
        inst = workloads.mcf(512, 4096, 3)
        mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
        out = network_crossover(inst.x.copy(), mcf=mcf, method="cnet_mcf", solver="HGS+HIP")
    finally:
        hipmod.HipCaller._solve = orig
    capsys.readouterr()
    assert out.x is not None and out.basis is not None
    assert used and all(how in ("netdual", "netsimplex") for how in used)
