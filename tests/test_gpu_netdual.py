"""GPU: the dual network simplex of libsxhip.so (K16d, csrc/sx_netdual.hip) against its CPU statement
oracle/net_simplex.py -- the re-solves of the network crossover's column generation (reference:
network_methods/net_manager.py:211-222 -> solve_mcf with a warm basis; network_methods/algorithms.py:109-140).
On integral data the device makes the oracle's pivots: same iteration count, same flips, same tree, same flows bit
for bit.  Optimal values are checked against HiGHS at 1e-9 and by solver-independent certificates."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from oracle.net_simplex import dual_network_simplex
from test_gpu_netsimplex import big_m_network, certificates

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def run(ctx, A, b, c, u, vb, cb, **kw):
    m, n = A.shape
    dA = ctx.matrix(A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    d_vb, d_cb = ctx.empty(n, np.int8), ctx.empty(m, np.int8)
    res = ctx.net_dual(dA, put(b), put(c), put(np.zeros(n)), put(u), put(vb, np.int8), put(cb, np.int8), x=d_x, y=d_y,
                       vbasis_out=d_vb, cbasis_out=d_cb, **kw)
    out = (res, d_x.download(), d_y.download(), d_vb.download().astype(int), d_cb.download().astype(int))
    dA.free()
    return out


def highs(A, b, c, u):
    ref = linprog(c, A_eq=A, b_eq=b, bounds=list(zip(np.zeros(c.size), [None if np.isinf(v) else v for v in u])), method="highs")
    assert ref.status == 0
    return ref.fun


def same_pivots(res, x, vbo, want):
    assert int(res.status) == want["status"] == 0
    assert int(res.iters) == want["iters"] and int(res.phase1_iters) == want["flips"]
    np.testing.assert_array_equal(vbo, want["vbasis"].astype(int))
    assert np.array_equal(x, want["x"])                 # integral data: flows are exact on both sides (-0.0 == 0.0)


@pytest.mark.parametrize("V,E,seed", [(12, 40, 0), (60, 400, 1), (300, 3000, 2), (1500, 12000, 3)])
def test_from_the_artificial_star_pivot_for_pivot_with_the_oracle(ctx, V, E, seed):
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, seed, inf_frac=0.0)
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
    want = dual_network_simplex(tail, head, c, u, b, vb, root=V)
    same_pivots(res, x, vbo, want)
    assert int(res.iters) > 0
    certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
    assert float(res.obj) == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)
    assert float(c @ x) == pytest.approx(float(res.obj), rel=1e-12)
    # the optimal basis handed back is a fixed point
    res2, x2, y2, vb2, cb2 = run(ctx, A, b, c, u, vbo.astype(np.int8), cbo.astype(np.int8))
    assert int(res2.status) == 0 and int(res2.iters) == 0 and int(res2.phase1_iters) == 0
    assert int(res2.warm_start_used) == 2         # ... and the tree arrays kept from the first solve were reused
    assert np.array_equal(x2, x)
    np.testing.assert_array_equal(vb2, vbo)


def test_column_generation_round_new_arcs_at_either_bound(ctx):
    """A round of the column generation: the optimal tree of a sub-problem plus the arcs it did not have, which sit at
    0 or at their capacity -- the tree is no longer primal feasible once they are flipped to their right bound."""
    V, E = 400, 6000
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, 21, inf_frac=0.0)
    rng = np.random.default_rng(5)
    n = A.shape[1]
    have = np.ones(n, dtype=bool)
    have[:E] = rng.random(E) < 0.4
    sub = np.flatnonzero(have)
    r1 = run(ctx, sp.csr_matrix(A[:, sub]), b, c[sub], u[sub], vb[sub], cb)
    assert int(r1[0].status) == 0
    vb2 = np.where(rng.random(n) < 0.3, -2, -1).astype(np.int8)    # the new arcs: at their capacity or at zero
    vb2[sub] = r1[3]
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb2, cb)
    want = dual_network_simplex(tail, head, c, u, b, vb2, root=V)
    same_pivots(res, x, vbo, want)
    assert int(res.warm_start_used) == 1          # (the arcs are numbered differently here: kept tree arrays do not apply)
    assert int(res.phase1_iters) > 0                        # arcs were moved bound to bound
    certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
    assert float(res.obj) == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)


def test_any_spanning_tree_is_a_start_and_every_grid_gives_the_same_run(ctx):
    V, E = 700, 9000
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, 33, inf_frac=0.0)
    u[E:] = 1000.0      # capacitated artificial arcs: whatever bound an arc starts at, it can be flipped
    rng = np.random.default_rng(9)
    n = A.shape[1]
    # a random spanning tree of the whole graph (Kruskal over a random arc order) and random bounds for the rest
    comp = list(range(V + 1))

    def find(i):
        while comp[i] != i:
            comp[i] = comp[comp[i]]
            i = comp[i]
        return i

    tree = np.zeros(n, dtype=bool)
    for k in rng.permutation(n):
        i, j = find(int(tail[k])), find(int(head[k]))
        if i != j:
            comp[i] = j
            tree[k] = True
    assert tree.sum() == V
    vb0 = np.where(tree, 0, np.where(rng.random(n) < 0.5, -1, -2)).astype(np.int8)
    vb0[(vb0 == -2) & np.isinf(u)] = -1
    want = dual_network_simplex(tail, head, c, u, b, vb0, root=V)
    assert want["status"] == 0
    runs = []
    for grid in (0, 1, 3, 64):
        ctx.set_option("nd_grid", grid)
        res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb0, cb)
        same_pivots(res, x, vbo, want)
        certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
        runs.append((x.tobytes(), y.tobytes(), vbo.tobytes()))
    ctx.set_option("nd_grid", 0)
    assert all(r == runs[0] for r in runs)
    assert float(c @ np.frombuffer(runs[0][0])) == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)


def test_outside_the_domain_and_infeasible(ctx):
    A, b, c, u, tail, head, vb, cb = big_m_network(40, 300, 5, inf_frac=0.0)
    V = 40
    # an uncapacitated arc whose reduced cost has the wrong sign cannot be flipped: not the dual method's start
    u2 = u.copy()
    want = dual_network_simplex(tail, head, c, u, b, vb, root=V)
    y0 = np.where(np.arange(V + 1) < V, np.where(b[:V + 1] >= 0, c[-1], -c[-1]), 0.0)
    rc = c - A.T @ y0
    wrong = np.flatnonzero((vb == -1) & (rc < 0))
    assert wrong.size > 0
    u2[wrong[0]] = np.inf
    assert int(run(ctx, A, b, c, u2, vb, cb)[0].status) == 5
    assert dual_network_simplex(tail, head, c, u2, b, vb, root=V)["status"] == 5
    # not a tree
    vb_bad = vb.copy()
    vb_bad[0] = 0
    assert int(run(ctx, A, b, c, u, vb_bad, cb)[0].status) == 5
    # switched off
    ctx.set_option("netdual", 0)
    assert int(run(ctx, A, b, c, u, vb, cb)[0].status) == 5
    ctx.set_option("netdual", -1)
    assert want["status"] == 0
    # primal infeasible: a path 0 -> 1 -> 2 that must carry 5 units through capacity 1
    tail3, head3 = np.array([0, 1]), np.array([1, 2])
    A3 = sp.csr_matrix((np.array([1.0, -1.0, 1.0, -1.0]), (np.array([0, 1, 1, 2]), np.array([0, 0, 1, 1]))), shape=(3, 2))
    b3, c3, u3 = np.array([5.0, 0.0, -5.0]), np.array([1.0, 1.0]), np.array([1.0, 9.0])
    vb3, cb3 = np.array([0, 0], dtype=np.int8), np.array([-1, -1, 0], dtype=np.int8)
    assert dual_network_simplex(tail3, head3, c3, u3, b3, vb3, root=2)["status"] == 1
    assert int(run(ctx, A3, b3, c3, u3, vb3, cb3)[0].status) == 1


def test_non_integral_data_reach_the_optimum(ctx):
    """Real-valued supplies and capacities: flows are no longer exact, so the run is compared by value and
    certificates, not pivot for pivot."""
    V, E = 500, 5000
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, 44, inf_frac=0.0)
    rng = np.random.default_rng(2)
    u = u * rng.uniform(0.5, 1.5, u.size)
    u[-V:] = np.inf
    bb = rng.uniform(-3, 3, V)
    bb[-1] -= bb.sum()
    out = bb >= 0
    # artificial arcs follow the sign of the supplies
    n = A.shape[1]
    tail[E:] = np.where(out, np.arange(V), V)
    head[E:] = np.where(out, V, np.arange(V))
    arcs = np.arange(n)
    A = sp.csr_matrix((np.concatenate([np.ones(n), -np.ones(n)]), (np.concatenate([tail, head]), np.concatenate([arcs, arcs]))),
                      shape=(V + 1, n))
    A.sort_indices()
    b = np.concatenate([bb, [0.0]])
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
    assert int(res.status) == 0
    certificates(A, b, c, u, tail, head, x, y, vbo, cbo)
    assert float(res.obj) == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)
    want = dual_network_simplex(tail, head, c, u, b, vb, root=V)
    assert abs(int(res.iters) - want["iters"]) <= max(5, want["iters"] // 20)
    assert float(res.obj) == pytest.approx(want["obj"], rel=1e-9)


def test_iteration_limit_and_tiny_networks(ctx):
    # the limit stops the run with status 3 and a consistent state: a spanning tree, flows that conserve
    A, b, c, u, tail, head, vb, cb = big_m_network(300, 3000, 2, inf_frac=0.0)
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb, max_iter=25)
    assert int(res.status) == 3 and int(res.iters) == 25
    assert np.count_nonzero(vbo == 0) == A.shape[0] - 1
    assert np.abs(A @ x - b).max() <= 1e-9 * (1 + np.abs(b).max())
    # ... and continuing from that basis reaches the optimum of the uninterrupted run
    full = run(ctx, A, b, c, u, vb, cb)
    res2, x2, y2, vb2, cb2 = run(ctx, A, b, c, u, vbo.astype(np.int8), cbo.astype(np.int8))
    assert int(res2.status) == 0 and float(res2.obj) == pytest.approx(float(full[0].obj), rel=1e-12)
    # two nodes, one arc + the root's artificial arcs
    for V, E, seed in ((2, 1, 0), (3, 2, 1), (5, 4, 3)):
        A, b, c, u, tail, head, vb, cb = big_m_network(V, E, seed, inf_frac=0.0)
        res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
        want = dual_network_simplex(tail, head, c, u, b, vb, root=V)
        same_pivots(res, x, vbo, want)
        assert float(res.obj) == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)


def test_uncapacitated_networks_go_to_the_primal_method_at_once(ctx):
    """No capacitated arc outside the tree (the OT crossovers): the dual method declines before any set-up work,
    unless it is forced -- and forced it still answers correctly (optimal start) or declines (an arc would have to
    flip)."""
    A, b, c, u, tail, head, vb, cb = big_m_network(60, 400, 1, inf_frac=1.0)
    assert int(run(ctx, A, b, c, u, vb, cb)[0].status) == 5
    ctx.set_option("netdual", 1)
    try:
        assert int(run(ctx, A, b, c, u, vb, cb)[0].status) == 5       # wrong-signed uncapacitated arcs
    finally:
        ctx.set_option("netdual", -1)


def test_kept_tree_with_changed_costs_recomputes_the_potentials(ctx):
    """Solve, change the costs, solve again from the basis handed back: the tree arrays kept in the context are the
    same tree, but the kept potentials belong to the OLD costs -- the set-up has to run again (the kept path is
    taken only when every tree arc has reduced cost zero under the costs of this call).  Doubling every cost
    (network_methods/net_manager.py:276-283 rescale_cost is such a change) keeps the optimal tree optimal: no
    iteration, the same flows, potentials doubled; then a change of the non-tree costs that asks for pivots."""
    V, E = 300, 3000
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, 2, inf_frac=0.0)
    res, x, y, vbo, cbo = run(ctx, A, b, c, u, vb, cb)
    assert int(res.status) == 0
    res2, x2, y2, vb2, cb2 = run(ctx, A, b, 2.0 * c, u, vbo.astype(np.int8), cbo.astype(np.int8))
    assert int(res2.status) == 0 and int(res2.iters) == 0
    assert int(res2.warm_start_used) == 1                           # kept potentials rejected, tree set up again
    assert np.array_equal(x2, x) and np.array_equal(y2, 2.0 * y)    # (integral data: exact)
    np.testing.assert_array_equal(vb2, vbo)
    certificates(A, b, 2.0 * c, u, tail, head, x2, y2, vb2, cb2)
    # unchanged costs afterwards: the kept path is taken again
    res3, x3, _, vb3, _ = run(ctx, A, b, 2.0 * c, u, vb2.astype(np.int8), cb2.astype(np.int8))
    assert int(res3.status) == 0 and int(res3.iters) == 0 and int(res3.warm_start_used) == 2
    assert np.array_equal(x3, x2)
    # cheaper non-tree arcs (capacitated, so they may flip): the run is the oracle's, pivot for pivot
    c3 = 2.0 * c
    nontree = np.flatnonzero(vb2[:E] != 0)
    c3[nontree[::3]] = 1.0
    res4, x4, y4, vb4, cb4 = run(ctx, A, b, c3, u, vb2.astype(np.int8), cb2.astype(np.int8))
    want = dual_network_simplex(tail, head, c3, u, b, vb2, root=V)
    assert int(res4.warm_start_used) == 2                           # tree arcs kept their costs: potentials still valid
    same_pivots(res4, x4, vb4, want)
    assert int(res4.iters) > 0
    assert float(res4.obj) == pytest.approx(highs(A, b, c3, u), rel=1e-9, abs=1e-9)
