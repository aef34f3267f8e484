"""GPU property tests (hypothesis): randomly shaped inputs for the kernels whose correctness depends on
segment structure -- ragged rows, empty rows and columns, duplicates, tiles cut in odd places, ties and
special values -- each example checked bit for bit against the oracle.  Few examples per property; the
shapes that matter are forced by the strategies (tiny, around the 256-segment and 4096-entry edges)."""
import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import bits_equal
from oracle import lp_path as L
from oracle import net_path as N

pytestmark = pytest.mark.gpu

SET = settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture,
                                                                       HealthCheck.too_slow], derandomize=True)


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def random_csr(rng, m, n, style):
    """CSR matrices the walk must handle: 'ragged' (geometric row lengths incl. empty rows, unsorted,
    duplicates), 'one_long' (a row longer than two staging chunks), 'dense_band'."""
    if style == "ragged":
        lens = np.minimum(rng.geometric(0.15, size=m) - 1, 4 * n)
        lens[rng.random(m) < 0.2] = 0
    elif style == "one_long":
        lens = rng.integers(0, 6, size=m)
        lens[rng.integers(0, m)] = 9000 + int(rng.integers(0, 50))
    else:
        lens = np.full(m, min(n, 33))
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(indptr[-1])
    cols = rng.integers(0, n, size=nnz).astype(np.int32)              # unsorted, duplicates allowed
    vals = rng.standard_normal(nnz)
    vals[rng.random(nnz) < 0.02] = 0.0                                # explicit zeros stay entries
    return sp.csr_matrix((vals, cols, indptr), shape=(m, n))


@SET
@given(seed=st.integers(0, 2 ** 31 - 1), m=st.sampled_from([1, 2, 255, 256, 257, 700]),
       n=st.sampled_from([1, 3, 256, 257, 1500]), style=st.sampled_from(["ragged", "one_long", "dense_band"]))
def test_scoring_and_pricing_on_random_structures(ctx, seed, m, n, style):
    rng = np.random.default_rng(seed)
    A = random_csr(rng, m, n, style)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    l = np.where(rng.random(n) < 0.3, -np.inf, -rng.random(n))
    u = np.where(rng.random(n) < 0.5, np.inf, 1.0 + rng.random(n))
    want = L.scoring_pass(A, b, c, l, u, x, y, 0.3, 0.3)
    dA = ctx.matrix(A)
    d = {k: ctx.to_device(v) for k, v in dict(b=b, c=c, l=l, u=u, x=x, y=y).items()}
    s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], 0.3, s_d, code)
    ctx.score_rows(dA, d["x"], d["b"], d["y"], 0.3, s_p, flag)
    assert bits_equal(s_d.download(), want["s_d"]) and bits_equal(s_p.download(), want["s_p"])
    assert np.array_equal(code.download(), want["code"]) and np.array_equal(flag.download(), want["rowflag"])
    assert np.array_equal(ctx.where(code, 1), want["fix_low"]) and np.array_equal(ctx.where(code, 2), want["fix_up"])
    assert np.array_equal(ctx.where(flag, 0xFF), want["fixed_rows"])
    vb = rng.integers(-2, 1, n).astype(np.int8)
    rc = ctx.empty(n, np.float64)
    res = ctx.price(dA, d["y"], d["c"], ctx.to_device(vb), 1e-6, rc)
    want_rc = N.mcf_reduced_cost(A, c, y, vb.astype(int))
    assert bits_equal(rc.download(), want_rc)
    mn, am, bad = ctx.read_price(res)
    assert mn == want_rc.min() and am == int(np.flatnonzero(want_rc == want_rc.min())[0])
    assert bad == int(np.count_nonzero(~(want_rc >= -1e-6)))
    dA.free()


@SET
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([1, 2, 63, 64, 65, 2047, 2048, 2049, 10_000]),
       palette=st.sampled_from(["few", "many", "special"]))
def test_ranking_rule_on_random_keys(ctx, seed, n, palette):
    rng = np.random.default_rng(seed)
    if palette == "few":
        key = rng.choice(np.array([0.0, 0.25, 0.5, 1.0]), size=n)              # long runs of ties
    elif palette == "many":
        key = rng.random(n) * 10.0 ** rng.integers(-300, 300, size=n)
    else:
        key = rng.choice(np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 5e-324, -5e-324]), size=n)
    got = ctx.argsort_desc(ctx.to_device(key)).download()
    assert np.array_equal(got, N.rank_desc(key))


@SET
@given(seed=st.integers(0, 2 ** 31 - 1), S=st.integers(1, 40), D=st.integers(1, 40), zeros=st.floats(0.0, 0.9),
       levels=st.sampled_from([0, 3]))
def test_spanning_forest_weight_and_size(ctx, seed, S, D, zeros, levels):
    """Any maximum-weight spanning forest has the same total weight and the same number of arcs as
    scipy's (ties may be broken differently), and it never contains a zero-weight arc."""
    from scipy.sparse import csgraph
    rng = np.random.default_rng(seed)
    w = rng.random(S * D) + 0.01
    if levels:
        w = np.round(w * levels) / levels + 0.5                              # many equal weights
    w[rng.random(S * D) < zeros] = 0.0
    got = ctx.spanning_tree_ot(S, D, ctx.to_device(w))
    W = w.reshape(S, D)
    graph = sp.bmat([[None, sp.csr_matrix(-W)], [sp.csr_matrix((D, S)), None]], format="csr")
    ref = csgraph.minimum_spanning_tree(graph)
    assert got.size == ref.nnz
    assert np.all(w[got] > 0)
    assert w[got].sum() == pytest.approx(-ref.sum(), rel=1e-12, abs=1e-15)
    # acyclic and spanning the same components: union-find over the chosen arcs never closes a cycle
    parent = list(range(S + D))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for e in got:
        a, b2 = find(int(e) // D), find(S + int(e) % D)
        assert a != b2
        parent[a] = b2


@SET
@given(seed=st.integers(0, 2 ** 31 - 1), m=st.sampled_from([1, 40, 300]), n=st.sampled_from([2, 257, 900]),
       frac=st.floats(0.0, 1.0))
def test_sub_problem_compaction_on_random_codes(ctx, seed, m, n, frac):
    rng = np.random.default_rng(seed)
    A = random_csr(rng, m, n, "ragged")
    code = np.where(rng.random(n) < frac, rng.integers(1, 3, n), 0).astype(np.uint8)
    sub, non_fix = ctx.compact_columns(ctx.matrix(A), ctx.to_device(code))
    keep = np.flatnonzero(code == 0)
    assert np.array_equal(non_fix.download(keep.size), keep)
    # expected: every kept entry in its stored order, columns renumbered (duplicates and explicit zeros stay)
    kept = code[A.indices] == 0
    colmap = np.cumsum(code == 0) - 1
    row_of = np.repeat(np.arange(m), np.diff(A.indptr))
    want_indptr = np.concatenate([[0], np.cumsum(np.bincount(row_of[kept], minlength=m))])
    got = sub.to_scipy()
    assert got.shape == (m, keep.size)
    assert np.array_equal(got.indptr, want_indptr)
    assert np.array_equal(got.indices, colmap[A.indices[kept]])
    assert bits_equal(got.data, A.data[kept])


@settings(max_examples=120, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 2 ** 31 - 1), m=st.integers(1, 24), n=st.integers(1, 40),
       kind=st.sampled_from(["feasible", "degenerate", "free", "maybe_infeasible", "maybe_unbounded"]),
       defer=st.sampled_from([0, 1]), pricing=st.sampled_from([0, 1]))
def test_device_simplex_agrees_with_highs_on_random_small_lps(seed, m, n, kind, defer, pricing):
    """K16 against scipy's HiGHS on small LPs of every flavour, under every combination of its two options
    (inverse updated per pivot / per batch, Dantzig / Devex): same status, same optimal value, and a
    returned basis that certifies the vertex (primal and dual feasible, m basic variables)."""
    from scipy.optimize import linprog
    from smart_crossover.formats import GeneralLP
    from smart_crossover.hip import default_context
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    dev = default_context()
    dev.set_option("spx_defer", defer)
    dev.set_option("spx_pricing", pricing)
    try:
        _simplex_case(seed, m, n, kind, linprog, GeneralLP, SolverSettings, solve_lp)
    finally:
        dev.set_option("spx_defer", -1)
        dev.set_option("spx_pricing", 1)


def _simplex_case(seed, m, n, kind, linprog, GeneralLP, SolverSettings, solve_lp):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=min(1.0, 3.0 / max(n, 1) + 0.15), random_state=seed % (2 ** 31), format="csr")
    A.data = np.round(rng.uniform(-3, 3, A.nnz) * 2) / 2                       # halves: plenty of ties
    A.eliminate_zeros()
    sense = np.where(rng.random(m) < 0.5, "<", "=")
    l = np.zeros(n)
    u = np.where(rng.random(n) < 0.4, np.round(rng.uniform(1, 5, n)), np.inf)
    x0 = np.where(np.isinf(u), rng.integers(0, 4, n), np.minimum(rng.integers(0, 4, n), u)).astype(float)
    c = np.round(rng.uniform(-2, 4, n) * 2) / 2
    if kind == "free":
        free = rng.random(n) < 0.3
        l[free], u[free] = -np.inf, np.inf
        c[free] = 0.0
    slack = np.where(sense == "<", rng.integers(0, 3, m), 0).astype(float)
    if kind == "degenerate":
        slack[:] = 0.0
        x0[rng.random(n) < 0.6] = 0.0
    b = A @ x0 + slack
    if kind == "maybe_infeasible":
        b = b - np.where(sense == "=", rng.integers(0, 2, m) * 7.0, rng.integers(0, 2, m) * 50.0)
    if kind == "maybe_unbounded":
        c = c - 3.0
    lp = GeneralLP(A, b, c, l, u, sense)
    lt = sense == "<"
    ref = linprog(c, A_ub=A[lt] if lt.any() else None, b_ub=b[lt] if lt.any() else None,
                  A_eq=A[~lt] if (~lt).any() else None, b_eq=b[~lt] if (~lt).any() else None,
                  bounds=[(None if np.isinf(a) else a, None if np.isinf(z) else z) for a, z in zip(l, u)], method="highs",
                  options={"presolve": False})      # with presolve HiGHS may answer "infeasible" for "infeasible or unbounded"
    out = solve_lp(lp, "HIP", "default", SolverSettings(log_console=0))
    if ref.status == 0:
        assert out.status == "OPTIMAL", (kind, m, n, seed)
        assert out.obj_val == pytest.approx(ref.fun, rel=1e-7, abs=1e-7)
        x, y, vb, cb = out.x, out.y, out.basis.vbasis, out.basis.cbasis
        s_p = b - A @ x
        assert np.all(np.abs(s_p[~lt]) <= 1e-6) and np.all(s_p[lt] >= -1e-6)
        assert np.all(x >= l - 1e-6) and np.all(x <= u + 1e-6)
        rc = c - A.T @ y
        assert np.all(rc[vb == -1] >= -1e-6) and np.all(rc[vb == -2] <= 1e-6) and np.all(np.abs(rc[vb == -3]) <= 1e-6)
        assert int(np.count_nonzero(vb == 0) + np.count_nonzero(cb == 0)) == m
    elif ref.status == 2:
        assert out.status == "INFEASIBLE", (kind, m, n, seed)
    elif ref.status == 3:
        assert out.status == "UNBOUNDED", (kind, m, n, seed)


@settings(max_examples=30, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), S=st.integers(2, 14), D=st.integers(2, 14),
       method=st.sampled_from(["tnet", "cnet_ot"]), solver=st.sampled_from(["HIP", "HGS"]))
def test_network_crossover_reaches_the_ot_optimum(seed, S, D, method, solver):
    """Whole network crossover on random small transport problems, re-solves on the device (with the
    session that keeps the basis inverse across rounds) or in HiGHS: optimal cost, feasible plan."""
    import io
    from contextlib import redirect_stdout
    from scipy.optimize import linprog
    import workloads
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.ot(S, D, seed=seed)
    ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
    mcf = ot.to_MCF()
    ref = linprog(mcf.c, A_eq=mcf.A, b_eq=mcf.b, bounds=(0, None), method="highs")
    assert ref.status == 0
    with redirect_stdout(io.StringIO()):
        out = network_crossover(inst.x, ot=ot, method=method, solver=solver, solver_settings=SolverSettings(log_console=0))
    X = (out.x.reshape(S + 1, D + 1)[:S, :D] if method == "cnet_ot" else out.x.reshape(S, D))
    assert np.allclose(X.sum(axis=1), inst.s, atol=1e-7) and np.allclose(X.sum(axis=0), inst.d, atol=1e-7)
    assert np.all(X >= -1e-9)
    assert float((X * inst.M).sum()) == pytest.approx(ref.fun, rel=1e-8, abs=1e-10)
    assert np.count_nonzero(X > 1e-12) <= S + D - 1                               # a basic solution
