"""GPU: BASELINE.json's full-size configurations, checked through size-independent properties and
sampled bit-exact comparisons (the oracle finishes these sizes in seconds, so most checks are exact).

  c4  MCF 2^17 nodes x 2^20 arcs      indicators, ranking, pricing
  c5  LP  1e6 rows x 1e7 columns      scoring pass: sampled exact sums, cross-kernel identities,
                                      determinism, index-set invariants
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal
from oracle import lp_path as L
from oracle import net_path as N
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def test_config4_mcf_indicators_ranking_pricing(ctx):
    inst = workloads.config4()
    V, E = inst.A.shape
    assert (V, E) == (2 ** 17, 2 ** 20)
    dA = ctx.matrix(inst.A)
    ind = ctx.empty(E, np.float64)
    ctx.flow_indicator_mcf(dA, ctx.to_device(inst.x), ctx.to_device(inst.u), ind)
    got = ind.download()
    want, _ = N.mcf_flow_indicators(inst.A, inst.x, inst.u)
    assert bits_equal(got, want)
    q = ctx.argsort_desc(ind).download()
    # sortedness + permutation + the tie rule, without a second full sort on the host
    key = got[q]
    assert np.all(key[:-1] >= key[1:])
    assert np.array_equal(np.sort(q), np.arange(E))
    ties = key[:-1] == key[1:]
    assert np.all(q[:-1][ties] > q[1:][ties])
    assert np.array_equal(q, N.rank_desc(got))
    # pricing on the incidence matrix with random potentials
    rng = np.random.default_rng(1)
    y = rng.standard_normal(V)
    vb = rng.integers(-2, 1, E).astype(np.int8)
    rc = ctx.empty(E, np.float64)
    res = ctx.price(dA, ctx.to_device(y), ctx.to_device(inst.c), ctx.to_device(vb), 1e-6, rc)
    want_rc = N.mcf_reduced_cost(inst.A, inst.c, y, vb.astype(int))
    assert bits_equal(rc.download(), want_rc)
    mn, am, bad = ctx.read_price(res)
    assert mn == want_rc.min() and am == int(np.flatnonzero(want_rc == want_rc.min())[0])
    assert bad == int(np.count_nonzero(~(want_rc >= -1e-6)))
    dA.free()


def test_config5_scoring_pass_properties(ctx):
    sh = workloads.lp_shard(0, 1)                       # 1e6 x 1e7, 8e7 nnz, netlib-style structure
    m, n = sh.m, sh.n_block
    dC, dR = ctx.column_shard(sh.col_block), ctx.row_shard(sh.row_block)
    d = {k: ctx.to_device(getattr(sh, k)) for k in ("y", "x", "c", "l", "u", "b")}
    s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    gamma = 1e-3

    def run():
        ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], gamma, s_d, code)
        ctx.score_rows(dR, d["x"], d["b"], d["y"], gamma, s_p, flag)
        return s_d.download(), code.download(), s_p.download(), flag.download()

    sd1, cd1, sp1, fl1 = run()
    sd2, cd2, sp2, fl2 = run()
    # determinism: no atomics, fixed orders -> bitwise identical on every launch
    assert sd1.tobytes() == sd2.tobytes() and sp1.tobytes() == sp2.tobytes()
    assert np.array_equal(cd1, cd2) and np.array_equal(fl1, fl2)

    # sampled exactness: explicit left-to-right sums for 5000 random columns and 2000 random rows
    rng = np.random.default_rng(7)
    cols = np.sort(rng.choice(n, 5000, replace=False))
    C = sh.col_block
    for j in cols[:500]:
        lo, hi = C.indptr[j], C.indptr[j + 1]
        acc = np.float64(0.0)
        for k in range(lo, hi):
            acc = acc + np.float64(C.data[k]) * np.float64(sh.y[C.indices[k]])
        assert sd1[j] == sh.c[j] - acc
    sub = C[:, cols]
    assert bits_equal(sd1[cols], sh.c[cols] - sub.T @ sh.y)              # scipy on the sampled columns
    rows = np.sort(rng.choice(m, 2000, replace=False))
    assert bits_equal(sp1[rows], sh.b[rows] - sh.row_block[rows] @ sh.x)

    # the flags are pure functions of the slacks
    assert np.array_equal(cd1, L.column_codes(sh.x[:n], sh.l, sh.u, sd1, gamma))
    assert np.array_equal(fl1, L.row_flags(sp1, sh.y, gamma))

    # cross-kernel identity: pricing without basis flips reproduces s_d bit for bit
    rc = ctx.empty(n, np.float64)
    res = ctx.price(dC, d["y"], d["c"], None, 1e-6, rc)
    assert rc.download().tobytes() == sd1.tobytes()
    mn, am, bad = ctx.read_price(res)
    assert mn == sd1.min() and am == int(np.argmin(sd1)) and bad == int(np.count_nonzero(~(sd1 >= -1e-6)))

    # index sets: ascending, complete, consistent with the codes
    low, up, fr = ctx.where(code, 1), ctx.where(code, 2), ctx.where(flag)
    for idx, mask in ((low, (cd1 & 1) != 0), (up, (cd1 & 2) != 0), (fr, fl1 != 0)):
        assert idx.dtype == np.int64 and np.all(np.diff(idx) > 0) and idx.size == int(mask.sum())
        assert mask[idx].all()
    assert low.size > n // 2 and up.size > 0 and fr.size > 0

    # a checksum of checksums against an independent float128-free bound: sum(s_d) = sum(c) - y.(A 1)
    lhs = float(np.sum(sd1))
    rhs = float(np.sum(sh.c) - sh.y @ (C @ np.ones(n)))
    assert abs(lhs - rhs) <= 1e-9 * (abs(rhs) + np.sum(np.abs(sd1)))
    dC.free()
    dR.free()


@pytest.mark.parametrize("method", ["tnet", "cnet_ot"])
def test_config3_network_crossover_device_resident_equals_host_solver(method):
    """Config 3 (OT 784 x 784, 614,656 arcs) through the whole network crossover, once with the re-solves
    on the device and once in HiGHS: same optimal cost, a feasible basic plan, and the device path reuses
    the kept basis inverse in every round after the first."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.config3()
    S, D = inst.M.shape
    costs = {}
    for solver in ("HIP", "HGS"):
        ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
        with redirect_stdout(io.StringIO()) as text:
            out = network_crossover(inst.x, ot=ot, method=method, solver=solver, solver_settings=SolverSettings(log_console=0))
        assert "Column generation fails" not in text.getvalue()
        X = (out.x.reshape(S + 1, D + 1)[:S, :D] if method == "cnet_ot" else out.x.reshape(S, D))
        assert np.abs(X.sum(axis=1) - inst.s).max() < 1e-9 and np.abs(X.sum(axis=0) - inst.d).max() < 1e-9
        assert X.min() >= -1e-12 and np.count_nonzero(X > 1e-13) <= S + D - 1
        costs[solver] = float((X * inst.M).sum())
    assert costs["HIP"] == pytest.approx(costs["HGS"], rel=1e-9)


def test_config4_network_crossover_on_the_device():
    """BASELINE config 4 (min-cost flow, 131,072 nodes, 1,048,576 arcs) through the whole CNET_MCF crossover with
    every re-solve on the device (dual network simplex K16d; reference network_methods/algorithms.py:14-77 with
    net_manager.py:211-222).  HiGHS needs 96 s for its leg, so it is not re-run here: its optimum is the recorded
    one (tests/golden/c4_cnet_mcf.json: two independent HiGHS-backed runs, profiles/r01 and profiles/r02) and the
    rest are solver-independent certificates -- conservation, bounds, a spanning-tree basis, dual feasibility."""
    import io
    import json
    import os
    from contextlib import redirect_stdout
    import scipy.sparse as sp
    from conftest import ROOT
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller import hip as hipmod
    inst = workloads.config4()
    V, E = inst.A.shape
    mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
    used = []
    orig = hipmod.HipCaller._solve

    def spy(self):
        orig(self)
        used.append((self.solved_by, int(self._res.iters)))

    hipmod.HipCaller._solve = spy
    try:
        with redirect_stdout(io.StringIO()) as text:
            out = network_crossover(inst.x.copy(), mcf=mcf, method="cnet_mcf", solver="HIP")
    finally:
        hipmod.HipCaller._solve = orig
    assert "Column generation fails" not in text.getvalue()
    assert used and all(how == "netdual" for how, _ in used)
    x = out.x[:E]
    assert np.abs(inst.A @ x - inst.b).max() <= 1e-7 * (1 + np.abs(inst.b).max())
    assert x.min() >= -1e-8 and (x - inst.u).max() <= 1e-8          # tree arcs: within the solver's 1e-9 of a bound
    assert np.abs(out.x[E:E + V]).max() < 1e-8                       # no flow left on the artificial arcs
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_cnet_mcf.json")))
    assert float(inst.c @ x) == pytest.approx(want["optimal_cost_highs"], rel=1e-9)
    assert sum(it for _, it in used) == out.iter_count
    # the basis handed back: a spanning tree of the extended network (V + 1 nodes) + the root row
    vb, cb = out.basis.vbasis, out.basis.cbasis
    assert np.count_nonzero(vb == 0) == V and np.count_nonzero(cb == 0) == 1
    assert np.all(x[vb[:E] == -1] == 0.0) and np.all(x[vb[:E] == -2] == inst.u[vb[:E] == -2])
    basic = np.flatnonzero(vb[:E] == 0)
    g = sp.coo_matrix((np.ones(basic.size), (inst.tail[basic], inst.head[basic])), shape=(V, V))
    ncomp, _ = sp.csgraph.connected_components(g, directed=False)
    assert ncomp == V - basic.size                                  # the basic original arcs are a forest


def test_config2_perturbation_crossover_end_to_end_on_the_device():
    """BASELINE config 2 (2e4 x 1e5, 2e6 entries) through the crossover proper, all on the GPU: from the
    interior point (x, y) to a vertex of the perturbed sub-problem with its basis -- get_perturb_problem, then
    the re-solve of the 2e4-row sub-LP by the device simplex (crash basis from the interior point, dense
    inverse of 3.2 GB), then the gap test (reference lp_methods/algorithms.py:45-67).  HiGHS-independent
    certificates: primal and dual feasibility by basis status at the reference's tolerances, |B| = m, and the
    gap against the interior objective."""
    import io
    from contextlib import redirect_stdout
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.config2()
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    with redirect_stdout(io.StringIO()):
        mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        sub = mgr.lp_sub
        out = solve_lp(sub, "HIP", "barrier", SolverSettings(presolve="on", log_console=0),
                       warm_start_solution=(mgr.get_subx(inst.x), inst.y))
    assert out.status == "OPTIMAL"
    m, n = sub.A.shape
    x, y, vb, cb = out.x, out.y, out.basis.vbasis, out.basis.cbasis
    tol = 1e-6
    s_p = sub.b - sub.A @ x
    lt = np.asarray(sub.sense) == "<"
    assert np.all(np.abs(s_p[~lt]) <= tol) and np.all(s_p[lt] >= -tol)
    assert np.all(x >= sub.l - tol) and np.all(x <= sub.u + tol)
    rc = sub.c - sub.A.T @ y
    assert np.all(rc[vb == -1] >= -tol) and np.all(rc[vb == -2] <= tol) and np.all(np.abs(rc[vb == 0]) <= tol)
    assert np.all(y[lt & (cb == -1)] <= tol) and np.all(np.abs(y[cb == 0]) <= tol)
    assert int(np.count_nonzero(vb == 0) + np.count_nonzero(cb == 0)) == m
    # the vertex is (numerically) optimal for the original objective too: the reference's own acceptance test
    x_full = mgr.get_orix(x)
    obj_interior = float(lp.c @ inst.x)
    gap = abs(float(lp.c @ x_full) - obj_interior) / (abs(float(lp.c @ x_full)) + abs(obj_interior) + 1)
    assert gap < 1e-8                                                # PRIMAL_DUAL_GAP_THRESHOLD (parameters.py:26)


def test_config5_projector_cg_history_against_the_oracle(ctx):
    """K4 at full config-5 size (1e6 x 1e7, 8e7 entries): the first 20 CG iterations of the projector against the
    CPU statement of the same algorithm (oracle.lp_path.cg_legacy on the matrix-free operator) -- ||proj|| and the
    relative residual after 5, 10 and 20 iterations, 1e-9 relative (the two sides sum in different orders)."""
    sh = workloads.lp_shard(0, 1)
    m, n = sh.m, sh.n_block
    rng = np.random.default_rng(11)
    sense = np.where(rng.random(m) < 0.5, "<", "=")
    A = sh.row_block
    xr = L.x_perturb_val(sh.x, sh.l, sh.u)
    xx = L.standard_x(A, sh.b, sense, xr)
    rows = L.slack_rows(sense)
    xa, xs_c = xx[:n], xx[n:]
    xs = np.zeros(m)
    xs[rows] = xs_c
    C = sh.col_block                                   # CSC of A = CSR of A^T, same arrays
    AT = sp.csr_matrix((C.data, C.indices, C.indptr), shape=(n, m))
    c_std = L.standard_c(sh.c, sense)
    v = xx * c_std

    def Yt(p):
        return np.concatenate([xa * (AT @ p), xs_c * p[rows]])

    def Ymul(w):
        out = A @ (xa * w[:n])
        out[rows] += xs_c * w[n:]
        return out

    Yv = Ymul(v)
    bn = float(np.linalg.norm(Yv))
    dA = ctx.matrix(A)
    d_xa, d_xs, d_c = ctx.to_device(xa), ctx.to_device(xs), ctx.to_device(sh.c)
    for iters in (5, 10, 20):
        res = ctx.projector_norm(dA, d_xa, d_xs, d_c, 1e-8, iters)
        z, it, conv = L.cg_legacy(lambda p: Ymul(Yt(p)), Yv, 1e-8, iters)
        assert it == iters == int(res.iters) and not conv and int(res.converged) == 0
        proj = v - Yt(z)
        rel = float(np.linalg.norm(Yv - Ymul(Yt(z)))) / bn
        assert res.proj_norm == pytest.approx(float(np.linalg.norm(proj)), rel=1e-9)
        assert res.rel_residual == pytest.approx(rel, rel=1e-6)      # (a difference of nearly equal vectors)
    dA.free()
