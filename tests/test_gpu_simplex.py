"""GPU: K16, the device simplex behind solver="HIP".  Solver-chosen bases are not comparable across
solvers, so the checks are solver-independent certificates: objective equal to an independent HiGHS
solve, primal feasibility, dual feasibility by basis status, complementary basis size."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

import workloads

pytestmark = pytest.mark.gpu


def settings():
    from smart_crossover.solver_caller.caller import SolverSettings
    return SolverSettings(log_console=0)


def check_vertex(lp, out, want_obj, tol=1e-6):
    x, y, basis = out.x, out.y, out.basis
    assert out.status == "OPTIMAL"
    assert out.obj_val == pytest.approx(want_obj, rel=1e-7, abs=1e-7)
    assert float(lp.c @ x) == pytest.approx(want_obj, rel=1e-7, abs=1e-7)
    s_p = lp.b - lp.A @ x
    lt = np.asarray(lp.sense) == "<"
    assert np.all(np.abs(s_p[~lt]) <= tol) and np.all(s_p[lt] >= -tol)
    assert np.all(x >= lp.l - tol) and np.all(x <= lp.u + tol)
    rc = lp.c - lp.A.T @ y
    vb, cb = basis.vbasis, basis.cbasis
    assert np.all(rc[vb == -1] >= -tol) and np.all(rc[vb == -2] <= tol) and np.all(np.abs(rc[vb == -3]) <= tol)
    assert np.all(np.abs(rc[vb == 0]) <= tol)
    assert np.all(y[lt & (cb == -1)] <= tol)                         # a tight '<' row has a non-positive dual
    assert np.all(np.abs(y[cb == 0]) <= tol)                         # basic slack: zero dual
    assert int(np.count_nonzero(vb == 0) + np.count_nonzero(cb == 0)) == lp.b.size


def general_lp(inst):
    from smart_crossover.formats import GeneralLP
    return GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)


@pytest.mark.parametrize("m,n,k,seed", [(27, 51, 2, 2024), (60, 150, 4, 1), (200, 700, 5, 2), (400, 1600, 5, 77)])
def test_cold_start_matches_highs(m, n, k, seed):
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.config1() if (m, n) == (27, 51) else workloads.sparse_lp(m, n, k, seed=seed, stratified=False, frac_upper=0.4)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    assert ref.status == "OPTIMAL"
    out = solve_lp(lp, "HIP", "primal_simplex", settings())
    check_vertex(lp, out, ref.obj_val)
    assert out.iter_count > 0


@pytest.fixture
def inverse_update_mode():
    """Sets the 'spx_defer' option of the shared context for one test and restores the default."""
    from smart_crossover.hip import default_context
    ctx = default_context()

    def choose(mode, pricing=1):
        ctx.set_option("spx_defer", mode)
        ctx.set_option("spx_pricing", pricing)
    yield choose
    ctx.set_option("spx_defer", -1)
    ctx.set_option("spx_pricing", 1)


@pytest.mark.parametrize("mode,pricing", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k,seed", [(27, 51, 2, 2024), (45, 130, 3, 12), (200, 700, 5, 2), (700, 2400, 5, 9)])
def test_inverse_update_modes_agree(inverse_update_mode, mode, pricing, m, n, k, seed):
    """Rank-one update per pivot (0) and product form folded in every 32 pivots (1), Dantzig (0) and Devex
    (1) pricing: cold start, a warm start that crashes in more than one batch of basic columns, phase 1 on
    '=' rows -- all must end at the optimum HiGHS finds, and a warm start from the final basis must need
    no pivot in any of them."""
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.config1() if (m, n) == (27, 51) else workloads.sparse_lp(m, n, k, seed=seed, stratified=False, frac_upper=0.4)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    assert ref.status == "OPTIMAL"
    inverse_update_mode(mode, pricing)
    out = solve_lp(lp, "HIP", "primal_simplex", settings())
    check_vertex(lp, out, ref.obj_val)
    again = solve_lp(lp, "HIP", "default", settings(), warm_start_basis=out.basis)
    assert again.iter_count == 0
    check_vertex(lp, again, ref.obj_val)
    warm = solve_lp(lp, "HIP", "default", settings(), warm_start_basis=ref.basis)
    check_vertex(lp, warm, ref.obj_val)


def test_tiny_and_degenerate_shapes():
    """One row; an empty column (bounded, so it just sits at the cheaper bound); a row nobody touches."""
    from smart_crossover.formats import GeneralLP
    from smart_crossover.solver_caller.solving import solve_lp
    one = GeneralLP(sp.csr_matrix(np.array([[1.0, 2.0, 0.0]])), np.array([4.0]), np.array([-1.0, -1.0, 3.0]),
                    np.zeros(3), np.array([np.inf, np.inf, 5.0]), np.array(["<"]))
    out = solve_lp(one, "HIP", "default", settings())
    assert out.status == "OPTIMAL" and out.obj_val == pytest.approx(-4.0)
    assert out.x == pytest.approx([4.0, 0.0, 0.0])
    neg = GeneralLP(one.A, one.b, np.array([-1.0, -1.0, -3.0]), one.l, one.u, one.sense)     # empty column wants its upper bound
    out = solve_lp(neg, "HIP", "default", settings())
    assert out.status == "OPTIMAL" and out.obj_val == pytest.approx(-19.0) and out.x[2] == pytest.approx(5.0)
    A = sp.csr_matrix(np.array([[1.0, 1.0], [0.0, 0.0], [1.0, -1.0]]))                      # middle row is empty
    idle = GeneralLP(A, np.array([2.0, 0.0, 0.0]), np.array([1.0, 2.0]), np.zeros(2), np.full(2, np.inf),
                     np.array(["=", "=", "="]))
    out = solve_lp(idle, "HIP", "default", settings())
    assert out.status == "OPTIMAL" and out.x == pytest.approx([1.0, 1.0]) and out.obj_val == pytest.approx(3.0)
    bad = GeneralLP(A, np.array([2.0, 1.0, 0.0]), idle.c, idle.l, idle.u, idle.sense)       # 0 = 1 in the empty row
    assert solve_lp(bad, "HIP", "default", settings()).status == "INFEASIBLE"


def test_warm_start_from_optimal_basis_needs_no_pivot():
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(120, 400, 4, seed=5, stratified=False, frac_upper=0.4)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    out = solve_lp(lp, "HIP", "primal_simplex", settings(), warm_start_basis=ref.basis)
    check_vertex(lp, out, ref.obj_val)
    assert out.iter_count == 0
    # a warm start from the HIP solver's own basis is a fixed point too
    again = solve_lp(lp, "HIP", "default", settings(), warm_start_basis=out.basis)
    assert again.iter_count == 0 and np.array_equal(again.basis.vbasis, out.basis.vbasis)
    # a garbage basis is dropped, not trusted
    from smart_crossover.output import Basis
    junk = Basis(np.zeros(lp.c.size), -np.ones(lp.b.size))
    out2 = solve_lp(lp, "HIP", "default", settings(), warm_start_basis=junk)
    check_vertex(lp, out2, ref.obj_val)


def test_free_variables_upper_bounds_and_inequalities():
    from smart_crossover.formats import GeneralLP
    from smart_crossover.solver_caller.solving import solve_lp
    rng = np.random.default_rng(3)
    m, n = 40, 90
    A = sp.random(m, n, density=0.15, random_state=4, format="csr")
    A.data = rng.uniform(-1, 1, A.nnz)
    x0 = rng.uniform(-1, 1, n)
    l = np.where(rng.random(n) < 0.3, -np.inf, x0 - rng.random(n))
    u = np.where(rng.random(n) < 0.3, np.inf, x0 + rng.random(n))
    free = rng.random(n) < 0.1
    l[free], u[free] = -np.inf, np.inf
    sense = np.where(rng.random(m) < 0.5, "<", "=")
    b = A @ x0 + np.where(sense == "<", rng.random(m), 0.0)
    y0 = rng.standard_normal(m)
    y0[sense == "<"] = -np.abs(y0[sense == "<"])
    c = A.T @ y0 + rng.standard_normal(n) * 0.1 * (~free)          # bounded below: dual feasible up to a small slack
    lp = GeneralLP(A, b, c, l, u, sense)
    ref = solve_lp(lp, "HGS", "default", settings())
    if ref.status != "OPTIMAL":
        pytest.skip("generated LP is not bounded")
    out = solve_lp(lp, "HIP", "default", settings())
    check_vertex(lp, out, ref.obj_val)


def test_infeasible_and_unbounded():
    from smart_crossover.formats import GeneralLP
    from smart_crossover.solver_caller.solving import solve_lp
    bad = GeneralLP(sp.csr_matrix(np.array([[1.0, 1.0], [1.0, 1.0]])), np.array([1.0, 3.0]), np.ones(2), np.zeros(2),
                    np.full(2, np.inf), np.array(["=", "="]))
    out = solve_lp(bad, "HIP", "default", settings())
    assert out.status == "INFEASIBLE" and out.x is None
    unb = GeneralLP(sp.csr_matrix(np.array([[1.0, -1.0]])), np.array([1.0]), np.array([-1.0, -1.0]), np.zeros(2),
                    np.full(2, np.inf), np.array(["<"]))
    out = solve_lp(unb, "HIP", "default", settings())
    assert out.status == "UNBOUNDED"
    from smart_crossover.solver_caller.caller import SolverSettings
    with pytest.raises(NotImplementedError):      # no interior-point method: barrier without crossover has no answer
        solve_lp(bad, "HIP", "barrier", SolverSettings(log_console=0, crossover="off"))


def test_mcf_and_degenerate_assignment():
    from smart_crossover.formats import MinCostFlow, OptTransport
    from smart_crossover.solver_caller.solving import solve_mcf, solve_ot
    mi = workloads.mcf(60, 400, seed=9)
    mcf = MinCostFlow(A=mi.A, b=mi.b, c=mi.c, u=mi.u)
    ref = solve_mcf(mcf, "HGS", "default", settings())
    out = solve_mcf(mcf, "HIP", "network_simplex", settings())
    assert out.status == "OPTIMAL" and out.obj_val == pytest.approx(ref.obj_val, rel=1e-8)
    assert np.allclose(mi.A @ out.x, mi.b, atol=1e-6) and np.all(out.x >= -1e-7) and np.all(out.x <= mi.u + 1e-7)
    # assignment problem: every vertex is highly degenerate (tests the Bland fallback path)
    k = 12
    rng = np.random.default_rng(2)
    ot = OptTransport(np.full(k, 1.0 / k), np.full(k, 1.0 / k), rng.integers(1, 20, (k, k)).astype(float))
    ref = solve_ot(ot, "HGS", "default", settings())
    out = solve_ot(ot, "HIP", "default", settings())
    assert out.status == "OPTIMAL" and out.obj_val == pytest.approx(ref.obj_val, rel=1e-8)
    X = out.x.reshape(k, k)
    assert np.allclose(X.sum(0), 1.0 / k, atol=1e-8) and np.allclose(X.sum(1), 1.0 / k, atol=1e-8)


def test_network_crossover_on_the_device_solver():
    """cnet_mcf with every re-solve on the GPU: ranking, sub-problems, pricing and simplex all device."""
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.solving import solve_mcf
    inst = workloads.mcf(80, 600, seed=31)
    want = solve_mcf(MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy()), "HGS", "default",
                     settings()).obj_val
    with redirect_stdout(io.StringIO()):
        out = network_crossover(inst.x, mcf=MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy()),
                                method="cnet_mcf", solver="HIP", solver_settings=settings())
    assert out.obj_val == pytest.approx(want, rel=1e-8)
    E = inst.A.shape[1]
    assert np.allclose(inst.A @ out.x[:E], inst.b, atol=1e-6) and np.all(out.x[E:] < 1e-8)


def test_perturbation_crossover_with_split_backend():
    """Barrier on HiGHS, every simplex-type re-solve on the device ('HGS+HIP')."""
    from smart_crossover.lp_methods.algorithms import run_perturb_algorithm
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(150, 500, 4, seed=8, stratified=False, frac_upper=0.3)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = run_perturb_algorithm(lp, "HGS+HIP", 1e-10, 1e-6)
    assert out.status == "OPTIMAL"
    if "A primal optimal BFS is found" not in buf.getvalue():
        check_vertex(lp, out, ref.obj_val)


def test_session_reuses_the_inverse_across_column_generation_rounds():
    """A second solve whose warm basis is the first solve's final basis (same rows, more columns, columns
    named by stable ids) starts from the kept inverse (warm_start_used == 2), pivots on, and ends at
    the optimum of the larger problem; a basis the session does not know falls back to the crash."""
    from smart_crossover.hip import Context
    ctx = Context(0)
    mi = workloads.mcf(60, 500, seed=17)
    V, E = mi.A.shape
    # big-M style start: artificial columns +-e_i make any subset of arcs feasible
    sign = np.where(mi.b >= 0, 1.0, -1.0)
    art = sp.diags(sign).tocsr()
    A_full = sp.hstack([mi.A, art]).tocsr()
    c_full = np.concatenate([mi.c, np.full(V, 1e6)])
    u_full = np.concatenate([mi.u, np.full(V, np.inf)])
    ids_all = np.arange(E + V, dtype=np.int64)

    def solve(cols, session, warm):
        A = A_full[:, cols]
        dA = ctx.matrix(A)
        n = cols.size
        put = lambda v: ctx.to_device(np.ascontiguousarray(v, dtype=np.float64))   # noqa: E731
        x, y = ctx.empty(n, np.float64), ctx.empty(V, np.float64)
        vb, cb = ctx.empty(n, np.int8), ctx.empty(V, np.int8)
        vin = cin = None
        if warm is not None:
            vin, cin = ctx.to_device(warm[0].astype(np.int8)), ctx.to_device(warm[1].astype(np.int8))
        res = ctx.simplex(dA, put(mi.b), put(c_full[cols]), put(np.zeros(n)), put(u_full[cols]),
                          ctx.to_device(np.zeros(V, dtype=np.uint8)), vin, cin, 0, 1e-7, 1e-7, x, y, vb, cb,
                          session=session, col_ids=ids_all[cols])
        out = (res, x.download(), vb.download().astype(int), cb.download().astype(int))
        dA.free()
        return out

    sess = ctx.simplex_session()
    first = np.concatenate([np.arange(0, E, 3), np.arange(E, E + V)])           # a third of the arcs + artificials
    r1, x1, vb1, cb1 = solve(first, sess, None)
    assert r1.status == 0 and r1.warm_start_used == 0
    second = np.arange(E + V)                                                    # every arc
    warm_vb = np.full(second.size, -1)
    warm_vb[first] = vb1                                                         # new columns non-basic at lower
    r2, x2, vb2, cb2 = solve(second, sess, (warm_vb, cb1))
    assert r2.status == 0 and r2.warm_start_used == 2
    # same optimum as a cold solve without a session, and as HiGHS
    r3, x3, _, _ = solve(second, None, None)
    assert r3.status == 0 and r2.obj == pytest.approx(r3.obj, rel=1e-9)
    from scipy.optimize import linprog
    ref = linprog(c_full, A_eq=A_full, b_eq=mi.b, bounds=list(zip(np.zeros(E + V), [None if np.isinf(v) else v for v in u_full])),
                  method="highs")
    assert ref.status == 0
    want = ref.fun
    assert r2.obj == pytest.approx(want, rel=1e-9)
    assert np.allclose(A_full @ x2, mi.b, atol=1e-6)
    # a warm basis that is not the session's: ordinary crash (1) or cold start (0), still optimal
    other_vb = np.full(second.size, -1)
    other_vb[E:] = 0
    r4, _, _, _ = solve(second, sess, (other_vb, np.full(V, -1)))
    assert r4.status == 0 and r4.warm_start_used in (0, 1) and r4.obj == pytest.approx(want, rel=1e-9)
    sess.free()
    ctx.close()


def test_infeasible_warm_basis_is_repaired_by_phase_one_not_dropped():
    """A warm basis whose basic solution violates bounds -- structurals as well as logicals -- is kept and
    driven to feasibility by phase 1 (the reference's solvers repair warm bases; lp_manager.py:79-89 hands
    over bases whose fixed-low columns are all at -1)."""
    from smart_crossover.output import Basis
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(200, 700, 5, seed=41, stratified=False, frac_upper=0.4)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    rng = np.random.default_rng(5)
    vb = np.full(700, -1)
    vb[rng.choice(700, size=200, replace=False)] = 0           # an arbitrary, almost surely infeasible basis
    vb[(vb == -1) & np.isfinite(inst.u) & (rng.random(700) < 0.3)] = -2
    cb = np.full(200, -1)
    out = solve_lp(lp, "HIP", "primal_simplex", settings(), warm_start_basis=Basis(vb, cb))
    check_vertex(lp, out, ref.obj_val)


def test_reinversion_from_the_basis_columns_keeps_the_solve_on_course():
    """spx_force_reinvert rebuilds the inverse from the columns of the current basis at every check
    (every 64 pivots here); the solve must still end at the optimum with a point that satisfies A x + s = b."""
    from smart_crossover.hip.device import default_context
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(300, 1200, 5, seed=19, stratified=False, frac_upper=0.3)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    ctx = default_context()
    ctx.set_option("spx_check", 64)
    ctx.set_option("spx_force_reinvert", 1)
    try:
        out = solve_lp(lp, "HIP", "primal_simplex", settings())
    finally:
        ctx.set_option("spx_check", 2048)
        ctx.set_option("spx_force_reinvert", 0)
    check_vertex(lp, out, ref.obj_val)
    assert out.iter_count > 64


def test_barrier_with_crossover_on_the_device_starts_from_the_interior_point():
    """method='barrier' (crossover on) with solver 'HIP': crash basis from the interior point handed over as
    warm start + primal simplex (the substitute for the reference's barrier + crossover re-solve,
    lp_methods/algorithms.py:50-54); without crossover it has no answer and says so."""
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(250, 900, 5, seed=23, stratified=False, frac_upper=0.3)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    cold = solve_lp(lp, "HIP", "barrier", SolverSettings(log_console=0))
    check_vertex(lp, cold, ref.obj_val)
    # a point close to the optimal vertex: far fewer pivots than from the slack basis
    x_near = 0.98 * ref.x + 0.02 * np.clip(ref.x + 0.1, lp.l, np.where(np.isfinite(lp.u), lp.u, ref.x + 1.0))
    warm = solve_lp(lp, "HIP", "barrier", SolverSettings(log_console=0), warm_start_solution=(x_near, ref.y))
    check_vertex(lp, warm, ref.obj_val)
    assert warm.iter_count < cold.iter_count
    with pytest.raises(NotImplementedError):
        solve_lp(lp, "HIP", "barrier", SolverSettings(log_console=0, crossover="off"))


def test_perturbation_crossover_re_solves_on_the_device():
    """run_perturb_algorithm with 'HGS+HIP': the initial barrier solve (the input of the crossover) in HiGHS,
    the perturbed sub-problem and the final warm simplex on the device."""
    from smart_crossover.lp_methods.algorithms import run_perturb_algorithm
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(600, 3000, 6, seed=61, stratified=False, frac_upper=0.3)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = run_perturb_algorithm(lp, "HGS+HIP", 1e-10, 1e-6)
    text = buf.getvalue()
    assert out.status == "OPTIMAL"
    assert "Getting and solving a perturb subproblem" in text
    if "A primal optimal BFS is found" not in text:
        check_vertex(lp, out, ref.obj_val)


@pytest.mark.parametrize("m,n,k,seed", [(37, 90, 3, 1), (300, 900, 4, 2), (1000, 2500, 5, 3)])
def test_matrix_core_fold_agrees_with_the_scalar_fold(monkeypatch, m, n, k, seed):
    """The rank-64 update of the dense inverse on the fp64 matrix cores (k_spx_fold_mfma; by default from 8192 rows
    on) forced on at small sizes whose row counts are no multiples of its 16 / 64 / 256 tiles: same optimum as HiGHS
    and as the scalar fold, the final basis a fixed point."""
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=False, frac_upper=0.4)
    lp = general_lp(inst)
    ref = solve_lp(lp, "HGS", "default", settings())
    assert ref.status == "OPTIMAL"
    outs = {}
    for mode in ("1", "1000000"):
        monkeypatch.setenv("SX_SPX_FOLD_MFMA_MIN", mode)
        out = solve_lp(lp, "HIP", "primal_simplex", settings())
        check_vertex(lp, out, ref.obj_val)
        again = solve_lp(lp, "HIP", "default", settings(), warm_start_basis=out.basis)
        assert again.iter_count == 0
        outs[mode] = out
    assert outs["1"].obj_val == pytest.approx(outs["1000000"].obj_val, rel=1e-10)
