"""GPU: the sparse crossover (K16s, csrc/sx_crossover_band.hip: position matching + band LU + tableau of the tracked
columns) behind the first-order stage (K16p) -- the re-solve of the perturbed sub-problem
(reference lp_methods/algorithms.py:50-54: barrier + crossover inside Gurobi, parity unpinned).  Checked the way
tests/test_gpu_full_size.py checks the dense crossover: HiGHS' optimal value where HiGHS is quick, and
solver-independent certificates everywhere (primal / dual feasibility by basis status at the reference's
tolerances, |B| = m, the reference's own gap test at 1e-8)."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
from scipy.optimize import linprog

import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def perturbed_sub_problem(inst):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    with redirect_stdout(io.StringIO()):
        mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
    return lp, mgr


def resolve(mgr, inst, monkeypatch, mode):
    from smart_crossover.solver_caller import solving
    from smart_crossover.solver_caller.caller import SolverSettings
    monkeypatch.setenv("SX_LP_CROSSOVER", mode)
    caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
    caller.read_genlp(mgr.lp_sub)
    caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
    with redirect_stdout(io.StringIO()):
        caller.run_barrier()
        out = caller.return_output()
    return caller, out


def certificates(sub, out, tol=1e-6):
    m, n = sub.A.shape
    x, y, vb, cb = out.x, out.y, out.basis.vbasis, out.basis.cbasis
    s_p = sub.b - sub.A @ x
    lt = np.asarray(sub.sense) == "<"
    assert np.all(np.abs(s_p[~lt]) <= tol) and np.all(s_p[lt] >= -tol)
    assert np.all(x >= sub.l - tol) and np.all(x <= sub.u + tol)
    rc = sub.c - sub.A.T @ y
    assert np.all(rc[vb == -1] >= -tol) and np.all(rc[vb == -2] <= tol) and np.all(np.abs(rc[vb == 0]) <= tol)
    assert np.all(y[lt & (cb == -1)] <= tol) and np.all(np.abs(y[cb == 0]) <= tol)
    assert int(np.count_nonzero(vb == 0) + np.count_nonzero(cb == 0)) == m
    assert not np.any(vb == -3)
    assert np.all(x[vb == -1] == sub.l[vb == -1]) and np.all(x[vb == -2] == sub.u[vb == -2])   # non-basic: AT the bound


@pytest.mark.parametrize("m,n,seed", [(300, 3000, 1), (3000, 30000, 2), (20000, 200000, 3)])
def test_band_crossover_reaches_the_vertex(ctx, monkeypatch, m, n, seed):
    from smart_crossover.lp_methods import algorithms as alg
    inst = workloads.netlib_lp(m, n, seed=seed)
    lp, mgr = perturbed_sub_problem(inst)
    caller, out = resolve(mgr, inst, monkeypatch, "band")
    assert caller.solved_by == "crossover_band" and out.status == "OPTIMAL"
    certificates(mgr.lp_sub, out)
    with redirect_stdout(io.StringIO()):
        assert alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x)) is True   # gap < 1e-8
    if m <= 3000:
        sub = mgr.lp_sub
        lt = np.asarray(sub.sense) == "<"
        ref = linprog(sub.c, A_ub=sub.A[lt], b_ub=sub.b[lt], A_eq=sub.A[~lt], b_eq=sub.b[~lt], bounds=np.c_[sub.l, sub.u],
                      method="highs")
        assert ref.status == 0
        assert float(sub.c @ out.x) == pytest.approx(ref.fun, rel=1e-8, abs=1e-9)
        # the perturbed LP has ONE optimal vertex: same point, hence the same basic set where it is non-degenerate
        np.testing.assert_allclose(out.x, ref.x, rtol=1e-6, atol=1e-7)


def test_band_and_dense_crossover_agree(ctx, monkeypatch):
    inst = workloads.netlib_lp(1500, 15000, seed=5)
    lp, mgr = perturbed_sub_problem(inst)
    c1, o1 = resolve(mgr, inst, monkeypatch, "band")
    c2, o2 = resolve(mgr, inst, monkeypatch, "dense")
    assert c1.solved_by == "crossover_band" and c2.solved_by == "simplex"
    assert o1.status == o2.status == "OPTIMAL"
    np.testing.assert_allclose(o1.x, o2.x, rtol=1e-7, atol=1e-8)
    assert float(mgr.lp_sub.c @ o1.x) == pytest.approx(float(mgr.lp_sub.c @ o2.x), rel=1e-10)
    # ONE optimal vertex: the two basic sets may differ only in degenerate positions (a basic variable AT a bound / a
    # basic logical of an active row), and those are named, not waved through
    sub = mgr.lp_sub
    dv = (o1.basis.vbasis == 0) != (o2.basis.vbasis == 0)
    dc = (o1.basis.cbasis == 0) != (o2.basis.cbasis == 0)
    at_bound = (np.abs(o1.x - sub.l) <= 1e-8) | (np.abs(sub.u - o1.x) <= 1e-8)
    active = np.abs(sub.b - sub.A @ o1.x) <= 1e-8
    assert at_bound[dv].all() and active[dc].all(), "basic sets differ at a non-degenerate position"


def test_unstructured_basis_is_refused_and_the_caller_falls_back(ctx, monkeypatch):
    """A random sparse LP has no band structure in its natural order: SX_ERR_UNSUPPORTED from the sparse crossover,
    and the 'HIP' backend then runs the dense one."""
    inst = workloads.sparse_lp(2000, 8000, 6, seed=9, stratified=True)
    lp, mgr = perturbed_sub_problem(inst)
    sub = mgr.lp_sub
    dA = ctx.matrix(sub.A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    with pytest.raises(NotImplementedError):
        ctx.crossover_band(dA, put(sub.b), put(sub.c), put(sub.l), put(sub.u), put(np.asarray(sub.sense) == "<", np.uint8),
                           put(mgr.get_subx(inst.x)))
    dA.free()
    caller, out = resolve(mgr, inst, monkeypatch, "band")
    assert caller.solved_by == "simplex" and out.status == "OPTIMAL"
    certificates(sub, out)


def test_headline_size_1e6_variables(ctx, monkeypatch):
    """The configuration BASELINE.json's metric is quoted on: a 1e6-variable netlib-style LP (1e5 rows), interior point
    to optimal vertex + basis of the perturbed 1e5 x 1e5 sub-problem, all on the GPU."""
    from smart_crossover.lp_methods import algorithms as alg
    inst = workloads.netlib_lp()
    lp, mgr = perturbed_sub_problem(inst)
    assert mgr.lp_sub.A.shape[0] == 100_000
    caller, out = resolve(mgr, inst, monkeypatch, "auto")
    assert caller.solved_by == "crossover_band" and out.status == "OPTIMAL"
    certificates(mgr.lp_sub, out)
    with redirect_stdout(io.StringIO()):
        assert alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x)) is True


def small_band_lp(m=400, n=1600, seed=11):
    """A well-conditioned staircase LP in computational form for direct calls of the sparse crossover."""
    inst = workloads.netlib_lp(m, n, seed=seed)
    return inst


def direct(ctx, A, b, c, l, u, lt, x_start):
    m, n = A.shape
    dA = ctx.matrix(A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    d_vb, d_cb = ctx.empty(n, np.int8), ctx.empty(m, np.int8)
    res = ctx.crossover_band(dA, put(b), put(c), put(l), put(u), put(lt, np.uint8), put(x_start), 0, 1e-7, 1e-7, d_x, d_y, d_vb, d_cb)
    out = (res, d_x.download(), d_y.download(), d_vb.download().astype(int), d_cb.download().astype(int))
    dA.free()
    return out


def test_direct_call_from_a_poor_start_reaches_highs_optimum(ctx):
    """No first-order stage in front: the interior point of the ORIGINAL problem as the start of the crossover of the
    whole LP (all 1,600 columns; most of them non-basic at a bound) -- phase 1, superbasic pushes and column
    generation all have to work.  Optimum = HiGHS', certificates hold."""
    inst = small_band_lp()
    lt = inst.sense == "<"
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, inst.x)
    assert int(res.status) == 0
    ref = linprog(inst.c, A_ub=inst.A[lt], b_ub=inst.b[lt], A_eq=inst.A[~lt], b_eq=inst.b[~lt],
                  bounds=list(zip(inst.l, [None if np.isinf(v) else v for v in inst.u])), method="highs")
    assert ref.status == 0
    assert float(inst.c @ x) == pytest.approx(ref.fun, rel=1e-8, abs=1e-8)
    m = inst.A.shape[0]
    s_p = inst.b - inst.A @ x
    assert np.abs(s_p[~lt]).max() < 1e-6 and s_p[lt].min() > -1e-6
    assert np.all(x >= inst.l - 1e-9) and np.all(x <= inst.u + 1e-9)
    rc = inst.c - inst.A.T @ y
    assert np.all(rc[vb == -1] >= -1e-6) and np.all(rc[vb == -2] <= 1e-6) and np.abs(rc[vb == 0]).max() < 1e-6
    assert int((vb == 0).sum() + (cb == 0).sum()) == m


def test_infeasible_and_unbounded_are_reported(ctx):
    inst = small_band_lp(seed=12)
    lt = inst.sense == "<"
    # infeasible: one '=' row asks for more than its columns' bounds can give
    b = inst.b.copy()
    i = int(np.flatnonzero(~lt)[5])
    u = np.where(np.isinf(inst.u), 50.0, inst.u)
    b[i] = 1e6
    res = direct(ctx, inst.A, b, inst.c, inst.l, u, lt, np.clip(inst.x, inst.l, u))[0]
    assert int(res.status) == 1
    # unbounded: a column without an upper bound that touches only '<' rows with negative entries and pays to grow
    A = inst.A.tolil()
    j = int(np.flatnonzero(np.isinf(inst.u))[3])
    rows_lt = np.flatnonzero(lt)[:3]
    A[:, j] = 0
    for r in rows_lt:
        A[r, j] = -1.0
    A = A.tocsr()
    c = inst.c.copy()
    c[j] = -1.0
    res = direct(ctx, A, inst.A @ inst.x + np.where(lt, 0.5, 0.0), c, inst.l, inst.u, lt, inst.x)[0]
    assert int(res.status) == 2


def test_shuffled_rows_fall_back_into_a_band(ctx, monkeypatch):
    """The same staircase LP with its rows (and columns) shuffled: no band in the natural order; the Cuthill-McKee
    order of the rows finds it again and the sparse crossover reaches the vertex of the un-shuffled problem."""
    from smart_crossover.lp_methods import algorithms as alg
    inst = workloads.netlib_lp(4000, 40000, seed=8)
    rng = np.random.default_rng(1)
    pr, pc = rng.permutation(inst.A.shape[0]), rng.permutation(inst.A.shape[1])
    shuf = workloads.LPInstance(A=inst.A[pr][:, pc].tocsr(), b=inst.b[pr], c=inst.c[pc], l=inst.l[pc], u=inst.u[pc],
                                sense=inst.sense[pr], x=inst.x[pc], y=inst.y[pr])
    shuf.A.sort_indices()
    lp0, mgr0 = perturbed_sub_problem(inst)
    lp1, mgr1 = perturbed_sub_problem(shuf)
    c0, o0 = resolve(mgr0, inst, monkeypatch, "band")
    c1, o1 = resolve(mgr1, shuf, monkeypatch, "band")
    assert c0.solved_by == c1.solved_by == "crossover_band"
    assert o0.status == o1.status == "OPTIMAL"
    certificates(mgr1.lp_sub, o1)
    with redirect_stdout(io.StringIO()):
        assert alg.check_perturb_output_precision(mgr1, o1.x, lp1.c, float(lp1.c @ shuf.x)) is True
    # same LP up to the permutation (the perturbation xi is drawn per position, so only the original objectives agree)
    x0, x1 = mgr0.get_orix(o0.x), mgr1.get_orix(o1.x)
    assert float(inst.c @ x0) == pytest.approx(float(shuf.c @ x1), rel=1e-7)


def test_a_full_eta_file_restarts_with_a_fresh_factorisation(ctx, monkeypatch):
    """SX_BAND_EPOCH = 40: the eta file holds 40 basis changes, so the run has to stop, match and factor the current
    basis afresh and go on (several epochs); same optimum as with the default capacity."""
    inst = small_band_lp()
    lt = inst.sense == "<"
    want = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, inst.x)
    assert int(want[0].status) == 0 and int(want[0].iters) > 100
    monkeypatch.setenv("SX_BAND_EPOCH", "40")
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, inst.x)
    assert int(res.status) == 0
    assert float(inst.c @ x) == pytest.approx(float(inst.c @ want[1]), rel=1e-9, abs=1e-9)
    s_p = inst.b - inst.A @ x
    assert np.abs(s_p[~lt]).max() < 1e-6 and s_p[lt].min() > -1e-6
    assert int((vb == 0).sum() + (cb == 0).sum()) == inst.A.shape[0]


@pytest.mark.parametrize("seed,scale", [(4, 0.3), (5, 1.0)])
def test_another_cost_brings_columns_in_by_pricing(ctx, seed, scale):
    """The interior point of the original LP as the start for a DIFFERENT cost vector: the columns the optimum needs are
    non-basic at a bound at the start, so pricing over all columns has to bring them into the tableau after the first
    basis changes (duals through the eta file, new columns through it 64 etas at a time).  Optimum = HiGHS'."""
    inst = small_band_lp()
    lt = inst.sense == "<"
    rng = np.random.default_rng(seed)
    c = inst.c + scale * rng.standard_normal(inst.c.size)
    u = np.where(np.isinf(inst.u), 30.0, inst.u)            # (bounded: any cost has an optimum)
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, c, inst.l, u, lt, np.clip(inst.x, inst.l, u))
    assert int(res.status) == 0
    assert int(res.phase1_iters) > 0        # (the field counts the columns that pricing brought into the tableau)
    ref = linprog(c, A_ub=inst.A[lt], b_ub=inst.b[lt], A_eq=inst.A[~lt], b_eq=inst.b[~lt], bounds=list(zip(inst.l, u)), method="highs")
    assert ref.status == 0
    assert float(c @ x) == pytest.approx(ref.fun, rel=1e-8, abs=1e-8)
    s_p = inst.b - inst.A @ x
    assert np.abs(s_p[~lt]).max() < 1e-6 and s_p[lt].min() > -1e-6
    assert np.all(x >= inst.l - 1e-9) and np.all(x <= u + 1e-9)
    rc = c - inst.A.T @ y
    assert np.all(rc[vb == -1] >= -1e-6) and np.all(rc[vb == -2] <= 1e-6) and np.abs(rc[vb == 0]).max() < 1e-6
    assert int((vb == 0).sum() + (cb == 0).sum()) == inst.A.shape[0]
