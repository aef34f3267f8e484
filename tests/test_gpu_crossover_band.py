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


def test_config5_size_1e6_rows(ctx, monkeypatch):
    """BASELINE configs[4]'s size: netlib_lp(1e6, 1e7), 8e7 entries -- interior point to optimal vertex + basis of the
    perturbed 1e6 x 1e6 sub-problem on the GPU (first-order stage + sparse crossover on the bordered factorisation: band in
    32 blocks, Schur complement of ~13,800 rows), with the certificates of the headline test and the reference's gap test.
    (Sparse matrix products on the host check a 1e6-row vertex in seconds; HiGHS has no answer at this size.)"""
    from smart_crossover.lp_methods import algorithms as alg
    inst = workloads.netlib_lp(1_000_000, 10_000_000)
    lp, mgr = perturbed_sub_problem(inst)
    assert mgr.lp_sub.A.shape[0] == 1_000_000
    caller, out = resolve(mgr, inst, monkeypatch, "auto")
    assert caller.solved_by == "crossover_band" and out.status == "OPTIMAL"
    assert int(out.iter_count) < 5000          # (round 3: 8,273 -- one per linking row -- on a 66-99 GB tableau)
    certificates(mgr.lp_sub, out)
    with redirect_stdout(io.StringIO()):
        assert alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x)) is True


def small_band_lp(m=400, n=1600, seed=11):
    """A well-conditioned staircase LP in computational form for direct calls of the sparse crossover."""
    inst = workloads.netlib_lp(m, n, seed=seed)
    return inst


def direct(ctx, A, b, c, l, u, lt, x_start, vbasis_in=None, cbasis_in=None):
    m, n = A.shape
    dA = ctx.matrix(A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    d_vb, d_cb = ctx.empty(n, np.int8), ctx.empty(m, np.int8)
    res = ctx.crossover_band(dA, put(b), put(c), put(l), put(u), put(lt, np.uint8), put(x_start), 0, 1e-7, 1e-7, d_x, d_y, d_vb, d_cb,
                             vbasis_in=None if vbasis_in is None else put(vbasis_in, np.int8),
                             cbasis_in=None if cbasis_in is None else put(cbasis_in, np.int8))
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


def vertex_checks(A, b, c, l, u, lt, x, y, vb, cb, tol=1e-6):
    s_p = b - A @ x
    assert np.abs(s_p[~lt]).max(initial=0.0) <= tol and s_p[lt].min(initial=0.0) >= -tol
    assert np.all(x >= l - 1e-9) and np.all(x <= u + 1e-9)
    rc = c - A.T @ y
    assert np.all(rc[vb == -1] >= -tol) and np.all(rc[vb == -2] <= tol) and np.abs(rc[vb == 0]).max(initial=0.0) <= tol
    assert int((vb == 0).sum() + (cb == 0).sum()) == A.shape[0] and not np.any(vb == -3)


def test_a_given_optimal_basis_is_factored_as_it_is_and_needs_no_pivot(ctx):
    """sx_crossover_band_basis_dev, the warm-started simplex of the reference's last step (lp_methods/algorithms.py:69-74)
    on the bordered factorisation: the optimal basis of a first run handed back -> every member is in the factors (none is
    re-guessed), 0 iterations, the same vertex; then the same basis with a few members swapped for wrong ones and with
    members missing (completed by logicals) -> a few pivots back to the same optimum."""
    inst = small_band_lp(1200, 6000, seed=21)
    lt = inst.sense == "<"
    first = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, inst.x)
    assert int(first[0].status) == 0 and int(first[0].iters) > 0
    _, x0, y0, vb0, cb0 = first
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, x0, vb0, cb0)
    assert int(res.status) == 0 and int(res.iters) == 0 and int(res.warm_start_used) == 1
    assert np.array_equal(vb, vb0) and np.array_equal(cb, cb0)
    np.testing.assert_allclose(x, x0, rtol=1e-9, atol=1e-10)
    obj0 = float(inst.c @ x0)
    rng = np.random.default_rng(4)
    # (a) members swapped: 15 basic columns declared non-basic at their lower bound, 15 non-basic ones declared basic
    vb1 = vb0.copy()
    out_ = rng.choice(np.flatnonzero(vb0 == 0), 15, replace=False)
    in_ = rng.choice(np.flatnonzero(vb0 == -1), 15, replace=False)
    vb1[out_], vb1[in_] = -1, 0
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, x0, vb1, cb0)
    assert int(res.status) == 0 and 0 < int(res.iters) <= int(first[0].iters)      # (no worse than from the point alone)
    assert float(inst.c @ x) == pytest.approx(obj0, rel=1e-9, abs=1e-9)
    vertex_checks(inst.A, inst.b, inst.c, inst.l, inst.u, lt, x, y, vb, cb)
    # (b) members missing: 25 basic columns dropped and nothing put in their place
    vb2 = vb0.copy()
    vb2[rng.choice(np.flatnonzero(vb0 == 0), 25, replace=False)] = -1
    res, x, y, vb, cb = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, x0, vb2, cb0)
    assert int(res.status) == 0
    assert float(inst.c @ x) == pytest.approx(obj0, rel=1e-9, abs=1e-9)
    vertex_checks(inst.A, inst.b, inst.c, inst.l, inst.u, lt, x, y, vb, cb)


def test_final_warm_simplex_of_run_perturb_algorithm_takes_the_sparse_path(ctx, monkeypatch):
    """run_perturb_algorithm forced down its last branch (lp_methods/algorithms.py:69-74: the gap test answers None) on
    netlib_lp(20000, 200000): the perturbed re-solve AND the warm-started primal simplex on the ORIGINAL 20,000 x 200,000 LP
    run on the device, the latter from the recovered basis through the sparse path (a dense inverse of the original LP
    would be 3.2 GB installed by 20,000 crash pivots).  The interior point is the instance's own strictly complementary
    pair (HiGHS' barrier does not finish on this family, profiles/r03): the optimal value is c^T x of that pair."""
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.output import Output
    from smart_crossover.solver_caller import hip as hipmod
    inst = workloads.netlib_lp(20000, 200000, seed=3)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    real_solve = alg.solve_lp
    seen = []

    def fake_solve(lp_, solver="GRB", method="default", settings=None, **kw):
        if method == "barrier" and settings is not None and settings.crossover == "off":
            return Output(x=inst.x, y=inst.y, obj_val=float(inst.c @ inst.x), status="OPTIMAL")
        out = real_solve(lp_, solver, method, settings=settings, **kw)
        seen.append((method, lp_.A.shape, "warm_start_basis" in kw and kw["warm_start_basis"] is not None, out.iter_count))
        return out

    band_calls = []
    real_band = hipmod.HipCaller._band

    def spy_band(self, *a, **k):
        r = real_band(self, *a, **k)
        band_calls.append((self._A.shape, None if r is None else int(r.status), None if r is None else int(r.iters)))
        return r

    monkeypatch.setattr(alg, "solve_lp", fake_solve)
    monkeypatch.setattr(alg, "check_feasibility_problem", lambda lp_: False)
    monkeypatch.setattr(alg, "check_perturb_output_precision", lambda *a, **k: None)
    monkeypatch.setattr(hipmod.HipCaller, "_band", spy_band)
    monkeypatch.setenv("SX_LP_CROSSOVER", "band")
    with redirect_stdout(io.StringIO()):
        out = alg.run_perturb_algorithm(lp, "HIP")
    assert out.status == "OPTIMAL"
    assert [c[0] for c in seen] == ["barrier", "primal_simplex"] and seen[1][1] == (20000, 200000) and seen[1][2]
    assert len(band_calls) == 2 and band_calls[1][0] == (20000, 200000) and band_calls[1][1] == 0     # both re-solves on the sparse path
    assert band_calls[1][2] <= 200          # the recovered basis is (next to) optimal for the original LP: a few pivots at most
    lt = inst.sense == "<"
    vertex_checks(inst.A, inst.b, inst.c, inst.l, inst.u, lt, out.x, out.y, out.basis.vbasis, out.basis.cbasis)
    assert float(inst.c @ out.x) == pytest.approx(float(inst.c @ inst.x), rel=1e-7, abs=1e-7)


def test_warm_start_at_headline_size_returns_without_a_pivot(ctx, monkeypatch):
    """A 1e5-row warm start from the optimal basis (the perturbed sub-problem of the 1e6-variable LP): factored as given,
    0 iterations."""
    from smart_crossover.solver_caller import solving
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.netlib_lp()
    lp, mgr = perturbed_sub_problem(inst)
    caller, out = resolve(mgr, inst, monkeypatch, "auto")
    assert caller.solved_by == "crossover_band" and out.status == "OPTIMAL"
    with redirect_stdout(io.StringIO()):
        again = solving.solve_lp(mgr.lp_sub, "HIP", "primal_simplex", SolverSettings(presolve="on", log_console=0),
                                 warm_start_basis=out.basis, warm_start_solution=(out.x, out.y))
    assert again.status == "OPTIMAL" and int(again.iter_count) == 0
    np.testing.assert_allclose(again.x, out.x, rtol=1e-9, atol=1e-9)
    assert np.array_equal(again.basis.vbasis, out.basis.vbasis) and np.array_equal(again.basis.cbasis, out.basis.cbasis)


def test_infeasible_perturbation_widens_the_face_and_retries(ctx, monkeypatch):
    """run_perturb_algorithm's retry loop (lp_methods/algorithms.py:45-61) through K16p + K16s at 30,000 rows: the first
    re-solve is INFEASIBLE -- at gamma = 1e-3 the indicator test fixes EVERY column of one equality row at a bound (their
    dual slacks are made large: c_k += 2e3 (x_k - l_k) in the LP handed over), the row is left empty with a right-hand side
    that is not zero --, gamma is multiplied by 1e-5 (those columns stay free: 1e-8 * 2e3 < 1) and the second re-solve
    succeeds."""
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.output import Output
    inst = workloads.netlib_lp(30000, 300000, seed=9)
    lt = inst.sense == "<"
    A = inst.A.tocsr()
    i = int(np.flatnonzero(~lt)[1000])
    cols = A[i].indices
    interior = cols[(inst.x[cols] - inst.l[cols] > 1e-6) & (inst.u[cols] - inst.x[cols] > 1e-6)]
    assert interior.size >= 1
    lp = GeneralLP(inst.A, inst.b, inst.c.copy(), inst.l, inst.u, inst.sense)
    lp.c[interior] += 2e3 * (inst.x[interior] - inst.l[interior])
    real_solve = alg.solve_lp
    statuses = []

    def fake_solve(lp_, solver="GRB", method="default", settings=None, **kw):
        if method == "barrier" and settings is not None and settings.crossover == "off":
            return Output(x=inst.x, y=inst.y, obj_val=float(lp.c @ inst.x), status="OPTIMAL")
        out = real_solve(lp_, solver, method, settings=settings, **kw)
        statuses.append((method, lp_.A.shape, out.status))
        return out

    monkeypatch.setattr(alg, "solve_lp", fake_solve)
    monkeypatch.setattr(alg, "check_feasibility_problem", lambda lp_: False)
    monkeypatch.setattr(alg, "check_perturb_output_precision", lambda *a, **k: True)   # (the changed costs are not the point here)
    monkeypatch.setenv("SX_LP_CROSSOVER", "band")
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = alg.run_perturb_algorithm(lp, "HIP")
    assert [(st[0], st[2]) for st in statuses] == [("barrier", "INFEASIBLE"), ("barrier", "OPTIMAL")]
    assert statuses[1][1][1] > statuses[0][1][1]          # the wider face keeps more columns
    assert "Increasing the optimal face and try again" in buf.getvalue()
    assert out.status == "OPTIMAL"


def wide_band_lp(m=4000, per_row=2, below=1000, above=960, seed=31):
    """A consistent LP whose matched basis is a band with kl = `below` and ku = `above`: every column has its large entry in
    its own row and two small ones `below` rows further down / `above` rows further up; one column per row is basic."""
    rng = np.random.default_rng(seed)
    n = per_row * m
    own = np.arange(n) // per_row
    rows = np.stack([own, np.clip(own + below, 0, m - 1), np.clip(own - above, 0, m - 1)], axis=1)
    vals = np.stack([rng.uniform(0.8, 1.2, n), rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n)], axis=1)
    vals[rows[:, 1] == own, 1] = 0.0
    vals[rows[:, 2] == own, 2] = 0.0
    import scipy.sparse as sp
    A = sp.coo_matrix((vals.ravel(), (rows.ravel(), np.repeat(np.arange(n), 3))), shape=(m, n)).tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    basic = np.zeros(n, dtype=bool)
    basic[np.arange(m) * per_row + rng.integers(0, per_row, m)] = True
    x = np.where(basic, rng.uniform(0.1, 1.0, n), 1e-9)
    y = rng.standard_normal(m)
    s_d = np.where(basic, 1e-10, np.abs(rng.standard_normal(n)))
    return workloads.LPInstance(A=A, b=A @ x, c=A.T @ y + s_d, l=np.zeros(n), u=np.full(n, np.inf), sense=np.full(m, "="), x=x, y=y)


def test_a_band_between_the_two_width_limits_is_refused_not_crashed(ctx, monkeypatch):
    """kl + 32 <= 1536 but kl + ku + 32 = 2,024 > 1,980: round 3 accepted this basis in the crossover (limit 2,400) and then
    failed inside sx_bandlu_create_dev with SX_ERR_INVALID -- a ValueError the 'HIP' backend did not catch.  Both now ask
    one function (sx_bandlu_supports): SX_ERR_UNSUPPORTED, and the backend falls back on the dense crossover."""
    from smart_crossover.hip.device import BandLU
    put = lambda v, t: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    one = (put(np.zeros(1), np.int32), put(np.zeros(1), np.int32), put(np.ones(1), np.float64))
    BandLU(ctx, 3000, 1000, 900, *one).free()                      # 1,932 rows of LDS window: accepted
    with pytest.raises(ValueError):
        BandLU(ctx, 3000, 1000, 992, *one)                         # 2,024: refused by the band LU itself
    # an LP whose basis has that shape in the natural order of its rows (three diagonals 1,000 below / 960 above): the
    # crossover either finds a narrower band (its Cuthill-McKee order folds the diagonals together) and solves it, or
    # refuses -- never the ValueError of round 3; through the backend the vertex is reached on one path or the other
    inst = wide_band_lp()
    lt = inst.sense == "<"
    try:
        res = direct(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, lt, inst.x)[0]
        assert int(res.status) == 0
    except NotImplementedError:
        pass
    lp, mgr = perturbed_sub_problem(inst)
    caller, out = resolve(mgr, inst, monkeypatch, "band")
    assert caller.solved_by in ("crossover_band", "simplex") and out.status == "OPTIMAL"
    certificates(mgr.lp_sub, out)


@pytest.mark.parametrize("knob", ["window_4096", "uniform_entries"])
def test_hostile_structure_reaches_highs_optimum_on_one_path_or_the_other(ctx, monkeypatch, knob, record_property):
    """One case per knob of the generator that round 3 tuned until every basis was a narrow, column-dominant band: a window
    of 4,096 rows (at 3,000 rows: every column reaches a third of the matrix -> kl + ku ~ 1,100, still a band the LU takes)
    and entries drawn U(-1, 1) on the same pattern (no dominant own-row entry: ill-conditioned bases, the band LU and the
    dense LU set columns aside).  Whichever path answers -- the sparse crossover, or the dense one behind its refusal or
    its failure -- the vertex is HiGHS' and certified; which one it was is recorded."""
    from smart_crossover.lp_methods import algorithms as alg
    if knob == "window_4096":
        inst = workloads.netlib_lp(3000, 30000, seed=12, window=4096)
    else:
        base = workloads.netlib_lp(3000, 30000, seed=13)
        rng = np.random.default_rng(14)
        A = base.A.copy()
        A.data = rng.uniform(-1.0, 1.0, A.data.size)
        slack, s_d = base.b - base.A @ base.x, base.c - base.A.T @ base.y
        inst = workloads.LPInstance(A=A, b=A @ base.x + slack, c=A.T @ base.y + s_d, l=base.l, u=base.u, sense=base.sense, x=base.x, y=base.y)
    lp, mgr = perturbed_sub_problem(inst)
    caller, out = resolve(mgr, inst, monkeypatch, "band")
    record_property("solved_by", caller.solved_by)
    print(f"[hostile structure] {knob}: solved by {caller.solved_by}, {int(out.iter_count)} iterations")
    assert caller.solved_by in ("crossover_band", "simplex") and out.status == "OPTIMAL"
    sub = mgr.lp_sub
    certificates(sub, out)
    lt = np.asarray(sub.sense) == "<"
    ref = linprog(sub.c, A_ub=sub.A[lt], b_ub=sub.b[lt], A_eq=sub.A[~lt], b_eq=sub.b[~lt], bounds=np.c_[sub.l, sub.u], method="highs")
    assert ref.status == 0
    assert float(sub.c @ out.x) == pytest.approx(ref.fun, rel=1e-7, abs=1e-8)


def test_probe_answers_before_the_first_order_stage_is_sized(ctx, monkeypatch):
    """sx_crossover_band_probe_dev: the sparse crossover's set-up up to its band-width check.  A staircase LP is taken, an
    unstructured one is not -- and the backend that asked sizes the first-order stage for the crossover that will run: the
    full 20,000 iterations in front of the dense one (config 2's shape: 239,796 pivots behind 5,000 iterations, 692 behind
    20,000), m / 20 (at least 5,000) in front of the sparse one."""
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    takes = {}
    for name, inst in (("staircase", workloads.netlib_lp(3000, 30000, seed=2)),
                       ("unstructured", workloads.sparse_lp(2000, 8000, 6, seed=9, stratified=True))):
        lp, mgr = perturbed_sub_problem(inst)
        sub = mgr.lp_sub
        dA = ctx.matrix(sub.A)
        takes[name] = ctx.crossover_band_takes(dA, put(sub.b), put(sub.c), put(sub.l), put(sub.u), put(np.asarray(sub.sense) == "<", np.uint8),
                                               put(np.clip(mgr.get_subx(inst.x), sub.l, sub.u)))
        dA.free()
        caller, out = resolve(mgr, inst, monkeypatch, "auto")
        assert out.status == "OPTIMAL"
        certificates(sub, out)
        assert caller.solved_by == ("crossover_band" if takes[name] else "simplex")
        assert int(caller.pdlp.iters) <= (5056 if takes[name] else 20032)
        if not takes[name]:
            assert int(caller.pdlp.iters) > 5056 or int(caller.pdlp.status) == 0      # (the full budget, unless it converged before)
    assert takes == {"staircase": True, "unstructured": False}
