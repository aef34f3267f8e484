"""GPU: the drop-in Python API of the perturbation crossover (smart_crossover.lp_methods.*,
smart_crossover.formats) against the reference goldens, the oracle and solver-independent
certificates.  The re-solves use the HiGHS stand-in backend ('HGS')."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal, csr_from, same_csr
from oracle import lp_path as L
import workloads

pytestmark = pytest.mark.gpu


def make_lp(g):
    from smart_crossover.formats import GeneralLP
    return GeneralLP(csr_from(g, "A"), g["b"].copy(), g["c"].copy(), g["l"].copy(), g["u"].copy(), g["sense"].copy())


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


@pytest.mark.parametrize("gname,rel", [("g1", 1e-9), ("g2", 1e-5)])
def test_get_perturb_problem_matches_reference_golden(gname, rel, request):
    from smart_crossover.lp_methods import algorithms as alg
    g = request.getfixturevalue(gname)
    lp = make_lp(g)
    gamma, gamma_dual = g["gamma"]
    mgr, text = quiet(alg.get_perturb_problem, lp, g["x"], g["y"], gamma, gamma_dual, False)
    assert f"The number of fixed variables is {g['fix'].size}." in text
    assert f"The number of fixed constraints is {g['fixed_rows'].size}." in text
    for key in ("fix_low", "fix_up", "non_fix", "fix"):
        assert mgr.var_info[key].dtype == np.int64 and np.array_equal(mgr.var_info[key], g[key]), key
    assert np.array_equal(mgr.fixed_constraints, g["fixed_rows"])
    assert mgr.perturb_info["scale_factor"] == pytest.approx(float(g["sf"]), rel=rel)
    p_max = float(np.max(np.abs(g["c_pt_opt"] - g["c"])))
    np.testing.assert_allclose(mgr.lp.c, g["c_pt_opt"], rtol=0, atol=10 * rel * p_max + 1e-15)
    sub = mgr.lp_sub
    assert same_csr(sub.A, csr_from(g, "Asub"))
    assert bits_equal(sub.b, g["b_sub"]) and bits_equal(sub.l, g["l_sub"]) and bits_equal(sub.u, g["u_sub"])
    np.testing.assert_allclose(sub.c, g["c_sub"], rtol=0, atol=10 * rel * p_max + 1e-15)
    assert np.array_equal(sub.sense, g["sense_sub"])
    assert not np.array_equal(lp.c, mgr.lp.c) and bits_equal(lp.c, g["c"])        # caller's LP untouched
    # recovery helpers (host glue, quirk Q1 included)
    assert bits_equal(mgr.get_subx(g["x"]), g["sub_of_x"])
    assert bits_equal(mgr.recover_x_from_sub_x(g["x_sub"]), g["recover_x"])
    assert bits_equal(mgr.get_orix(g["x_sub"]), g["orix"])
    from smart_crossover.output import Basis
    rb = mgr.recover_basis_from_sub_basis(Basis(g["vb_sub"], g["cb_sub"]))
    assert np.array_equal(rb.vbasis, g["recover_vb"]) and np.array_equal(rb.cbasis, g["recover_cb"])
    obj = float(lp.c @ mgr.get_orix(g["x_sub"]))
    (ok, _), (bad, _) = (quiet(alg.check_perturb_output_precision, mgr, g["x_sub"], lp.c, v)
                         for v in (obj * (1 + 1e-12), obj * 1.5 + 1.0))
    assert (ok is True) == bool(g["gap_flags"][0]) and (bad is None) == bool(g["gap_flags"][1])


@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_feasibility_branch_and_plain_functions(gname, request):
    from smart_crossover.lp_methods import algorithms as alg
    g = request.getfixturevalue(gname)
    lp = make_lp(g)
    mgr, _ = quiet(alg.get_perturb_problem, lp, g["x"], g["y"], 1e-3, 1e-3, True)
    assert bits_equal(mgr.lp.c, g["c_pt_feas"])
    assert bits_equal(alg.perturb_c(lp, g["x"], True), g["c_pt_feas"])
    assert bits_equal(lp.get_dual_slack(g["y"]), g["s_d"])
    assert bits_equal(lp.get_primal_slack(g["x"]), g["s_p"])
    assert bits_equal(lp.get_standard_x(g["x"]), g["std_x_of_x"])
    assert bits_equal(alg.get_x_perturb_val(lp, g["x"]), g["x_min_raw"])
    assert same_csr(lp.get_standard_A(), csr_from(g, "Astd")) and bits_equal(lp.get_standard_c(), g["std_c"])


def test_projector_functions_against_oracle(g1):
    from smart_crossover.lp_methods import algorithms as alg
    lp = make_lp(g1)
    xr = g1["x_real"]
    want, _ = L.projector_Xc(lp.A, lp.b, lp.c, lp.l, lp.u, lp.sense, xr, explicit=True)
    got = alg.get_projector_Xc(lp, xr)
    # both sides are CG iterates stopped at a 1e-8 relative residual: the vectors agree to ~1e-7 of
    # their norm componentwise, the norm itself (what feeds the scale factor) much tighter
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6 * np.linalg.norm(want))
    assert alg.get_scale_factor(got, 10) == pytest.approx(np.linalg.norm(want) / 10, rel=1e-9)
    # apply_projector on an explicit Y
    Y = L.standard_A(lp.A, lp.sense) @ sp.diags(L.standard_x(lp.A, lp.b, lp.sense, xr))
    v = L.standard_x(lp.A, lp.b, lp.sense, xr) * L.standard_c(lp.c, lp.sense)
    ref, _ = L.projector_explicit(Y, v)
    np.testing.assert_allclose(alg.apply_projector(Y, v), ref, rtol=0, atol=1e-6 * np.linalg.norm(ref))
    # projection of c onto null(A_std): A_std * proj == 0
    pc = alg.get_projector_c(lp)
    assert np.linalg.norm(L.standard_A(lp.A, lp.sense) @ pc) < 1e-6 * np.linalg.norm(lp.c)
    (flag, _) = quiet(alg.check_feasibility_problem, lp)
    assert flag is False
    # a cost in the row space of A_std is a feasibility problem
    from smart_crossover.formats import GeneralLP
    eq = GeneralLP(lp.A, lp.b, lp.A.T @ np.arange(lp.b.size, dtype=float), lp.l, lp.u, np.full(lp.b.size, "="))
    (flag, text) = quiet(alg.check_feasibility_problem, eq)
    assert flag is True and "feasibility problem" in text


def test_global_rng_side_effect_is_reproduced(g1):
    """Quirk Q5: perturb_c re-seeds numpy's global generator with 42 and draws n numbers."""
    from smart_crossover.lp_methods import algorithms as alg
    lp = make_lp(g1)
    n = lp.c.size
    np.random.seed(42)
    np.random.uniform(0.9, 1, n)
    want_next = np.random.random(3)
    for _ in range(2):                       # second call hits the direction cache
        np.random.seed(7)
        alg.perturb_c(lp, g1["x"], True)
        assert np.array_equal(np.random.random(3), want_next)


def certificates(lp, out, ref_obj):
    """Solver-independent checks of a returned vertex."""
    x, y, basis = out.x, out.y, out.basis
    assert abs(float(lp.c @ x) - ref_obj) <= 1e-9 * (1 + abs(ref_obj))
    s_p = lp.b - lp.A @ x
    lt = lp.sense == "<"
    assert np.all(np.abs(s_p[~lt]) <= 1e-6) and np.all(s_p[lt] >= -1e-6)
    assert np.all(x >= lp.l - 1e-6) and np.all(x <= lp.u + 1e-6)
    rc = lp.c - lp.A.T @ y
    assert np.all(rc[basis.vbasis == -1] >= -1e-6) and np.all(rc[basis.vbasis == -2] <= 1e-6)
    assert np.all(np.abs(rc[basis.vbasis == 0]) <= 1e-6)
    assert int(np.count_nonzero(basis.vbasis == 0) + np.count_nonzero(basis.cbasis == 0)) == lp.b.size


@pytest.mark.parametrize("case", ["c1", "medium"])
def test_run_perturb_algorithm_end_to_end(case):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods.algorithms import run_perturb_algorithm
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.config1() if case == "c1" else workloads.sparse_lp(400, 1600, 5, seed=77, stratified=False)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    ref = solve_lp(lp, "HGS", "default", SolverSettings(log_console=0))
    assert ref.status == "OPTIMAL"
    out, text = quiet(run_perturb_algorithm, lp, "HGS", 1e-10, 1e-6)     # HiGHS logs to the C stdout, like Gurobi
    assert out.status == "OPTIMAL"
    assert "*** Running the perturbation crossover algorithm... ***" in text
    assert "Primal-dual gap" in text
    if "*** A primal optimal BFS is found. ***" in text:
        # quirk Q2: the early return is the sub-problem's output -- lift it before checking
        assert out.x.size <= lp.c.size
    else:
        certificates(lp, out, ref.obj_val)


# ----------------------------------------------------------------------------- free variables
def lp_with_free_columns(m=40, n=120, nf=6, seed=3):
    from smart_crossover.formats import GeneralLP
    inst = workloads.sparse_lp(m, n, 6, seed=seed, frac_lt=0.5, frac_upper=0.3, stratified=False)
    rng = np.random.default_rng(seed + 100)
    free = np.sort(rng.choice(n, nf, replace=False))
    l, u = inst.l.copy(), inst.u.copy()
    l[free], u[free] = -np.inf, np.inf
    x = inst.x.copy()
    x[free] = rng.standard_normal(nf)                                   # free columns may sit anywhere
    lp = GeneralLP(inst.A, inst.b, inst.c, l, u, inst.sense)
    return lp, x, inst.y, free


@pytest.mark.parametrize("seed", [3, 4])
def test_projector_with_free_variables_matches_the_exact_qp(seed):
    """Free-variable branch (lp_methods/algorithms.py:173-180): the reference's QP runs inside Gurobi
    (parity unpinned); the device result must match the exact minimiser of that QP."""
    from smart_crossover.lp_methods import algorithms as alg
    lp, x, _, free = lp_with_free_columns(seed=seed)
    xr = L.x_perturb_val(x, lp.l, lp.u)
    want = L.projector_Xc_free(lp.A, lp.b, lp.c, lp.l, lp.u, lp.sense, xr)
    got = alg.get_projector_Xc(lp, xr)
    assert got.shape == want.shape == (lp.c.size - free.size + int(np.count_nonzero(lp.sense == "<")),)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.linalg.norm(want))
    assert np.linalg.norm(got) == pytest.approx(np.linalg.norm(want), rel=1e-5)
    # perturb_c end to end: same scale factor -> same perturbed cost; free columns are not perturbed
    want_c, info = L.perturbed_cost_full(lp.A, lp.b, lp.c, lp.l, lp.u, lp.sense, x, False)
    got_c = alg.perturb_c(lp, x, False)
    # compare the perturbations themselves (c + p can cancel): they scale with the projection norm
    np.testing.assert_allclose(got_c - lp.c, want_c - lp.c, rtol=2e-5, atol=0)
    assert np.array_equal(got_c[free], lp.c[free]) and info["sf"] > 0
    # and the whole sub-problem construction runs with free columns present
    mgr, _ = quiet(alg.get_perturb_problem, lp, x, np.zeros(lp.b.size), 1e-3, 1e-3, False)
    assert mgr.lp_sub.c.size <= lp.c.size


def test_apply_projector_qp_with_free_block():
    from smart_crossover.lp_methods import algorithms as alg
    rng = np.random.default_rng(12)
    A = sp.random(25, 70, density=0.2, random_state=5, format="csr")
    A.data = rng.uniform(-1, 1, A.nnz)
    A_f = sp.random(25, 4, density=0.5, random_state=6, format="csr")
    A_f.data = rng.uniform(-2, 2, A_f.nnz)
    v = rng.standard_normal(70)
    want = L.projector_qp_exact(A, v, A_f)
    got = alg.apply_projector_qp(A, v, A_f)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.linalg.norm(want))
    # the defining property: A x lies in the range of A_f, and x is no farther from v than v's own projection
    resid = A @ got
    coef = np.linalg.lstsq(A_f.toarray(), -resid, rcond=None)[0]
    assert np.linalg.norm(resid + A_f @ coef) < 1e-5 * np.linalg.norm(v)
    plain = alg.apply_projector_qp(A, v)
    assert np.linalg.norm(got - v) <= np.linalg.norm(plain - v) * (1 + 1e-9)
