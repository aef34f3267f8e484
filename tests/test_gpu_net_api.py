"""GPU: the drop-in Python API of the network crossover (smart_crossover.network_methods.*)
against the reference goldens and solver-independent certificates ('HGS' backend for the re-solves)."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal, csr_from, same_csr
from oracle import net_path as N
import workloads

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def test_mcf_manager_matches_reference_golden(g3):
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.net_manager import MCFManagerStd
    from smart_crossover.output import Basis
    mcf = MinCostFlow(A=csr_from(g3, "A"), b=g3["b"].copy(), c=g3["c"].copy(), u=g3["u"].copy())
    mgr = MCFManagerStd(mcf)
    queue, ind = mgr.get_sorted_flows(g3["x"])
    assert bits_equal(ind, g3["ind"])
    assert queue.dtype == np.int64 and N.same_up_to_ties(ind, queue, g3["queue_ref"])
    mgr.rescale_cost(np.max(np.abs(mcf.c)))
    assert bits_equal(mcf.c, g3["c_scaled"])                       # the caller's object is rebound (quirk Q7)
    mgr.fix_variables(ind_fix_to_up=np.where(g3["x"] >= mcf.u / 2)[0], ind_fix_to_low=np.where(g3["x"] < mcf.u / 2)[0])
    bigM = mgr.m * np.max(mcf.c)
    assert bigM == float(g3["bigM"])
    mgr.extend_by_bigM(bigM)
    assert same_csr(mgr.mcf.A, csr_from(g3, "A1"))
    assert bits_equal(mgr.mcf.b, g3["b1"]) and bits_equal(mgr.mcf.c, g3["c1"]) and bits_equal(mgr.mcf.u, g3["u1"])
    assert np.array_equal(mgr.artificial_vars, g3["artificial"])
    mgr.update_subproblem()
    mgr.set_initial_basis()
    assert np.array_equal(mgr.basis.vbasis, g3["vb0"]) and np.array_equal(mgr.basis.cbasis, g3["cb0"])
    assert same_csr(mgr.mcf_sub.A, csr_from(g3, "Asub0")) and bits_equal(mgr.mcf_sub.b, g3["bsub0"])
    lft, rgt = (int(v) for v in g3["batch0"])
    mgr.add_free_variables(g3["queue_ref"][lft:rgt])               # the reference's own tie order for this check
    mgr.update_subproblem()
    assert np.array_equal(mgr.var_info["non_fix"], g3["non_fix1"])
    assert np.array_equal(mgr.var_info["fix_up"], g3["fix_up1"]) and np.array_equal(mgr.var_info["fix_low"], g3["fix_low1"])
    ref_sub = csr_from(g3, "Asub1")
    assert mgr.mcf_sub.A.shape == ref_sub.shape and (mgr.mcf_sub.A != ref_sub).nnz == 0
    assert np.array_equal(mgr.mcf_sub.A.indptr, ref_sub.indptr)
    assert bits_equal(mgr.mcf_sub.b, g3["bsub1"]) and bits_equal(mgr.mcf_sub.c, g3["csub1"]) and bits_equal(mgr.mcf_sub.u, g3["usub1"])
    mgr.set_basis(Basis(g3["vb0"], g3["cb0"]))
    assert bits_equal(mgr.get_reduced_cost_for_original_mcf(g3["y"]), g3["rc"])
    assert mgr.check_optimality_condition(np.zeros(mgr.mcf.c.size), g3["y"]) == bool(g3["opt_flag"])
    assert bits_equal(mgr.recover_x_from_sub_x(g3["x_sub"]), g3["recover_x"])
    rb = mgr.recover_basis_from_sub_basis(Basis(g3["vb_sub"], g3["cb_sub"]))
    assert np.array_equal(rb.vbasis, g3["recover_vb"])


def test_ot_manager_matches_reference_golden(g4):
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.net_manager import OTManager
    from smart_crossover.network_methods import tree_BI
    S, D = g4["M"].shape
    ot = OptTransport(g4["s"].copy(), g4["d"].copy(), g4["M"].copy())
    mcf = ot.to_MCF()
    ref_A = csr_from(g4, "Amcf")
    assert mcf.A.shape == ref_A.shape and (mcf.A != ref_A).nnz == 0
    assert bits_equal(mcf.b, g4["b_mcf"]) and bits_equal(mcf.c, g4["c_mcf"]) and np.all(np.isinf(mcf.u))
    mgr = OTManager(ot)
    queue, ind = mgr.get_sorted_flows(g4["x"])
    assert bits_equal(ind, g4["ind"]) and N.same_up_to_ties(ind, queue, g4["queue_ref"])
    # TNET tree basis
    mgr.get_mcf()
    assert np.array_equal(tree_BI.max_weight_spanning_tree(ot, ind), g4["tree_edges"])
    basis, pushes = tree_BI.tree_basis_identify(mgr, ind)
    assert np.array_equal(basis.vbasis, g4["tree_vb"].astype(int)) and np.array_equal(basis.cbasis, g4["tree_cb"].astype(int))
    assert pushes == int(g4["push_iter"])
    # CNET_OT set-up
    mgr2 = OTManager(OptTransport(g4["s"].copy(), g4["d"].copy(), g4["M"].copy()))
    mgr2.extend_by_bigM(mgr2.m * np.max(g4["M"]))
    mgr2.get_mcf()
    mgr2.set_initial_basis()
    assert bits_equal(mgr2.ot.s, g4["s1"]) and bits_equal(mgr2.ot.d, g4["d1"]) and bits_equal(mgr2.ot.M, g4["M1"])
    assert np.array_equal(mgr2.mask_sub_ot, g4["mask1"]) and np.array_equal(mgr2.artificial_vars, g4["artificial"])
    assert np.array_equal(mgr2.basis.vbasis, g4["vb0"].astype(int)) and np.array_equal(mgr2.basis.cbasis, g4["cb0"].astype(int))
    lft, rgt = (int(v) for v in g4["batch0"])
    mgr2.add_free_variables(g4["queue_ref"][lft:rgt])
    assert np.array_equal(mgr2.mask_sub_ot, g4["mask2"])
    sub = mgr2.get_sub_problem()
    ref_sub = csr_from(g4, "Asub")
    assert sub.A.shape == ref_sub.shape and (sub.A != ref_sub).nnz == 0
    assert bits_equal(sub.c, g4["csub"]) and bits_equal(sub.u, g4["usub"])
    assert bits_equal(mgr2.get_reduced_cost_for_original_OT(g4["y"]), g4["rc"])
    xs = np.zeros(mgr2.ot.s.size * mgr2.ot.d.size)
    assert mgr2.check_optimality_condition(xs, g4["y"]) == bool(g4["opt_flag"])
    assert mgr2.check_optimality_condition(xs, np.zeros_like(g4["y"])) == bool(g4["opt_flag_zero_y"])


def _solve_reference_objective(problem):
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_mcf
    out = solve_mcf(problem, "HGS", "default", SolverSettings(log_console=0))
    assert out.status == "OPTIMAL"
    return out.obj_val


def test_network_crossover_cnet_mcf_end_to_end():
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.mcf(200, 1600, seed=21)
    # make the inexact flow the perturbed optimum so that the ranking is informative
    base = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
    want = _solve_reference_objective(base)
    mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
    c_before = mcf.c.copy()
    out, text = quiet(network_crossover, inst.x, mcf=mcf, method="cnet_mcf", solver="HGS",
                      solver_settings=SolverSettings(log_console=0))
    assert "*** Running cnet_mcf algorithm. ***" in text and "CG iteration 1 completed" in text
    assert out.obj_val == pytest.approx(want, rel=1e-9)
    assert not np.array_equal(mcf.c, c_before)                     # quirk Q7: the caller's cost was rescaled
    V, E = inst.A.shape
    x = out.x[:E]
    assert np.all(out.x[E:] < 1e-8)                                # no flow left on artificial arcs
    assert np.allclose(inst.A @ x, inst.b, atol=1e-6) and np.all(x >= -1e-9) and np.all(x <= inst.u + 1e-9)
    assert c_before @ x == pytest.approx(want, rel=1e-9)
    assert int(np.count_nonzero(out.basis.vbasis == 0) + np.count_nonzero(out.basis.cbasis == 0)) == V + 1


@pytest.mark.parametrize("method", ["tnet", "cnet_ot"])
def test_network_crossover_ot_end_to_end(method):
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.ot(20, 25, seed=11)
    ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
    want = _solve_reference_objective(ot.to_MCF())
    out, text = quiet(network_crossover, inst.x, ot=ot, method=method, solver="HGS",
                      solver_settings=SolverSettings(log_console=0))
    assert f"*** Running {method} algorithm. ***" in text
    assert out.obj_val == pytest.approx(want, rel=1e-9)
    S, D = inst.M.shape
    X = (out.x.reshape(S + 1, D + 1)[:S, :D] if method == "cnet_ot" else out.x.reshape(S, D))
    assert np.allclose(X.sum(axis=1), inst.s, atol=1e-7) and np.allclose(X.sum(axis=0), inst.d, atol=1e-7)
    assert float((X * inst.M).sum()) == pytest.approx(want, rel=1e-9)


def test_network_crossover_rejects_unknown_method():
    from smart_crossover.network_methods.algorithms import network_crossover
    with pytest.raises(ValueError):
        quiet(network_crossover, np.zeros(4), method="simplex", solver="HGS")


def test_tree_basis_identify_rejects_disconnected_support():
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.net_manager import OTManager
    from smart_crossover.network_methods import tree_BI
    ot = OptTransport(np.array([0.5, 0.5]), np.array([0.5, 0.5]), np.ones((2, 2)))
    mgr = OTManager(ot)
    mgr.get_mcf()
    with pytest.raises(ValueError):
        tree_BI.tree_basis_identify(mgr, np.array([1.0, 0.0, 0.0, 1.0]))
