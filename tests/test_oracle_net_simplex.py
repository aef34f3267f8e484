"""CPU: the oracle's dual network simplex (oracle/net_simplex.py, the statement csrc/sx_netdual.hip is tested
against pivot for pivot) against an independent solver -- scipy's HiGHS -- and its own invariants.  Reference:
network_methods/net_manager.py:211-222 (solve_subproblem -> solve_mcf with a warm basis); the reference's solver
is Gurobi / CPLEX, absent here, so the optimum is pinned by HiGHS and by certificates."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from oracle.net_simplex import dual_network_simplex
from test_gpu_netsimplex import big_m_network, certificates


def highs(A, b, c, u):
    ref = linprog(c, A_eq=A, b_eq=b, bounds=list(zip(np.zeros(c.size), [None if np.isinf(v) else v for v in u])), method="highs")
    assert ref.status == 0
    return ref.fun


@pytest.mark.parametrize("V,E,seed", [(2, 1, 0), (12, 40, 0), (60, 400, 1), (150, 1500, 4)])
@pytest.mark.parametrize("steepest,bfrt", [(True, True), (False, True), (True, False)])
def test_optimum_and_certificates_from_the_artificial_star(V, E, seed, steepest, bfrt):
    A, b, c, u, tail, head, vb, cb = big_m_network(V, E, seed, inf_frac=0.0)
    out = dual_network_simplex(tail, head, c, u, b, vb, root=V, steepest=steepest, bfrt=bfrt)
    assert out["status"] == 0
    assert out["obj"] == pytest.approx(highs(A, b, c, u), rel=1e-9, abs=1e-9)
    cbo = np.where(np.arange(V + 1) == V, 0, -1)
    certificates(A, b, c, u, tail, head, out["x"], out["y"], out["vbasis"].astype(int), cbo)
    # a fixed point: no iterations, no flips from its own optimal basis
    again = dual_network_simplex(tail, head, c, u, b, out["vbasis"], root=V, steepest=steepest, bfrt=bfrt)
    assert again["status"] == 0 and again["iters"] == 0 and again["flips"] == 0
    assert np.array_equal(again["x"], out["x"])


def test_the_rules_cut_the_iterations():
    """Why the device uses both rules (profiles/r02/netdual.md): fewer iterations with the bound-flipping ratio test
    and with violation^2 / |subtree| than without either."""
    A, b, c, u, tail, head, vb, cb = big_m_network(200, 3000, 8, inf_frac=0.0)
    both = dual_network_simplex(tail, head, c, u, b, vb, root=200)
    no_flip = dual_network_simplex(tail, head, c, u, b, vb, root=200, bfrt=False)
    no_steep = dual_network_simplex(tail, head, c, u, b, vb, root=200, steepest=False)
    assert both["status"] == no_flip["status"] == no_steep["status"] == 0
    assert both["obj"] == pytest.approx(no_flip["obj"], rel=1e-12) and both["obj"] == pytest.approx(no_steep["obj"], rel=1e-12)
    assert both["iters"] < no_flip["iters"] and both["iters"] < no_steep["iters"]


def test_domain_and_infeasibility():
    A, b, c, u, tail, head, vb, cb = big_m_network(40, 300, 5, inf_frac=1.0)
    assert dual_network_simplex(tail, head, c, u, b, vb, root=40)["status"] == 5      # uncapacitated arcs at the wrong bound
    vb_bad = vb.copy()
    vb_bad[0] = 0
    A2, b2, c2, u2, tail2, head2, vb2, cb2 = big_m_network(40, 300, 5, inf_frac=0.0)
    vb2[0] = 0
    assert dual_network_simplex(tail2, head2, c2, u2, b2, vb2, root=40)["status"] == 5  # V arcs coded basic: no tree
    tail3, head3 = np.array([0, 1]), np.array([1, 2])
    b3, c3, u3 = np.array([5.0, 0.0, -5.0]), np.array([1.0, 1.0]), np.array([1.0, 9.0])
    assert dual_network_simplex(tail3, head3, c3, u3, b3, np.array([0, 0], dtype=np.int8), root=2)["status"] == 1
