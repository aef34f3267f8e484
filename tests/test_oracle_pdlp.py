"""CPU: oracle/pdlp.py (the statement of the first-order stage K16p) reaches HiGHS' optimal value and a
primal-dual pair that satisfies the LP's optimality conditions.  The reference runs Gurobi's barrier at this
point (lp_methods/algorithms.py:50-54): parity unpinned, HiGHS stands in as the independent solver."""
import numpy as np
import pytest
from scipy.optimize import linprog

from oracle import pdlp as P
import workloads


def highs(inst):
    lt = inst.sense == "<"
    r = linprog(inst.c, A_ub=inst.A[lt], b_ub=inst.b[lt], A_eq=inst.A[~lt], b_eq=inst.b[~lt],
                bounds=list(zip(inst.l, [None if np.isinf(v) else v for v in inst.u])), method="highs")
    assert r.status == 0
    return r


@pytest.mark.parametrize("m,n,k,seed", [(27, 51, 2, 2024), (60, 200, 4, 1), (150, 400, 5, 2)])
def test_oracle_pdlp_reaches_the_optimum(m, n, k, seed):
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=False)
    # the crossover's situation: an interior point of the LP, a perturbed cost (lp_methods/algorithms.py:148-151)
    inst.c = inst.c + 1e-2 * np.random.default_rng(seed).uniform(0.9, 1.0, n) / np.maximum(inst.x, 1e-6).clip(1e-2)
    out = P.pdlp(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense == "<", inst.x, inst.y, max_iter=200000, tol=1e-9)
    ref = highs(inst)
    assert out["status"] == 0
    assert out["primal_obj"] == pytest.approx(ref.fun, rel=1e-6, abs=1e-6)
    x, y = out["x"], out["y"]
    lt = inst.sense == "<"
    r = inst.b - inst.A @ x
    assert np.abs(r[~lt]).max(initial=0) < 1e-6 and r[lt].min(initial=0) > -1e-6
    assert np.all(x >= inst.l) and np.all(x <= inst.u) and np.all(y[lt] <= 0)


def test_oracle_pdlp_iteration_limit_and_cold_start():
    inst = workloads.sparse_lp(40, 120, 3, seed=5, stratified=False)
    out = P.pdlp(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense == "<", None, None, max_iter=100, tol=1e-12)
    assert out["status"] == 3 and out["iters"] == 128          # rounded up to whole periods of 64
    dr, dc = P.scalings(inst.A)
    assert P.operator_norm(inst.A, dr, dc) <= 1.0 + 1e-12       # Pock-Chambolle (alpha = 1) bounds the norm by 1
