"""GPU: control flow of ``run_perturb_algorithm`` against the reference (golden G6).

tests/golden/make_golden.py ran the reference's function (lp_methods/algorithms.py:18-76) with the name
``solve_lp`` of that module replaced by a canned backend -- a fixed function of its arguments, no
optimisation -- scripted to walk every branch: early return, gap failure -> warm primal simplex, and the
gamma retry after INFEASIBLE / UNBOUNDED re-solves (:56-59).  Here the build's function is driven by the same
canned backend and must make the same calls with the same arguments, print the same lines, pass the same
gamma sequence and index-set sizes, and return the same object."""
import datetime
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import GOLDEN
import workloads

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(GOLDEN, "g6_control_flow.json")))


def instance(name):
    if name.startswith("afiro"):
        return workloads.config1()
    return workloads.sparse_lp(300, 1500, 6, seed=12, stratified=False, frac_upper=0.3)


def canned_backend(inst, script, barrier_obj, log):
    """Same function as canned_backend of tests/golden/make_golden.py (see its docstring)."""
    from smart_crossover.output import Basis, Output
    state = {"resolves": 0}

    def solve_lp(lp, solver="GRB", method="default", settings=None, warm_start_basis=None, warm_start_solution=None):
        rec = dict(method=method, n=int(lp.c.size), m=int(lp.b.size), solver=solver,
                   presolve=settings.presolve, crossover=settings.crossover, barrierTol=settings.barrierTol,
                   optimalityTol=settings.optimalityTol, log_file=settings.log_file,
                   has_ws_solution=warm_start_solution is not None, has_ws_basis=warm_start_basis is not None)
        if len(log) == 0:
            out = Output(x=inst.x.copy(), y=inst.y.copy(), obj_val=barrier_obj, status="OPTIMAL",
                         runtime=datetime.timedelta(0), iter_count=0, bar_iter_count=7)
        elif method == "barrier":
            status = script[state["resolves"]]
            state["resolves"] += 1
            xs, ys = warm_start_solution
            rec["c_sub"] = np.asarray(lp.c).tolist()
            rec["n_eq_rows"] = int(np.count_nonzero(np.asarray(lp.sense) == "="))
            out = Output(x=np.asarray(xs).copy(), y=np.asarray(ys).copy(), obj_val=float(lp.c @ xs), status=status,
                         runtime=datetime.timedelta(0), iter_count=0,
                         basis=Basis(np.where(np.asarray(xs) > 1e-3, 0, -1), np.full(lp.b.size, -1)))
        else:
            xs, ys = warm_start_solution
            rec["ws_x"] = np.asarray(xs).tolist()
            rec["ws_vbasis"] = np.asarray(warm_start_basis.vbasis).astype(int).tolist()
            rec["ws_cbasis"] = np.asarray(warm_start_basis.cbasis).astype(int).tolist()
            out = Output(x=np.asarray(xs).copy(), y=np.asarray(ys).copy(), obj_val=float(lp.c @ xs), status="OPTIMAL",
                         runtime=datetime.timedelta(0), iter_count=11,
                         basis=Basis(np.asarray(warm_start_basis.vbasis), np.asarray(warm_start_basis.cbasis)))
        rec["returned_status"] = out.status
        log.append((rec, out))
        return out

    return solve_lp


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_run_perturb_algorithm_walks_the_reference_branches(case, monkeypatch):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    inst = instance(case["name"])
    lp = GeneralLP(inst.A.copy(), inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    log, gammas = [], []
    real_gpp = alg.get_perturb_problem

    def rec_gpp(lp_, x, y, gamma, gamma_dual, is_feas):
        m_ = real_gpp(lp_, x, y, gamma, gamma_dual, is_feas)
        gammas.append(dict(gamma=gamma, gamma_dual=gamma_dual, is_feas=bool(is_feas),
                           n_fix_low=int(m_.var_info["fix_low"].size), n_fix_up=int(m_.var_info["fix_up"].size),
                           n_fixed_rows=int(m_.fixed_constraints.size)))
        return m_

    monkeypatch.setattr(alg, "solve_lp", canned_backend(inst, case["script"], case["barrier_obj"], log))
    monkeypatch.setattr(alg, "get_perturb_problem", rec_gpp)
    buf = io.StringIO()
    with redirect_stdout(buf):
        result = alg.run_perturb_algorithm(lp, solver="CANNED", barrierTol=1e-7, optimalityTol=1e-5, log_file="")

    # the gamma / gamma_dual sequence (exact doubles) and the sizes of the index sets at every attempt
    assert gammas == case["gammas"]
    # the same solver calls with the same arguments
    assert len(log) == len(case["calls"])
    for (rec, _), want in zip(log, case["calls"]):
        for key, val in want.items():
            if key == "c_sub":        # perturbed cost: the scale factor comes out of a CG run
                got = np.asarray(rec[key])
                np.testing.assert_allclose(got, np.asarray(val), rtol=1e-5, atol=1e-12)
            elif key == "ws_x":
                assert np.array_equal(np.asarray(rec[key]), np.asarray(val)), key
            else:
                assert rec[key] == val, (key, rec[key], val)
    # printed lines (the reference's log scrapers key on them)
    assert buf.getvalue().splitlines() == case["printed"]
    # which call's Output object is handed back (quirk Q2: the sub-problem's on an early return)
    returned_by = [i for i, (_, o) in enumerate(log) if o is result]
    assert returned_by == [case["returned_by_call"]]
    assert result.status == case["result_status"] and np.asarray(result.x).size == case["result_x_len"]
