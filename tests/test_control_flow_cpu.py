"""CPU: control flow of the build's ``run_perturb_algorithm`` against golden G6 (the reference's own run,
lp_methods/algorithms.py:18-76) without a GPU.  The two device steps it calls -- ``get_perturb_problem`` and
``check_feasibility_problem`` -- are replaced *in this test only* by stand-ins built on the CPU oracle, so
what is exercised is the host logic: the gamma retry (:56-59), the arguments of every solver call, the gap
test, quirk Q2 (sub-problem Output on the early return), the warm start handed to the final simplex.  The
device versions of the same five scripts run in tests/test_gpu_control_flow.py."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import lp_path as L
from test_gpu_control_flow import CASES, canned_backend, instance


def oracle_get_perturb_problem(lp, x, y, gamma, gamma_dual, is_feas):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods.lp_manager import LPManager
    res = L.scoring_pass(lp.A, lp.b, lp.c, lp.l, lp.u, x, y, gamma, gamma_dual)
    c_pt, _ = L.perturbed_cost_full(lp.A, lp.b, lp.c, lp.l, lp.u, lp.sense, x, is_feas, explicit=True)
    mgr = LPManager(GeneralLP(lp.A, lp.b.copy(), c_pt, lp.l.copy(), lp.u.copy(), lp.sense.copy()))
    mgr.fix_variables(res["fix_low"], res["fix_up"])
    mgr.fix_constraints(res["fixed_rows"])
    print("  The number of fixed variables is %d." % mgr.get_num_fixed_variables())
    print("  The number of fixed constraints is %d." % mgr.get_num_fixed_constraints())
    sub = L.sub_problem(lp.A, lp.b, c_pt, lp.l, lp.u, lp.sense, res["fix_low"], res["fix_up"], res["fixed_rows"])
    mgr.lp_sub = GeneralLP(sub["A"], sub["b"], sub["c"], sub["l"], sub["u"], sub["sense"])
    return mgr


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_host_control_flow_matches_reference(case, monkeypatch):
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    inst = instance(case["name"])
    lp = GeneralLP(inst.A.copy(), inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
    log, gammas = [], []

    def rec_gpp(lp_, x, y, gamma, gamma_dual, is_feas):
        m_ = oracle_get_perturb_problem(lp_, x, y, gamma, gamma_dual, is_feas)
        gammas.append(dict(gamma=gamma, gamma_dual=gamma_dual, is_feas=bool(is_feas),
                           n_fix_low=int(m_.var_info["fix_low"].size), n_fix_up=int(m_.var_info["fix_up"].size),
                           n_fixed_rows=int(m_.fixed_constraints.size)))
        return m_

    monkeypatch.setattr(alg, "solve_lp", canned_backend(inst, case["script"], case["barrier_obj"], log))
    monkeypatch.setattr(alg, "get_perturb_problem", rec_gpp)
    monkeypatch.setattr(alg, "check_feasibility_problem", lambda lp_: False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        result = alg.run_perturb_algorithm(lp, solver="CANNED", barrierTol=1e-7, optimalityTol=1e-5, log_file="")
    assert gammas == case["gammas"]
    assert len(log) == len(case["calls"])
    for (rec, _), want in zip(log, case["calls"]):
        for key, val in want.items():
            if key == "c_sub":
                np.testing.assert_allclose(np.asarray(rec[key]), np.asarray(val), rtol=1e-9, atol=1e-14)
            elif key == "ws_x":
                assert np.array_equal(np.asarray(rec[key]), np.asarray(val))
            else:
                assert rec[key] == val, (key, rec[key], val)
    assert buf.getvalue().splitlines() == case["printed"]
    assert [i for i, (_, o) in enumerate(log) if o is result] == [case["returned_by_call"]]
    assert result.status == case["result_status"] and np.asarray(result.x).size == case["result_x_len"]


def test_retry_branch_is_covered_by_the_goldens():
    """The golden set itself: at least one case retries once, one twice, one returns early after a retry."""
    retries = sorted(len(c["gammas"]) for c in CASES)
    assert retries[0] == 1 and retries[-1] == 3
    assert any(c["early"] and len(c["gammas"]) > 1 for c in CASES)
    assert any(not c["early"] and len(c["gammas"]) > 1 for c in CASES)
    for c in CASES:
        g = [1e-3]
        gd = [1e-3]
        for _ in c["gammas"][1:]:
            g.append(g[-1] * 1e-5)
            gd.append(gd[-1] * 1e-5 ** 2)
        assert [q["gamma"] for q in c["gammas"]] == g and [q["gamma_dual"] for q in c["gammas"]] == gd
