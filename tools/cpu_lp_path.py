#!/usr/bin/env python3
"""The CPU path beside an LP crossover, by the same route the device takes: host arithmetic of get_perturb_problem by
the numpy/scipy oracle, the first-order stage by oracle/pdlp.py (numpy), then HiGHS' simplex (scipy's bundled
build) warm-started from the basis that point indicates -- what the reference's final step does with its solver
(lp_methods/algorithms.py:69-74).  HiGHS' interior point method and its
simplex from scratch do not finish on these sub-problems (profiles/r03/lp_1e6_highs.json); this is the CPU path that
might.  Development / measurement tool: bench.py reads the record it writes.

    python tools/cpu_lp_path.py [which=c2 | m=100000 n=1000000] [iters=20000] [limit=7200] [out=path.json]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402


def run(kw):
    """kw: which / m / n / iters / limit / tol / log / out as strings or numbers (see the module docstring); returns the
    record and writes it to kw['out'] when given."""
    kw = {k: str(v) for k, v in kw.items()}
    m, n = int(kw.get("m", 100_000)), int(kw.get("n", 1_000_000))
    iters, limit = int(kw.get("iters", 20_000)), float(kw.get("limit", 7200))
    from oracle import lp_path as L
    from oracle import pdlp as P
    import scipy.optimize._highspy._core as hc
    inst = workloads.config2() if kw.get("which") == "c2" else workloads.netlib_lp(m, n)
    m, n = inst.A.shape
    t0 = time.perf_counter()
    res = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    c_pt, _ = L.perturbed_cost_full(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, inst.x, False, explicit=False)
    sub = L.sub_problem(inst.A, inst.b, c_pt, inst.l, inst.u, inst.sense, res["fix_low"], res["fix_up"], res["fixed_rows"])
    t1 = time.perf_counter()
    A = sp.csr_matrix(sub["A"])
    ms, ns = A.shape
    lt = np.asarray(sub["sense"]) == "<"
    b, c, l, u = (np.asarray(sub[k], dtype=np.float64) for k in ("b", "c", "l", "u"))
    keep = np.ones(n, dtype=bool)
    keep[res["fix_low"]] = False
    keep[res["fix_up"]] = False
    x0 = inst.x[keep]
    rows = np.ones(m, dtype=bool)
    rows[res["fixed_rows"]] = False
    y0 = inst.y[:m][rows] if len(inst.y) >= m else np.zeros(ms)
    if x0.size != ns or y0.size != ms:
        x0, y0 = None, None
    print(f"sub-problem {A.shape}, {A.nnz} entries; host arithmetic {t1 - t0:.1f} s", flush=True)
    r = P.pdlp(A, b, c, l, u, lt, x0, y0, max_iter=iters)
    t2 = time.perf_counter()
    print(f"first-order stage: {r['iters']} iterations, {r['restarts']} restarts, {t2 - t1:.1f} s, pr {r['primal_residual']:.2e} "
          f"du {r['dual_residual']:.2e} gap {r['gap']:.2e}", flush=True)
    # HiGHS crossover from (x, y)
    h = hc._Highs()
    h.setOptionValue("output_flag", bool(int(kw.get("log", 0))))
    h.setOptionValue("time_limit", limit)
    lp = hc.HighsLp()
    Ac = A.tocsc()
    lp.num_col_, lp.num_row_ = ns, ms
    lp.col_cost_, lp.col_lower_, lp.col_upper_ = c, l, u
    inf = h.getInfinity()
    lp.row_lower_ = np.where(lt, -inf, b)
    lp.row_upper_ = b
    lp.a_matrix_.format_ = hc.MatrixFormat.kColwise
    lp.a_matrix_.start_ = Ac.indptr.astype(np.int32)
    lp.a_matrix_.index_ = Ac.indices.astype(np.int32)
    lp.a_matrix_.value_ = Ac.data
    h.passModel(lp)
    # A starting basis from the first-order point, the way a crossover reads it: the variables strictly inside
    # their bounds and the '<' rows with slack are basic, by margin, exactly ms of them (the rows left over are covered
    # by their logicals); HiGHS repairs a singular guess by itself and runs its simplex from there.
    # (Highs::crossover from the same point -- IPX "crossover from starting point" -- ends in a segmentation fault
    # inside scipy 1.15.3's HiGHS 1.8.0, at 2,000 rows already.)
    xs = np.minimum(np.maximum(r["x"], l), u)
    span = np.where(np.isfinite(u - l), u - l, 1.0)
    marg_c = np.minimum(xs - l, np.where(np.isfinite(u), u - xs, np.inf)) / np.maximum(span, 1e-300)
    slack = b - A @ xs
    marg_r = np.where(lt, slack / (1.0 + np.abs(b)), 0.0)
    marg = np.concatenate([marg_c, marg_r])
    tol = float(kw.get("tol", 1e-7))
    cand = np.flatnonzero(marg > tol)
    if cand.size > ms:
        cand = cand[np.argsort(-marg[cand], kind="stable")[:ms]]
    basic = np.zeros(ns + ms, dtype=bool)
    basic[cand] = True
    if cand.size < ms:   # cover rows by logicals: equality rows first in row order
        free_rows = np.flatnonzero(~basic[ns:])
        basic[ns + free_rows[:ms - cand.size]] = True
    S = hc.HighsBasisStatus
    hb = hc.HighsBasis()
    hb.col_status = [S.kBasic if basic[j] else (S.kUpper if (np.isfinite(u[j]) and u[j] - xs[j] < xs[j] - l[j]) else S.kLower)
                     for j in range(ns)]
    hb.row_status = [S.kBasic if basic[ns + i] else S.kUpper for i in range(ms)]
    t3 = time.perf_counter()
    st_b = h.setBasis(hb)
    h.setOptionValue("solver", "simplex")
    h.setOptionValue("presolve", "off")
    st = h.run()
    t4 = time.perf_counter()
    print(f"setBasis {st_b}; {int(basic[:ns].sum())} structurals + {int(basic[ns:].sum())} logicals basic", flush=True)
    ms_ = h.getModelStatus()
    info = h.getInfo()
    status = h.modelStatusToString(ms_)
    obj = float(info.objective_function_value)
    rec = {"cpu_kind": "port of the device's route: numpy/scipy oracle (matrix-free CG) + oracle/pdlp.py first-order stage + "
                       "HiGHS' simplex (scipy's bundled build) warm-started from the basis that point indicates",
           "sub_problem": f"{ms} x {ns}, {A.nnz} entries (workloads.{'config2()' if kw.get('which') == 'c2' else f'netlib_lp({m}, {n})'})",
           "cpu_host_arithmetic_s": t1 - t0, "cpu_first_order_s": t2 - t1, "cpu_first_order_iterations": int(r["iters"]),
           "cpu_crossover_s": t4 - t3, "cpu_resolve_s": t4 - t1, "cpu_total_s": t4 - t0,
           "cpu_resolve_status": "OPTIMAL" if status == "Optimal" else status, "highs_return": str(st),
           "cpu_resolve_iterations": int(info.simplex_iteration_count), 
           "cpu_objective": obj if status == "Optimal" else None, "cpu_cores": os.cpu_count(), "cpu_time_limit_s": limit}
    if "out" in kw:
        json.dump(rec, open(kw["out"], "w"), indent=1)
    return rec


if __name__ == "__main__":
    print(json.dumps(run(dict(a.split("=") for a in sys.argv[1:]))), flush=True)
