#!/usr/bin/env python3
"""End-to-end network crossover at BASELINE sizes through the kept Python API (host memory in, basis out).

    python tests/perf/crossover_bench.py [--case c3_tnet|c3_cnet|c4_cnet|all] [--solver HGS] [--time-limit 600]

Prints one JSON object per case: wall time of ``network_crossover`` (the reference's ``Output.runtime``
definition: host set-up and bookkeeping + solver-reported solve times), how much of it the sub-problem
solver took, how much the host-side arithmetic took (flow indicators, ranking, sub-problem assembly,
pricing -- the part this repository moves to the GPU), the same arithmetic timed with the numpy/scipy
oracle on one host core, and solver-independent certificates of the result.
"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402


def quiet(fn, *a, **kw):
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = fn(*a, **kw)
    return out, buf.getvalue()


class SolveClock:
    """Wraps NetworkManager.solve_subproblem to record what the solver itself reports."""

    def __init__(self, manager_cls):
        self.cls = manager_cls
        self.orig = manager_cls.solve_subproblem
        self.seconds, self.wall, self.calls, self.sizes = 0.0, 0.0, 0, []

    def __enter__(self):
        clock = self

        def wrapped(mgr, solver, settings):
            t0 = time.perf_counter()
            out = clock.orig(mgr, solver, settings)
            clock.wall += time.perf_counter() - t0
            clock.seconds += out.runtime.total_seconds()
            clock.calls += 1
            clock.sizes.append(int(out.x.size))
            return out

        self.cls.solve_subproblem = wrapped
        return self

    def __exit__(self, *exc):
        self.cls.solve_subproblem = self.orig


def oracle_seconds_ot(inst):
    """CPU time of the host arithmetic the OT path does once per crossover (oracle restatement)."""
    from oracle import net_path as N
    t0 = time.perf_counter()
    ind = N.ot_flow_indicators(inst.x, inst.s, inst.d)
    N.rank_desc(ind)
    S, D = inst.M.shape
    y = np.zeros(S + D)
    N.ot_reduced_cost(inst.M, y)
    return time.perf_counter() - t0


def oracle_seconds_mcf(inst):
    from oracle import net_path as N
    t0 = time.perf_counter()
    ind, _ = N.mcf_flow_indicators(inst.A, inst.x, inst.u)
    N.rank_desc(ind)
    N.mcf_reduced_cost(inst.A, inst.c, np.zeros(inst.A.shape[0]), np.full(inst.A.shape[1], -1))
    return time.perf_counter() - t0


def run_ot(method, solver, limit):
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.network_methods.net_manager import OTManager
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.config3()
    ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
    S, D = inst.M.shape
    st = SolverSettings(log_console=0, timeLimit=limit)
    with SolveClock(OTManager) as clk:
        t0 = time.perf_counter()
        out, text = quiet(network_crossover, inst.x, ot=ot, method=method, solver=solver, solver_settings=st)
        wall = time.perf_counter() - t0
    X = (out.x.reshape(S + 1, D + 1)[:S, :D] if method == "cnet_ot" else out.x.reshape(S, D))
    rec = {
        "case": f"c3_{method}", "problem": f"OT {S}x{D} (n = {S * D}), Manhattan grid cost", "solver": solver,
        "wall_ms": wall * 1e3, "runtime_reported_ms": out.runtime.total_seconds() * 1e3,
        "solver_ms": clk.wall * 1e3, "host_path_ms": (wall - clk.wall) * 1e3, "subproblem_solves": clk.calls,
        "subproblem_columns": clk.sizes, "simplex_iterations": int(out.iter_count),
        "cg_rounds": text.count("CG iteration"), "objective": float(out.obj_val),
        "marginal_violation": float(max(np.abs(X.sum(axis=1) - inst.s).max(), np.abs(X.sum(axis=0) - inst.d).max())),
        "plan_cost": float((X * inst.M).sum()), "nonzeros_in_plan": int(np.count_nonzero(X > 1e-12)),
        "oracle_host_arithmetic_ms_one_pass": oracle_seconds_ot(inst) * 1e3,
    }
    return rec


def run_mcf(solver, limit, V=None, E=None):
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.network_methods.net_manager import MCFManagerStd
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.config4() if V is None else workloads.mcf(V, E, seed=3)
    mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
    V, E = inst.A.shape
    st = SolverSettings(log_console=0, timeLimit=limit)
    with SolveClock(MCFManagerStd) as clk:
        t0 = time.perf_counter()
        out, text = quiet(network_crossover, inst.x, mcf=mcf, method="cnet_mcf", solver=solver, solver_settings=st)
        wall = time.perf_counter() - t0
    x = out.x[:E]
    rec = {
        "case": "c4_cnet_mcf" if V == 2 ** 17 else "mcf_cnet_mcf", "problem": f"MCF V = {V}, E = {E}", "solver": solver,
        "wall_ms": wall * 1e3, "runtime_reported_ms": out.runtime.total_seconds() * 1e3,
        "solver_ms": clk.wall * 1e3, "host_path_ms": (wall - clk.wall) * 1e3, "subproblem_solves": clk.calls,
        "subproblem_columns": clk.sizes, "simplex_iterations": int(out.iter_count),
        "cg_rounds": text.count("CG iteration"), "objective": float(out.obj_val),
        "flow_conservation_violation": float(np.abs(inst.A @ x - inst.b).max()),
        "bound_violation": float(max((-x).max(), (x - inst.u).max(), 0.0)),
        "artificial_flow": float(np.abs(out.x[E:]).max()) if out.x.size > E else 0.0,
        "cost": float(inst.c @ x),
        "oracle_host_arithmetic_ms_one_pass": oracle_seconds_mcf(inst) * 1e3,
    }
    return rec


def run_lp(solver, limit, m=2000, n=10000, k=5):
    """Perturbation crossover end to end (run_perturb_algorithm): the initial barrier solve is the input
    of the crossover and is reported separately; everything after it is the crossover."""
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    inst = workloads.config2() if m == 20_000 else workloads.sparse_lp(m, n, k, seed=3, stratified=True, frac_upper=0.3)
    m, n = inst.A.shape
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    calls = []
    orig = alg.solve_lp

    def timed_solve(problem, solver="GRB", method="default", **kw):
        t0 = time.perf_counter()
        out = orig(problem, solver=solver, method=method, **kw)
        calls.append((method, time.perf_counter() - t0, int(problem.c.size), int(out.iter_count or 0),
                      int(out.bar_iter_count or 0), out.status))
        return out
    alg.solve_lp = timed_solve
    try:
        t0 = time.perf_counter()
        out, text = quiet(alg.run_perturb_algorithm, lp, solver=solver)
        wall = time.perf_counter() - t0
    finally:
        alg.solve_lp = orig
    barrier0 = calls[0][1]
    solver_after = sum(c[1] for c in calls[1:])
    return {
        "case": "lp_perturb", "problem": f"synthetic LP {m} x {n}, {k} nnz/col", "solver": solver,
        "wall_ms": wall * 1e3, "initial_barrier_ms": barrier0 * 1e3, "crossover_ms": (wall - barrier0) * 1e3,
        "solver_ms_after_barrier": solver_after * 1e3, "host_path_ms": (wall - barrier0 - solver_after) * 1e3,
        "solves": [{"method": c[0], "ms": c[1] * 1e3, "columns": c[2], "simplex_iters": c[3], "barrier_iters": c[4],
                    "status": c[5]} for c in calls],
        "early_return_on_subproblem": "A primal optimal BFS is found" in text,
        "fixed_columns_line": next((ln for ln in text.splitlines() if "fixed" in ln.lower()), ""),
        "objective": float(out.obj_val),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all")
    ap.add_argument("--solver", default="HGS")
    ap.add_argument("--time-limit", type=int, default=600)
    ap.add_argument("--repeat", type=int, default=1, help="run every case this many times in the process; the first "
                    "run pays library load, context creation and first-touch allocations (reported as cold)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="sx_ctx_set_option on the shared device context, e.g. spx_pricing=1")
    args = ap.parse_args()
    if args.option:
        from smart_crossover.hip import default_context
        for item in args.option:
            key, value = item.split("=")
            default_context().set_option(key, int(value))
    cases = ["c3_tnet", "c3_cnet", "c4_cnet"] if args.case == "all" else [args.case]
    first = True
    for case in cases:
        for rep in range(args.repeat):
            t0 = time.time()
            if case == "c3_tnet":
                rec = run_ot("tnet", args.solver, args.time_limit)
            elif case == "c3_cnet":
                rec = run_ot("cnet_ot", args.solver, args.time_limit)
            elif case == "c4_cnet":
                rec = run_mcf(args.solver, args.time_limit)
            elif case == "mcf_4k":
                rec = run_mcf(args.solver, args.time_limit, V=4096, E=32768)
            elif case == "mcf_12k":
                rec = run_mcf(args.solver, args.time_limit, V=12288, E=98304)
            elif case == "lp":
                rec = run_lp(args.solver, args.time_limit)
            elif case == "lp_c2":
                rec = run_lp(args.solver, args.time_limit, m=20_000, n=100_000, k=20)
            else:
                raise SystemExit(f"unknown case {case}")
            rec["process_state"] = "cold (first device call of the process)" if first else "warm"
            first = False
            rec["total_tool_seconds"] = time.time() - t0
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
