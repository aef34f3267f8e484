#!/usr/bin/env python3
"""Perturbation crossover of BASELINE config 2 (2e4 x 1e5) end to end: get_perturb_problem + the re-solve of the
perturbed sub-LP on the device ('HIP') and, with --highs SECONDS, by HiGHS on the host (time-limited).

    python tools/spx_c2.py [--highs 300] [--rows 20000 --cols 100000 --k 20]
"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000)
    ap.add_argument("--cols", type=int, default=100_000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--highs", type=float, default=0.0, help="also time HiGHS on the sub-LP, with this time limit (s)")
    args = ap.parse_args()
    from smart_crossover.formats import GeneralLP
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller.solving import solve_lp
    inst = workloads.sparse_lp(args.rows, args.cols, args.k, seed=2, stratified=True)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    rec = {"rows": args.rows, "cols": args.cols}
    for rep in range(2):
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        t1 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            out = solve_lp(mgr.lp_sub, "HIP", "barrier", SolverSettings(presolve="on", log_console=0),
                           warm_start_solution=(mgr.get_subx(inst.x), inst.y))
            ok = alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x))
        t2 = time.perf_counter()
        rec[f"run{rep}"] = {"get_perturb_problem_ms": (t1 - t0) * 1e3, "resolve_ms": (t2 - t1) * 1e3,
                            "total_ms": (t2 - t0) * 1e3, "pivots": int(out.iter_count), "status": out.status,
                            "gap_ok": bool(ok), "sub_shape": list(mgr.lp_sub.A.shape)}
        print(json.dumps(rec[f"run{rep}"]), flush=True)
    if args.highs > 0:
        from smart_crossover.solver_caller.highs import HgsCaller
        t0 = time.perf_counter()
        try:
            with redirect_stdout(io.StringIO()):
                ref = solve_lp(mgr.lp_sub, "HGS", "default", SolverSettings(presolve="on", log_console=0, timeLimit=int(args.highs)))
            rec["highs"] = {"seconds": time.perf_counter() - t0, "status": ref.status,
                            "obj_rel_diff": (abs(ref.obj_val - out.obj_val) / (1 + abs(ref.obj_val))) if ref.status == "OPTIMAL" else None}
        except Exception as exc:       # time limit
            rec["highs"] = {"seconds": time.perf_counter() - t0, "status": f"{type(exc).__name__}: {exc}"[:200]}
        print(json.dumps(rec["highs"]), flush=True)


if __name__ == "__main__":
    main()
