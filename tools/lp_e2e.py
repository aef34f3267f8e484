#!/usr/bin/env python3
"""Crossover wall time of one LP instance on the device, phase by phase (development tool; bench.py holds the
reported legs).  usage: lp_e2e.py c2|n1 [key=value ...]   (window=48 m=100000 n=1000000 for n1)"""
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
    which = sys.argv[1]
    kw = dict(a.split("=") for a in sys.argv[2:])
    from smart_crossover.formats import GeneralLP
    if os.environ.get("SX_ROWBLOCK") is not None:
        from smart_crossover.hip import default_context
        default_context().set_option("rowblock", int(os.environ["SX_ROWBLOCK"]))
    if os.environ.get("SX_OPTS"):   # context options for A/B runs: SX_OPTS="xcd_swizzle=0,slabs=0"
        from smart_crossover.hip import default_context
        for kv in os.environ["SX_OPTS"].split(","):
            key, val = kv.split("=")
            default_context().set_option(key, int(val))
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller.caller import SolverSettings
    from smart_crossover.solver_caller import solving
    if which == "c2":
        inst = workloads.config2()
    else:
        inst = workloads.netlib_lp(int(kw.get("m", 100_000)), int(kw.get("n", 1_000_000)), window=int(kw.get("window", 48)))
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    for _ in range(int(kw.get("gpp_reps", 1))):   # the last call is the one reported (the first also uploads the matrix)
        lp = GeneralLP(inst.A, inst.b.copy(), inst.c.copy(), inst.l.copy(), inst.u.copy(), inst.sense.copy())
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
        t1 = time.perf_counter()
    caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
    caller.read_genlp(mgr.lp_sub)
    caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
    if os.environ.get("SX_DUMP_MAPS"):     # (to name the frames of a native fault in the re-solve: profiles/r04/hipgraph_frames.md)
        with open(os.environ["SX_DUMP_MAPS"], "w") as f:
            f.write(open("/proc/self/maps").read())
    with redirect_stdout(io.StringIO()):
        caller.run_barrier()
        out = caller.return_output()
        ok = alg.check_perturb_output_precision(mgr, out.x, lp.c, float(lp.c @ inst.x))
    t2 = time.perf_counter()
    p = caller.pdlp
    rec = {"which": which, "sub_shape": list(mgr.lp_sub.A.shape), "get_perturb_problem_s": t1 - t0, "resolve_s": t2 - t1,
           "status": out.status, "gap_ok": bool(ok), "pivots": int(out.iter_count or 0), "solved_by": getattr(caller, "solved_by", None),
           "pdlp": None if p is None else {"status": int(p.status), "iters": int(p.iters), "restarts": int(p.restarts),
                                           "pr": p.primal_residual, "du": p.dual_residual, "gap": p.gap,
                                           "omega": p.primal_weight, "seconds": getattr(caller, "pdlp_seconds", None)}}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
