#!/usr/bin/env python3
"""Time of the dense-inverse fold (rank-64 update of an m x m fp64 matrix) inside a device simplex run: installs a
crash basis of m columns = m / 64 folds; the trace line gives the wall time.  usage: fold_bench.py [m]"""
import io, os, sys, time
from contextlib import redirect_stdout
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))
import workloads
os.environ["SX_SPX_TRACE"] = "1"
os.environ["SX_LP_CROSSOVER"] = "dense"
from smart_crossover.formats import GeneralLP
from smart_crossover.lp_methods import algorithms as alg
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.solver_caller import solving
inst = workloads.config2()
lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
with redirect_stdout(io.StringIO()):
    mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
for env in (sys.argv[1:2] or [None]):
    env = None if env in (None, "mfma") else "0"
    if env is None: os.environ.pop("SX_SPX_BLAS", None)
    else: os.environ["SX_SPX_BLAS"] = env
    print("fold:", "scalar kernel" if env == "0" else "matrix cores", flush=True)
    caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
    caller.read_genlp(mgr.lp_sub)
    caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        caller.run_barrier()
    print("  resolve %.3f s, status %s" % (time.perf_counter() - t0, caller.return_status()), flush=True)
