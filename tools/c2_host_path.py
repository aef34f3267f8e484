import io, sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "smart-crossover_amd"))
from contextlib import redirect_stdout
import numpy as np
import workloads
from smart_crossover.formats import GeneralLP
from smart_crossover.lp_methods.algorithms import get_perturb_problem
inst = workloads.config2()
lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
for i in range(6):
    t0 = time.perf_counter()
    with redirect_stdout(io.StringIO()):
        mgr = get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
    t1 = time.perf_counter()
    _ = mgr.lp_sub.A.indices
    t2 = time.perf_counter()
    print(i, round((t1 - t0) * 1e3, 1), "ms; touching lp_sub.A", round((t2 - t1) * 1e3, 1), "ms", mgr.perturb_info.get("cg_iters"))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with redirect_stdout(io.StringIO()):
    mgr = get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
