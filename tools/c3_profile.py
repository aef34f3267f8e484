import cProfile, pstats, io, sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "smart-crossover_amd"))
import numpy as np, workloads
from contextlib import redirect_stdout
from smart_crossover.formats import OptTransport
from smart_crossover.network_methods.algorithms import network_crossover
from smart_crossover.solver_caller.caller import SolverSettings
inst = workloads.config3()
def run():
    ot = OptTransport(inst.s.copy(), inst.d.copy(), inst.M.copy())
    with redirect_stdout(io.StringIO()):
        return network_crossover(inst.x, ot=ot, method="tnet", solver="HIP", solver_settings=SolverSettings(log_console=0))
run(); run()
t0 = time.perf_counter(); run(); print("wall ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue())
