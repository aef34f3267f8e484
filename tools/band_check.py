#!/usr/bin/env python3
"""Development check of the sparse crossover (K16s) on netlib-style LPs of growing size against HiGHS.
usage: band_check.py m n [window] [pdlp_iters]"""
import io, json, os, sys, time
from contextlib import redirect_stdout
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))
import workloads
from scipy.optimize import linprog

m, n = int(sys.argv[1]), int(sys.argv[2])
window = int(sys.argv[3]) if len(sys.argv) > 3 else 48
os.environ.setdefault("SX_LP_CROSSOVER", "band")
if len(sys.argv) > 4:
    os.environ["SX_PDLP_ITERS"] = sys.argv[4]
from smart_crossover.formats import GeneralLP
from smart_crossover.lp_methods import algorithms as alg
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.solver_caller import solving
inst = workloads.netlib_lp(m, n, window=window)
lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
with redirect_stdout(io.StringIO()):
    mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
sub = mgr.lp_sub
caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
caller.read_genlp(sub)
caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
t0 = time.perf_counter()
with redirect_stdout(io.StringIO()):
    caller.run_barrier()
t1 = time.perf_counter()
res = caller._res
x, y, vb, cb = caller._x, caller._y, caller._vb, caller._cb
lt = np.asarray(sub.sense) == "<"
s_p = sub.b - sub.A @ x
rc = sub.c - sub.A.T @ y
pd = caller.pdlp
rec = {"pdlp": None if pd is None else [int(pd.status), int(pd.iters), pd.primal_residual, pd.dual_residual, pd.gap], "sub": list(sub.A.shape), "solved_by": caller.solved_by, "status": int(res.status), "iters": int(res.iters), "added": int(res.phase1_iters),
       "seconds": t1 - t0, "obj": float(sub.c @ x), "res_eq": float(np.abs(s_p[~lt]).max(initial=0)), "res_lt": float(-s_p[lt].min(initial=0)),
       "bound_viol": float(max((sub.l - x).max(), (x - sub.u).max())), "n_basic": int((vb == 0).sum() + (cb == 0).sum()),
       "rc_low_min": float(rc[vb == -1].min(initial=0)), "rc_up_max": float(rc[vb == -2].max(initial=0)), "rc_basic": float(np.abs(rc[vb == 0]).max(initial=0)),
       "n_super": int((vb == -3).sum())}
if m <= 20000:
    t0 = time.perf_counter()
    ref = linprog(sub.c, A_ub=sub.A[lt], b_ub=sub.b[lt], A_eq=sub.A[~lt], b_eq=sub.b[~lt], bounds=np.c_[sub.l, sub.u], method="highs")
    rec.update({"highs_status": int(ref.status), "highs_obj": float(ref.fun) if ref.status == 0 else None, "highs_s": time.perf_counter() - t0})
print(json.dumps(rec), flush=True)
