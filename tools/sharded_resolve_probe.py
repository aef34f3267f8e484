#!/usr/bin/env python3
"""The column-sharded re-solve leg of bench.py (`sharded_resolve`) in ONE process, round by round: seconds of the
replicated re-solve, its route and iterations, columns added.  MI355X.
usage: python tools/sharded_resolve_probe.py [rows=20000] [batch=2048] [max_rounds=200] [limit_s=300] [sigma=0.3]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import torch  # noqa: E402

import workloads  # noqa: E402
from smart_crossover import distributed as D  # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from smart_crossover.hip import Context  # noqa: E402
from smart_crossover.solver_caller import solving  # noqa: E402

kv = dict(a.split("=") for a in sys.argv[1:])
rr, batch = int(kv.get("rows", 20000)), int(kv.get("batch", 2048))
max_rounds, limit_s = int(kv.get("max_rounds", 200)), float(kv.get("limit_s", 300))
sigma = float(kv.get("sigma", 0.3))
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = Context(0, stream.cuda_stream)
inst2 = workloads.netlib_lp(rr, 10 * rr, seed=17)
rng2 = np.random.default_rng(18)
lp2 = GeneralLP(inst2.A, inst2.b, inst2.c + sigma * rng2.standard_normal(10 * rr), inst2.l, np.where(np.isinf(inst2.u), 30.0, inst2.u), inst2.sense)
sh2 = D.ShardedLP(lp2, None, D.HipOps(ctx, torch))
t_start = time.perf_counter()
real = solving.solve_lp
rounds = []


def timed(lp, solver, method, settings, **kw):
    t0 = time.perf_counter()
    caller_box = {}
    real_gen = solving.generate_solver_caller

    def gen(*a, **k):
        caller_box["c"] = real_gen(*a, **k)
        return caller_box["c"]
    solving.generate_solver_caller = gen
    try:
        out = real(lp, solver, method, settings, **kw)
    finally:
        solving.generate_solver_caller = real_gen
    dt = time.perf_counter() - t0
    c = caller_box.get("c")
    rounds.append(dt)
    print(f"round {len(rounds):3d}: {lp.A.shape[1]:7d} columns, {method:14s} {dt:7.3f} s, status {out.status}, route {getattr(c, 'solved_by', '?')}, "
          f"iterations {out.iter_count}, elapsed {time.perf_counter() - t_start:6.1f} s", flush=True)
    if time.perf_counter() - t_start > limit_s:
        raise SystemExit("probe: time limit")
    return out


solving.solve_lp = timed
tr = []
x_R, y_R, R_R, basis_R, status_R, n_rounds = sh2.restricted_resolve(np.flatnonzero(inst2.x > 1e-6), solver="HIP", x_start=inst2.x, y_start=inst2.y,
                                                                    first_method="barrier", batch=batch, opt_tol=1e-6, trace=tr, max_rounds=max_rounds)
print(f"status {status_R}, rounds {n_rounds}, columns added {[len(t) for t in tr]}, total {time.perf_counter() - t_start:.1f} s, "
      f"objective {float(lp2.c[R_R] @ x_R) if status_R == 'OPTIMAL' else None}")
