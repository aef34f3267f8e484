#!/usr/bin/env python3
"""Device simplex (solver='HIP') vs HiGHS on synthetic LPs: objective agreement, pivots, time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from smart_crossover.solver_caller.caller import SolverSettings  # noqa: E402
from smart_crossover.solver_caller.solving import solve_lp  # noqa: E402


def main():
    q = SolverSettings(log_console=0)
    print("   m      n   | HGS obj        s    | HIP obj        s   pivots  pivots/s  status  | rel.diff")
    for (m, n, k, seed) in [(200, 800, 4, 1), (1000, 4000, 5, 2), (2000, 8000, 5, 3), (4000, 12000, 5, 4)]:
        inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=(m >= 2000), frac_upper=0.3)
        lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
        t0 = time.perf_counter()
        ref = solve_lp(lp, "HGS", "default", q)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = solve_lp(lp, "HIP", "default", q)
        t_hip = time.perf_counter() - t0
        obj = out.obj_val if out.obj_val is not None else float("nan")
        its = out.iter_count or 0
        rel = abs(obj - ref.obj_val) / (1 + abs(ref.obj_val)) if ref.obj_val is not None else float("nan")
        print(f"{m:6d} {n:6d} | {ref.obj_val: .6e} {t_ref:6.2f} | {obj: .6e} {t_hip:6.2f} {its:7d} {its / max(t_hip, 1e-9):9.0f}  {out.status:8s}| {rel:.1e}",
              flush=True)


if __name__ == "__main__":
    main()
