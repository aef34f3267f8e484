#!/usr/bin/env python3
"""Device simplex (solver='HIP') vs HiGHS on synthetic LPs: objective agreement, pivots, time -- with the
inverse updated after every pivot ("spx_defer" 0) and with a batch's updates folded in together (1).

    python tools/spx_bench.py [--rows 4000[,8000]] [--no-ref] [--modes 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from smart_crossover.hip import default_context  # noqa: E402
from smart_crossover.solver_caller.caller import SolverSettings  # noqa: E402
from smart_crossover.solver_caller.solving import solve_lp  # noqa: E402

SIZES = {200: (800, 4, 1), 500: (2000, 4, 5), 1000: (4000, 5, 2), 2000: (8000, 5, 3), 4000: (12000, 5, 4),
         8000: (24000, 5, 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="200,500,1000,2000,4000")
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--no-ref", action="store_true", help="skip the HiGHS solve (minutes at 4000 rows)")
    ap.add_argument("--no-graph", action="store_true",
                    help="direct launches instead of hipGraph replay (needed under rocprofv3 --kernel-trace, which "
                         "segfaults inside hipGraphLaunch on this image)")
    ap.add_argument("--pricing", type=int, default=None, help="0 Dantzig, 1 Devex (default: the library's)")
    args = ap.parse_args()
    q = SolverSettings(log_console=0)
    ctx = default_context()
    if args.no_graph:
        ctx.set_option("graph", 0)
    if args.pricing is not None:
        ctx.set_option("spx_pricing", args.pricing)
    rows = [int(v) for v in args.rows.split(",")]
    modes = [int(v) for v in args.modes.split(",")]
    warm = workloads.sparse_lp(60, 200, 3, seed=1, stratified=False, frac_upper=0.3)
    solve_lp(GeneralLP(warm.A, warm.b, warm.c, warm.l, warm.u, warm.sense), "HIP", "default", q)   # library warm-up
    print("   m      n   | HGS obj        s    | mode | HIP obj        s   pivots  pivots/s  status  | rel.diff")
    for m in rows:
        n, k, seed = SIZES[m]
        inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=(m >= 2000), frac_upper=0.3)
        lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
        ref_obj, t_ref = float("nan"), float("nan")
        if not args.no_ref:
            t0 = time.perf_counter()
            ref_obj = solve_lp(lp, "HGS", "default", q).obj_val
            t_ref = time.perf_counter() - t0
        for mode in modes:
            ctx.set_option("spx_defer", mode)
            t0 = time.perf_counter()
            out = solve_lp(lp, "HIP", "default", q)
            t_hip = time.perf_counter() - t0
            obj = out.obj_val if out.obj_val is not None else float("nan")
            its = out.iter_count or 0
            rel = abs(obj - ref_obj) / (1 + abs(ref_obj))
            print(f"{m:6d} {n:6d} | {ref_obj: .6e} {t_ref:6.2f} | {mode:4d} | {obj: .6e} {t_hip:6.2f} {its:7d} "
                  f"{its / max(t_hip, 1e-9):9.0f}  {out.status:8s}| {rel:.1e}", flush=True)
    ctx.set_option("spx_defer", -1)


if __name__ == "__main__":
    main()
