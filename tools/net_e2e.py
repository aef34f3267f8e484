#!/usr/bin/env python3
"""Whole network crossover (CNET_MCF) with the re-solves on the device (solver 'HIP': network simplex K16n) and in
HiGHS on the host cores; one JSON line per run.  usage: net_e2e.py [--cases 4096x32768,c4] [--solvers HIP,HGS]"""
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


def heartbeat():
    """gpurun takes seven silent minutes for a hang: say something once a minute."""
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(60)
            print(f"[net_e2e] running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    threading.Thread(target=beat, daemon=True).start()


def main():
    heartbeat()
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="4096x32768,c4")
    ap.add_argument("--solvers", default="HIP,HGS")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--ns-block", type=int, default=0, help="ctx option ns_block (arcs priced per lane and block; 0 = by size)")
    ap.add_argument("--ns-lds", type=int, default=1)
    ap.add_argument("--netsimplex", type=int, default=-1)
    ap.add_argument("--netdual", type=int, default=-1, help="ctx option netdual (0: primal network simplex only)")
    ap.add_argument("--nd-grid", type=int, default=0, help="ctx option nd_grid (workgroups of the dual method's grid)")
    ap.add_argument("--c3", action="store_true", help="also time bench.py's network leg (TNET on config 3)")
    args = ap.parse_args()
    if "HIP" in args.solvers:
        from smart_crossover.hip import default_context
        default_context().set_option("ns_block", args.ns_block)
        default_context().set_option("ns_lds", args.ns_lds)
        default_context().set_option("netsimplex", args.netsimplex)
        default_context().set_option("netdual", args.netdual)
        default_context().set_option("nd_grid", args.nd_grid)
    if args.c3:
        import bench
        rec = bench.crossover_network()
        rec["netsimplex"] = args.netsimplex
        print(json.dumps(rec), flush=True)
    from smart_crossover.formats import MinCostFlow
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.solver_caller import hip as hipmod
    from smart_crossover.solver_caller import highs as hgsmod
    for case in [c for c in args.cases.split(",") if c]:
        if case == "c4":
            V, E = 2 ** 17, 2 ** 20
        else:
            V, E = (int(t) for t in case.split("x"))
        for solver in args.solvers.split(","):
            for rep in range(args.repeat if solver == "HIP" else 1):
                inst = workloads.mcf(V, E, 3)
                mcf = MinCostFlow(A=inst.A.copy(), b=inst.b.copy(), c=inst.c.copy(), u=inst.u.copy())
                solves = []
                cls = hipmod.HipCaller if solver == "HIP" else hgsmod.HgsCaller
                orig = cls.return_output

                def spy(self, _orig=orig):
                    out = _orig(self)
                    solves.append({"cols": int(self.get_A().shape[1]) if hasattr(self, "get_A") else None,
                                   "iters": int(out.iter_count), "ms": out.runtime.total_seconds() * 1e3,
                                   "by": getattr(self, "solved_by", solver)})
                    return out

                cls.return_output = spy
                buf = io.StringIO()
                t0 = time.perf_counter()
                try:
                    with redirect_stdout(buf):
                        out = network_crossover(inst.x.copy(), mcf=mcf, method="cnet_mcf", solver=solver)
                finally:
                    cls.return_output = orig
                wall = time.perf_counter() - t0
                rec = {"case": f"cnet_mcf V={V} E={E}", "solver": solver, "ns_block": args.ns_block, "ns_lds": args.ns_lds, "netsimplex": args.netsimplex, "netdual": args.netdual, "nd_grid": args.nd_grid, "run": rep, "wall_ms": wall * 1e3,
                       "solver_ms": sum(s["ms"] for s in solves), "simplex_iterations": int(out.iter_count),
                       "solves": solves, "cost": float(inst.c @ out.x[:E])}
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
