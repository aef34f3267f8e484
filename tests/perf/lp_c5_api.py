#!/usr/bin/env python3
"""The LP host path at full config-5 size through the kept Python API: ``get_perturb_problem`` on a
1e6 x 1e7 LP (8e7 entries) from host memory (scipy CSR + numpy vectors) to the restricted sub-problem in
host memory -- matrix upload and device transposition, K1, K2, the three index sets, K3, K4 (1000 CG
iterations = 2000 sparse products), K6 compaction, downloads.

    python tests/perf/lp_c5_api.py [--m 1000000] [--n 10000000] [--feas]
"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from smart_crossover.lp_methods.algorithms import get_perturb_problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1_000_000)
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--feas", action="store_true", help="is_feas=True: skip the projector (no CG)")
    args = ap.parse_args()
    t0 = time.time()
    sh = workloads.lp_shard(0, 1, m=args.m, n_block=args.n)
    rng = np.random.default_rng(11)
    sense = np.where(rng.random(args.m) < 0.5, "<", "=")
    lp = GeneralLP(sh.row_block, sh.b, sh.c, sh.l, sh.u, sense)
    print(f"[lp] generated {lp.A.shape} with {lp.A.nnz} entries in {time.time() - t0:.1f}s", flush=True)
    times = []
    for rep in range(2):
        fresh = GeneralLP(lp.A, lp.b.copy(), lp.c.copy(), lp.l.copy(), lp.u.copy(), lp.sense.copy()) if rep else lp
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()) as text:
            mgr = get_perturb_problem(fresh, sh.x, sh.y, 1e-3, 1e-3, args.feas)
        times.append((time.perf_counter() - t0) * 1e3)
        print(f"[lp] call {rep}: {times[-1]:.0f} ms", flush=True)
    info = getattr(mgr, "perturb_info", {}) or {}
    rec = {"case": "lp_c5_get_perturb_problem", "rows": args.m, "cols": args.n, "nnz": int(lp.A.nnz), "is_feas": args.feas,
           "first_call_ms": times[0], "second_call_ms_same_matrix_object": times[1],
           "fixed_columns": int(mgr.get_num_fixed_variables()), "fixed_rows": int(mgr.get_num_fixed_constraints()),
           "sub_problem_shape": list(mgr.lp_sub.A.shape), "sub_problem_nnz": int(mgr.lp_sub.A.nnz),
           "cg_iters": int(info.get("cg_iters", 0)), "cg_converged": bool(info.get("cg_converged", False)),
           "printed": text.getvalue().strip().splitlines()[-2:]}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
