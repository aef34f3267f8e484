#!/usr/bin/env python3
"""Entropic OT warm starts of B instance pairs on the 28 x 28 grid (BASELINE config 3 size, the reference's
driver loops over ten of them): one sx_sinkhorn_dev call per instance vs one sx_sinkhorn_batch_dev call.
    python tools/sinkhorn_bench.py [--B 10] [--reg 0.5] [--iters 1000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=10)
    ap.add_argument("--reg", type=float, default=0.5, help="small: the loop runs to the iteration limit")
    ap.add_argument("--iters", type=int, default=1000)
    args = ap.parse_args()
    from smart_crossover.hip.device import default_context
    ctx = default_context()
    side = 28
    n = side * side
    rng = np.random.default_rng(3)
    a = rng.random((args.B, n)) + 0.05
    b = rng.random((args.B, n)) + 0.05
    a /= a.sum(axis=1, keepdims=True)
    b /= b.sum(axis=1, keepdims=True)
    M = workloads.grid_cost(side)
    dM = ctx.to_device(M.reshape(-1))
    da, db = ctx.to_device(a.reshape(-1)), ctx.to_device(b.reshape(-1))
    d_u, d_v = ctx.empty(args.B * n, np.float64), ctx.empty(args.B * n, np.float64)
    rec = {"B": args.B, "grid": f"{side}x{side}", "reg": args.reg, "iters": args.iters}
    for rep in range(2):
        t0 = time.perf_counter()
        it_single = 0
        for k in range(args.B):
            ak = ctx.wrap(da.ptr + 8 * k * n, n, np.float64, owner=da)
            bk = ctx.wrap(db.ptr + 8 * k * n, n, np.float64, owner=db)
            r = ctx.sinkhorn(n, n, ak, bk, dM, args.reg, args.iters, 1e-9, None,
                             ctx.wrap(d_u.ptr + 8 * k * n, n, np.float64, owner=d_u), ctx.wrap(d_v.ptr + 8 * k * n, n, np.float64, owner=d_v))
            it_single += int(r.iters)
        t1 = time.perf_counter()
        u_single = d_u.download()
        res = ctx.sinkhorn_batch(n, n, args.B, da, db, dM, args.reg, args.iters, 1e-9, None, d_u, d_v)
        t2 = time.perf_counter()
        u_batch = d_u.download()
        it_batch = max(int(r.iters) for r in res)
        rec[f"run{rep}"] = {"single_calls_ms": (t1 - t0) * 1e3, "single_us_per_iteration_and_instance": (t1 - t0) * 1e6 / max(it_single, 1),
                            "batch_ms": (t2 - t1) * 1e3, "batch_us_per_iteration": (t2 - t1) * 1e6 / max(it_batch, 1),
                            "iterations_single_total": it_single, "iterations_batch": it_batch,
                            "max_rel_diff_u": float(np.max(np.abs(u_batch - u_single) / np.abs(u_single)))}
    r = rec["run1"]
    flop = 2 * 2.0 * n * n * 16                       # two 784 x 784 x 16 products per iteration (16 = MFMA tile of instances)
    r["mfma_fp64_tflops"] = flop / (r["batch_us_per_iteration"] * 1e-6) / 1e12
    r["speedup_over_instance_loop"] = r["single_calls_ms"] / r["batch_ms"]
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
