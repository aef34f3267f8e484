#!/usr/bin/env python3
"""K2 (and the CG row pass) over the column-blocked row layout vs the plain row walk, MI355X.

    python tools/rb_bench.py [--m 1000000 --nb 10000000] [--structure staircase|uniform]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.hip import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1_000_000)
    ap.add_argument("--nb", type=int, default=10_000_000)
    ap.add_argument("--structure", default="staircase")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    t0 = time.time()
    sh = workloads.lp_shard(0, 1, m=args.m, n_block=args.nb, k=8, structure=args.structure)
    A = sh.row_block
    m, n = A.shape
    print(f"shard {A.shape} nnz={A.nnz} ({args.structure}) in {time.time() - t0:.1f}s", flush=True)
    ctx = Context(0)
    d_x, d_b, d_y = ctx.to_device(sh.x), ctx.to_device(sh.b), ctx.to_device(sh.y[:m])
    k2_bytes = 12 * A.nnz + 8 * n + 33 * m
    out = {}
    mats = {}
    variants = (("plain", 0, 0), ("blocked", 1, 0), ("staged", 1, 1))
    for name, opt, stage in variants:
        ctx.set_option("rowblock", opt)
        ctx.set_option("rb_stage_long", stage)
        mats[name] = ctx.row_shard(A)
        s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
        ctx.sync()
        t0 = time.time()
        ctx.score_rows(mats[name], d_x, d_b, d_y, 1e-3, s_p, flag)      # first call builds the layout
        ctx.sync()
        first = time.time() - t0
        out[name] = (s_p, flag, first)
        info = mats[name].rowblock()
        print(f"{name}: first call {first * 1e3:.1f} ms, layout {info}", flush=True)
    times = {"plain": [], "blocked": [], "staged": []}
    for _ in range(args.rounds):
        for name, opt, stage in variants:
            ctx.set_option("rowblock", opt)
            s_p, flag, _ = out[name]
            ctx.score_rows(mats[name], d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(0)
            for _ in range(args.reps):
                ctx.score_rows(mats[name], d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(1)
            times[name].append(ctx.marker_elapsed(0, 1) / args.reps)
    same = (np.array_equal(out["plain"][0].download().view(np.uint64), out["blocked"][0].download().view(np.uint64))
            and np.array_equal(out["plain"][1].download(), out["blocked"][1].download()))
    same = same and (np.array_equal(out["plain"][0].download().view(np.uint64), out["staged"][0].download().view(np.uint64))
                     and np.array_equal(out["plain"][1].download(), out["staged"][1].download()))
    print(f"bit-identical: {same}")
    for name in ("plain", "blocked", "staged"):
        t = np.array(times[name])
        print(f"K2 {name:8s} {np.median(t):.4f} ms (min {t.min():.4f}) = {k2_bytes / np.median(t) / 1e6:.0f} GB/s algorithmic, "
              f"{k2_bytes / np.median(t) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
