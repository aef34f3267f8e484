#!/usr/bin/env python3
"""K2 over the column-blocked layout: the XCD-contiguous tile map (rb_long_xcd = 0) against the long-row super-tiles dealt
over the XCDs (1: always, -1: the automatic rule), bit-identity included.  MI355X.
    python tools/rb_long_xcd_bench.py [--workload shard|netlib] [modes ...]"""
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
    ap.add_argument("--workload", default="shard")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("weights", nargs="*", type=int, default=[1, -1])
    args = ap.parse_args()
    if args.workload == "shard":
        sh = workloads.lp_shard(0, 1)
        A, x, b, y = sh.row_block, sh.x, sh.b, sh.y
    else:
        inst = workloads.netlib_lp(1_000_000, 10_000_000)
        A, x, b, y = inst.A, inst.x, inst.b, inst.y
    m, n = A.shape
    ctx = Context(0)
    d_x, d_b, d_y = ctx.to_device(x), ctx.to_device(b), ctx.to_device(y[:m])
    k2_bytes = 12 * A.nnz + 8 * n + 33 * m
    ctx.set_option("rowblock", 1)
    dA = ctx.row_shard(A)
    ref_sp, ref_flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.set_option("rb_long_xcd", 0)
    ctx.score_rows(dA, d_x, d_b, d_y, 1e-3, ref_sp, ref_flag)
    ctx.sync()
    print(f"{args.workload}: {A.shape} nnz={A.nnz}, layout {dA.rowblock()}", flush=True)
    want = (ref_sp.download().view(np.uint64), ref_flag.download())
    variants = [0] + list(args.weights)
    times = {w: [] for w in variants}
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    same = {}
    for _ in range(args.rounds):
        for w in variants:
            ctx.set_option("rb_long_xcd", w)
            ctx.score_rows(dA, d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(0)
            for _ in range(args.reps):
                ctx.score_rows(dA, d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(1)
            times[w].append(ctx.marker_elapsed(0, 1) / args.reps)
            same[w] = bool(np.array_equal(s_p.download().view(np.uint64), want[0]) and np.array_equal(flag.download(), want[1]))
    for w in variants:
        t = np.array(times[w])
        print(f"K2 rb_long_xcd={w:4d}: {np.median(t):.4f} ms (min {t.min():.4f}) = {k2_bytes / np.median(t) / 1e6:.0f} GB/s algorithmic, "
              f"{k2_bytes / np.median(t) / 1e6 / 8000:.3f} of 8 TB/s, bit-identical {same[w]}", flush=True)


if __name__ == "__main__":
    main()
