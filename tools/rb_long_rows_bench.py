#!/usr/bin/env python3
"""K2 over the column-blocked layout for different heights of the long-row super-tiles (option rb_long_rows, read when the
layout is built): a long super-tile is one lane per linking row, chunk after chunk -- the fewer rows it holds, the shorter
the chain.  Bit-identity against the plain walk included.  MI355X.
    python tools/rb_long_rows_bench.py [--workload shard|netlib] [heights ...]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.hip import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="shard")
    ap.add_argument("--option", default="rb_long_rows", help="layout option the positional values are for (rb_long_rows | rb_dense_min)")
    ap.add_argument("heights", nargs="*", type=int, default=[64, 32, 16, 8])
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
    ctx.set_option("rowblock", 0)
    plain = ctx.row_shard(A)
    want_sp, want_flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.score_rows(plain, d_x, d_b, d_y, 1e-3, want_sp, want_flag)
    want = (want_sp.download().view(np.uint64), want_flag.download())
    plain.free()
    ctx.set_option("rowblock", 1)
    mats = {}
    for h in args.heights:
        ctx.set_option(args.option, h)
        mats[h] = ctx.row_shard(A)
        s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
        ctx.score_rows(mats[h], d_x, d_b, d_y, 1e-3, s_p, flag)     # builds the layout
        ctx.sync()
        same = bool(np.array_equal(s_p.download().view(np.uint64), want[0]) and np.array_equal(flag.download(), want[1]))
        print(f"{args.option}={h}: layout {mats[h].rowblock()}, bit-identical to the plain walk {same}", flush=True)
    times = {h: [] for h in args.heights}
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    for _ in range(5):
        for h in args.heights:
            ctx.score_rows(mats[h], d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(0)
            for _ in range(10):
                ctx.score_rows(mats[h], d_x, d_b, d_y, 1e-3, s_p, flag)
            ctx.marker(1)
            times[h].append(ctx.marker_elapsed(0, 1) / 10)
    for h in args.heights:
        t = np.array(times[h])
        print(f"K2 {args.option}={h:6d}: {np.median(t):.4f} ms (min {t.min():.4f}) = {k2_bytes / np.median(t) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
