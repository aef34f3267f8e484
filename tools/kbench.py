#!/usr/bin/env python3
"""Interleaved A/B timing of the scoring kernels under different tuning knobs (one process,
same data, several rounds; median and min per variant).

    python tools/kbench.py [--workload c5|c2] [--structure staircase|uniform] [--rounds 7]
"""
import argparse
import itertools
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
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--structure", default="staircase")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--window", type=int, default=4096, help="staircase window W (rows) of the synthetic LP")
    ap.add_argument("--variants", default="all", help="'all' or 'default' (swizzle on, nt off, chunk 4096 only)")
    args = ap.parse_args()
    m, nb, k = (1_000_000, 10_000_000, 8) if args.workload == "c5" else (20_000, 100_000, 20)
    structure = args.structure if args.workload == "c5" else "uniform"
    sh = workloads.lp_shard(0, 1, m=m, n_block=nb, k=k, structure=structure, window=args.window)
    ctx = Context(0)
    dC, dR = ctx.column_shard(sh.col_block), ctx.row_shard(sh.row_block)
    d = {kk: ctx.to_device(getattr(sh, kk)) for kk in ("y", "x", "c", "l", "u", "b")}
    s_d, code = ctx.empty(nb, np.float64), ctx.empty(nb, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    k1_bytes = 12 * sh.col_block.nnz + 49 * nb + 8 * m
    k2_bytes = 12 * sh.row_block.nnz + 8 * nb + 33 * m
    k10_bytes = k1_bytes - 24 * nb   # reads idx/val/colptr/c/vbasis/y, writes rc
    vb = ctx.to_device(np.where(np.arange(nb) % 7 == 0, -2, -1).astype(np.int8))
    rc = ctx.empty(nb, np.float64)
    pres = None

    variants = list(itertools.product((0, 1), (0, 1), (4096, 2048), (0,)))   # swizzle, nt, chunk, lds-window
    if args.variants == "default":
        variants = [(1, 0, 4096, 0), (1, 0, 4096, 1)]
    elif args.variants == "window":   # LDS operand window, tiles per workgroup
        variants = [(1, 0, 4096, w) for w in (0, 1, 4, -1)]
    elif args.variants == "policy":   # cache policy of the streamed loads (see sx_segwalk.h)
        variants = [(1, p, 4096, 0) for p in (0, 1, 2, 16, 17, 18)]
    elif args.variants == "runwalk":  # windowed walk tile by tile (0) / with its loads one step ahead (1), sx_runwalk.h
        variants = [(1, 0, 4096, -1, pf) for pf in (0, 1)]
    elif args.variants == "slabs":    # operand slabs of the walks without locality (sx_slabs.h): never / automatic
        variants = [(1, 0, 4096, -1, 0, sl) for sl in ([0, -1] + [int(q) for q in os.environ.get("SLABS", "").split(",") if q])]
    res = {v: {"k1": [], "k2": [], "k10": []} for v in variants}
    ref = None
    for rnd in range(args.rounds):
        for v in variants:
            ctx.set_option("xcd_swizzle", v[0])
            ctx.set_option("nt_stream", v[1])
            ctx.set_option("chunk", v[2])
            ctx.set_option("window", v[3])
            if len(v) > 4:
                ctx.set_option("run_prefetch", v[4])
            if len(v) > 5:
                ctx.set_option("slabs", v[5])
            ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)   # warm (layouts are built on first use)
            ctx.score_rows(dR, d["x"], d["b"], d["y"], 1e-3, s_p, flag)
            ctx.marker(0)
            for _ in range(args.reps):
                ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
            ctx.marker(1)
            for _ in range(args.reps):
                ctx.score_rows(dR, d["x"], d["b"], d["y"], 1e-3, s_p, flag)
            ctx.marker(2)
            pres = ctx.price(dC, d["y"], d["c"], vb, 1e-6, rc, pres)
            ctx.marker(3)
            for _ in range(args.reps):
                ctx.price(dC, d["y"], d["c"], vb, 1e-6, rc, pres)
            ctx.marker(4)
            res[v]["k10"].append(ctx.marker_elapsed(3, 4) / args.reps)
            res[v]["k1"].append(ctx.marker_elapsed(0, 1) / args.reps)
            res[v]["k2"].append(ctx.marker_elapsed(1, 2) / args.reps)
            if rnd == 0:   # knobs must never change results
                got = (s_d.download().tobytes(), code.download().tobytes(), s_p.download().tobytes(), flag.download().tobytes(),
                       rc.download().tobytes(), pres.download().tobytes())
                if ref is None:
                    ref = got
                assert got == ref or os.environ.get("SX_KBENCH_NOCHECK"), f"variant {v} changed the results"   # (NOCHECK: phase-stripped library variants)
    print(f"workload {args.workload}/{structure} window={args.window}: K1 bytes {k1_bytes/1e9:.3f} GB, K2 bytes {k2_bytes/1e9:.3f} GB")
    print("swz nt chunk win |  K1 med ms   min ms   GB/s(med) |  K2 med ms   min ms   GB/s(med) | K10 med ms   min ms   GB/s(med)")
    for v in variants:
        extra = "".join(f"/{q}" for q in v[4:])
        a, b, p10 = np.array(res[v]["k1"]), np.array(res[v]["k2"]), np.array(res[v]["k10"])
        print(f" {v[0]} {v[1]:2d}  {v[2]:4d} {v[3]:3d}{extra} |  {np.median(a):8.4f} {a.min():8.4f} {k1_bytes/np.median(a)/1e6:9.0f} |"
              f"  {np.median(b):8.4f} {b.min():8.4f} {k2_bytes/np.median(b)/1e6:9.0f} |"
              f"  {np.median(p10):8.4f} {p10.min():8.4f} {k10_bytes/np.median(p10)/1e6:9.0f}")


if __name__ == "__main__":
    main()
