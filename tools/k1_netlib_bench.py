#!/usr/bin/env python3
"""K1 / K10 (column walks) on netlib_lp at config-5 size under the walk's knobs: LDS window off / auto / 4 tiles per load,
XCD map on / off, chunk 4096 / 2048.  MI355X.  usage: python tools/k1_netlib_bench.py"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from smart_crossover.hip import Context  # noqa: E402

inst = workloads.netlib_lp(1_000_000, 10_000_000)
m, n = inst.A.shape
ctx = Context(0)
dA = ctx.matrix(inst.A)
d = {k: ctx.to_device(getattr(inst, k)) for k in ("y", "x", "c", "l", "u", "b")}
vb = ctx.to_device(np.where(np.arange(n) % 7 == 0, -2, -1).astype(np.int8))
s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
k1_bytes = 12 * inst.A.nnz + 49 * n + 8 * m
k10_bytes = 12 * inst.A.nnz + 17 * n + 8 * m
variants = list(itertools.product((1, 0), (-1, 0, 1, 4), (4096, 2048)))
res = {v: {"k1": [], "k10": []} for v in variants}
ref = None
pres = None
for rnd in range(5):
    for v in variants:
        ctx.set_option("xcd_swizzle", v[0])
        ctx.set_option("window", v[1])
        ctx.set_option("chunk", v[2])
        ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        ctx.marker(0)
        for _ in range(5):
            ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        ctx.marker(1)
        for _ in range(5):
            pres = ctx.price(dA, d["y"], d["c"], vb, 1e-6, None, pres)
        ctx.marker(2)
        ctx.sync()
        res[v]["k1"].append(ctx.marker_elapsed(0, 1) / 5)
        res[v]["k10"].append(ctx.marker_elapsed(1, 2) / 5)
        got = s_d.download().view(np.uint64)
        if ref is None:
            ref = got
        assert np.array_equal(ref, got), v
for v in variants:
    t1, t10 = np.median(res[v]["k1"]), np.median(res[v]["k10"])
    print(f"swizzle={v[0]} window={v[1]:2d} chunk={v[2]}: K1 {t1:.4f} ms = {k1_bytes / t1 / 1e6 / 8000:.3f} of peak; K10 {t10:.4f} ms = {k10_bytes / t10 / 1e6 / 8000:.3f}", flush=True)
