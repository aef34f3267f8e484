import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))
import workloads
from smart_crossover.hip import Context
sh = workloads.lp_shard(0, 1)
ctx = Context(0)
dC = ctx.column_shard(sh.col_block)
d = {k: ctx.to_device(getattr(sh, k)) for k in ("y", "x", "c", "l", "u")}
n = sh.col_block.shape[1]
s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
vb = ctx.to_device(np.where(np.arange(n) % 7 == 0, -2, -1).astype(np.int8))
k1_bytes = 12 * sh.col_block.nnz + 49 * n + 8 * sh.m
res = {}
pres = None
for rnd in range(5):
    for w in (1, 2, 4, 8):
        ctx.set_option("window", w)
        ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        ctx.marker(0)
        for _ in range(10):
            ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        ctx.marker(1)
        for _ in range(10):
            pres = ctx.price(dC, d["y"], d["c"], vb, 1e-6, None, pres)
        ctx.marker(2); ctx.sync()
        res.setdefault(w, []).append((ctx.marker_elapsed(0, 1) / 10, ctx.marker_elapsed(1, 2) / 10))
for w, v in res.items():
    a = np.array(v)
    print(f"tiles per window load {w}: K1 {np.median(a[:,0]):.4f} ms = {k1_bytes/np.median(a[:,0])/1e6/8000:.3f} of peak; K10 {np.median(a[:,1]):.4f} ms")
