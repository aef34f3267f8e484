#!/usr/bin/env python3
"""Per-kernel timings of the network-side and elementwise kernels at BASELINE sizes (HIP events on the
library's stream, median of several repetitions), with the algorithmic bytes of SURVEY.md 8(d) and the
CPU oracle timed beside each one on one host core.

    python tests/perf/netbench.py [--reps 20]

K7  MCF flow indicators       config 4: V = 2^17, E = 2^20
K9  ranking (argsort desc)    config 4 (E keys) and config 3 (S*D keys)
K10 pricing                   config 4 incidence matrix
K8  OT flow indicators        config 3: 784 x 784
K13 spanning tree (Boruvka)   config 3
OT pricing                    config 3
K3  perturb cost / x_real     n = 1e7
select (np.where)             n = 1e7 flags
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import workloads  # noqa: E402
from oracle import lp_path as L  # noqa: E402  (baseline timing only)
from oracle import net_path as N  # noqa: E402
from smart_crossover.hip import Context  # noqa: E402

HBM_PEAK = 8000.0


def timed(ctx, fn, reps):
    fn()                                    # warm (first-use allocations)
    ms = []
    for _ in range(reps):
        ctx.marker(0)
        fn()
        ctx.marker(1)
        ms.append(ctx.marker_elapsed(0, 1))
    return float(np.median(ms)), float(np.min(ms))


def cpu(fn, max_reps=3):
    fn()
    t = []
    for _ in range(max_reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3


def emit(name, workload, gpu_ms, gpu_min, nbytes, cpu_ms):
    rec = {"kernel": name, "workload": workload, "gpu_ms_median": gpu_ms, "gpu_ms_min": gpu_min,
           "algorithmic_bytes": int(nbytes), "achieved_GBps": nbytes / gpu_ms / 1e6,
           "frac_of_hbm_peak": nbytes / gpu_ms / 1e6 / HBM_PEAK, "cpu_oracle_ms_1core": cpu_ms,
           "speedup_vs_cpu": cpu_ms / gpu_ms}
    print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    ctx = Context(0)
    R = args.reps

    # ---------------------------------------------------------------- config 4 (MCF)
    inst = workloads.config4()
    V, E = inst.A.shape
    dA = ctx.matrix(inst.A)
    dx, du, dc = ctx.to_device(inst.x), ctx.to_device(inst.u), ctx.to_device(inst.c)
    ind = ctx.empty(E, np.float64)
    g, gm = timed(ctx, lambda: ctx.flow_indicator_mcf(dA, dx, du, ind), R)
    emit("K7 flow_indicator_mcf", f"c4 V={V} E={E}", g, gm, 32 * E + 16 * V + 40 * E,
         cpu(lambda: N.mcf_flow_indicators(inst.A, inst.x, inst.u)))
    q = ctx.empty(E, np.int64)
    g, gm = timed(ctx, lambda: ctx.argsort_desc(ind, q), R)
    ind_h = ind.download()
    emit("K9 argsort_desc", f"c4 {E} keys", g, gm, 192 * E, cpu(lambda: N.rank_desc(ind_h)))
    rng = np.random.default_rng(1)
    y = ctx.to_device(rng.standard_normal(V))
    vb_h = rng.integers(-2, 1, E).astype(np.int8)
    vb = ctx.to_device(vb_h)
    rc = ctx.empty(E, np.float64)
    res = ctx.price(dA, y, dc, vb, 1e-6, rc)
    g, gm = timed(ctx, lambda: ctx.price(dA, y, dc, vb, 1e-6, rc, res), R)
    y_h = y.download()
    emit("K10 price", f"c4 V={V} E={E}", g, gm, 12 * inst.A.nnz + 25 * E + 8 * V,
         cpu(lambda: N.mcf_reduced_cost(inst.A, inst.c, y_h, vb_h.astype(int))))
    dA.free()

    # ---------------------------------------------------------------- config 3 (OT)
    ot = workloads.config3()
    S, D = ot.M.shape
    dX, ds, dd, dM = ctx.to_device(ot.x), ctx.to_device(ot.s), ctx.to_device(ot.d), ctx.to_device(ot.M.ravel())
    oind = ctx.empty(S * D, np.float64)
    g, gm = timed(ctx, lambda: ctx.flow_indicator_ot(S, D, dX, ds, dd, oind), R)
    emit("K8 flow_indicator_ot", f"c3 {S}x{D}", g, gm, 16 * S * D + 8 * (S + D),
         cpu(lambda: N.ot_flow_indicators(ot.x, ot.s, ot.d)))
    oq = ctx.empty(S * D, np.int64)
    g, gm = timed(ctx, lambda: ctx.argsort_desc(oind, oq), R)
    oind_h = oind.download()
    emit("K9 argsort_desc", f"c3 {S * D} keys", g, gm, 192 * S * D, cpu(lambda: N.rank_desc(oind_h)))
    flags = ctx.empty(S * D, np.uint8)
    g, gm = timed(ctx, lambda: ctx._lib.sx_spanning_tree_ot_dev(ctx.handle, S, D, oind.ptr, flags.ptr), R)

    def scipy_tree():
        import scipy.sparse as sp
        from scipy.sparse import csgraph
        W = oind_h.reshape(S, D)
        graph = sp.bmat([[None, sp.csr_matrix(-W)], [sp.csr_matrix((D, S)), None]], format="csr")
        return csgraph.minimum_spanning_tree(graph)
    rounds = int(np.ceil(np.log2(S + D)))
    emit("K13 spanning_tree_ot", f"c3 {S}x{D}, {rounds} Boruvka rounds", g, gm, rounds * 2 * 8 * S * D, cpu(scipy_tree))
    yy = ctx.to_device(np.random.default_rng(2).standard_normal(S + D))
    orc = ctx.empty(S * D, np.float64)
    ores = ctx.price_ot(S, D, dM, yy, 1e-6, orc)
    g, gm = timed(ctx, lambda: ctx.price_ot(S, D, dM, yy, 1e-6, orc, ores), R)
    yy_h = yy.download()
    emit("OT price", f"c3 {S}x{D}", g, gm, 16 * S * D + 8 * (S + D), cpu(lambda: N.ot_reduced_cost(ot.M, yy_h)))

    # ---------------------------------------------------------------- elementwise at n = 1e7
    n = 10_000_000
    rng = np.random.default_rng(3)
    x_h = rng.random(n)
    l_h = np.zeros(n)
    u_h = np.where(rng.random(n) < 0.3, 2.0, np.inf)
    c_h = rng.standard_normal(n)
    xi_h = L.xi_vector(n)
    x, l, u, c, xi = (ctx.to_device(v) for v in (x_h, l_h, u_h, c_h, xi_h))
    out = ctx.empty(n, np.float64)
    g, gm = timed(ctx, lambda: ctx.perturb_cost(n, x, l, u, c, xi, 0.37, False, out), R)
    emit("K3 perturb_cost", f"n={n}", g, gm, 48 * n, cpu(lambda: L.perturb_cost(c_h, x_h, l_h, u_h, xi_h, 0.37, False)))
    code_h = (rng.random(n) < 0.4).astype(np.uint8) | ((rng.random(n) < 0.1).astype(np.uint8) << 1)
    code = ctx.to_device(code_h)
    idx, cnt = ctx.empty(n, np.int64), ctx.empty(1, np.int64)
    g, gm = timed(ctx, lambda: ctx.select_indices(code, 1, idx, cnt), R)
    k = int(np.count_nonzero(code_h & 1))
    emit("select_indices", f"n={n}, {k} selected", g, gm, n + 8 * k, cpu(lambda: np.where(code_h & 1)[0]))


if __name__ == "__main__":
    main()
