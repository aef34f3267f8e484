#!/usr/bin/env python3
"""(test infrastructure: uses oracle/) CNET_MCF on the CPU with the oracle's dual network simplex (oracle/net_simplex.py) for the re-solves:
iterations per column-generation round, next to HiGHS' and the device primal method's (profiles/r02/network_simplex_ab.jsonl).
usage: tests/perf/net_dual_proto.py V E [--no-bfrt] [--no-steepest]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from oracle import net_path as N  # noqa: E402
from oracle.net_simplex import dual_network_simplex  # noqa: E402


def main():
    V, E = int(sys.argv[1]), int(sys.argv[2])
    bfrt = "--no-bfrt" not in sys.argv
    steep = "--no-steepest" not in sys.argv
    inst = workloads.mcf(V, E, 3)
    A, b, c, u, x0 = inst.A, inst.b, inst.c, inst.u, inst.x
    m, n = A.shape
    ind, _ = N.mcf_flow_indicators(A, x0, u)
    queue = N.rank_desc(ind)
    factor = np.max(np.abs(c))
    cs = c / factor
    low, up = N.mcf_initial_partition(x0, u)
    ext = N.mcf_bigM_extension(A, b, cs, u, up, m * np.max(cs))
    A1, b1, c1, u1 = ext["A"], ext["b"], ext["c"], ext["u"]
    C1 = sp.csc_matrix(A1)
    vb, cb = N.mcf_initial_basis(n, m, up)
    non_fix = ext["artificial"].copy()
    fix = np.arange(n)
    fix_low, fix_up = low, up
    target = int(10 * m) if n / m > 1000 else int(1.2 * m)
    left = 0
    rnd = 0
    total = 0
    while True:
        right = min(target, queue.size)
        non_fix, fix, fix_low, fix_up = N.release_columns(non_fix, fix, fix_low, fix_up, queue[left:right])
        sub = N.mcf_sub_problem(A1, b1, c1, u1, non_fix, fix_up)
        Cs = sp.csc_matrix(sub["A"])
        tail = np.empty(non_fix.size, dtype=np.int64)
        head = np.empty(non_fix.size, dtype=np.int64)
        for j in range(non_fix.size):
            p0 = Cs.indptr[j]
            r0, r1 = Cs.indices[p0], Cs.indices[p0 + 1]
            tail[j], head[j] = (r0, r1) if Cs.data[p0] > 0 else (r1, r0)
        t0 = time.time()
        res = dual_network_simplex(tail, head, sub["c"], sub["u"], sub["b"], vb[non_fix], root=m, bfrt=bfrt, steepest=steep)
        dt = time.time() - t0
        rnd += 1
        print(f"round {rnd}: cols {non_fix.size} status {res['status']} iters {res['iters']} flips {res['flips']} "
              f"obj {res.get('obj', float('nan')) * factor:.6f} ({dt:.1f} s)", flush=True)
        if res["status"] != 0:
            break
        total += res["iters"]
        vb = -np.ones(c1.size, dtype=int)
        vb[non_fix] = res["vbasis"]
        vb[fix_up] = -2
        x = np.zeros(c1.size)
        x[non_fix] = res["x"]
        x[fix_up] = u1[fix_up]
        if N.mcf_is_optimal(A1, c1, res["y"], vb, x, ext["artificial"]):
            print(f"optimal after {rnd} rounds, {total} dual iterations, cost {float(c1 @ x) * factor:.6f}")
            break
        target = int(N.CG_RATIO * target)
        left = right
        if left >= queue.size:
            print("column generation fails")
            break


if __name__ == "__main__":
    main()
