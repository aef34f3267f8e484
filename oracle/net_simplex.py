"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the dual network simplex the device runs for the
re-solves of the network crossover (reference network_methods/net_manager.py:211-222 solve_subproblem ->
solve_mcf(..., warm_start_basis); column generation of network_methods/algorithms.py:109-140).

The reference hands these re-solves to Gurobi / CPLEX's network simplex; what it consumes is the optimal vertex,
its duals and its basis.  The algorithm below is the statement the device kernel (csrc/sx_netdual.hip) is tested
against pivot for pivot: same leaving arc, same entering arc, same flips, hence the same iteration count and the
same final tree.  Parity with the reference's solver is by certificate (optimal cost, feasibility, |B| = m): the
solver-chosen basis of a degenerate network LP is not unique -- unpinned, as DESIGN.md section 5 says.

Problem: min c.x, A x = b, 0 <= x <= u, every column of A = +1 at the arc's tail row, -1 at its head row.
Basis = spanning tree (rooted at the row whose slack is basic).  Dual simplex with the bound-flipping ratio test:

  start     potentials from the tree; a non-tree arc whose reduced cost has the wrong sign for its bound goes to
            the other bound (needs a finite capacity; otherwise status 5: not dual feasible, the caller takes
            the primal method); tree flows = subtree sums of b - A x_N
  leaving   the tree arc with the largest  violation^2 / |subtree|  -- |subtree| is the squared norm of its
            row of the basis inverse, i.e. exact dual steepest edge -- ties to the smaller node
  entering  arcs across the cut (exactly one end in the subtree) whose reduced cost moves towards zero, in
            ascending |reduced cost| (ties: smaller arc index); arcs are passed -- flipped to their other
            bound -- while the flips leave the leaving arc infeasible; the first arc that cannot be passed enters
  update    potentials of the subtree shift by the entering arc's reduced cost, the subtree is re-hung at the
            entering arc, tree flows follow from the new b - A x_N
"""
from __future__ import annotations

from typing import Dict

import numpy as np

TREE, LOWER, UPPER = 0, 1, -1


def _tree_arrays(V: int, root: int, tail, head, tree_arcs):
    """parent / pred arc by breadth-first search from the root; None when the arcs are not a spanning tree."""
    adj = [[] for _ in range(V)]
    for a in tree_arcs:
        adj[tail[a]].append((head[a], a))
        adj[head[a]].append((tail[a], a))
    parent = np.full(V, -2, dtype=np.int64)
    pred = np.full(V, -1, dtype=np.int64)
    parent[root] = -1
    order = [root]
    for v in order:
        for w, a in adj[v]:
            if parent[w] == -2:
                parent[w] = v
                pred[w] = a
                order.append(w)
    if len(order) != V:
        return None
    return parent, pred, np.asarray(order, dtype=np.int64)


def _subtree(parent, order):
    """size of every subtree and membership test helper (preorder positions)."""
    V = parent.size
    children = [[] for _ in range(V)]
    for v in order[1:]:
        children[parent[v]].append(v)
    for c in children:
        c.sort()
    pos = np.zeros(V, dtype=np.int64)
    size = np.ones(V, dtype=np.int64)
    pre = []
    stack = [(order[0], 0)]
    while stack:
        v, i = stack.pop()
        if i == 0:
            pos[v] = len(pre)
            pre.append(v)
        if i < len(children[v]):
            stack.append((v, i + 1))
            stack.append((children[v][i], 0))
        else:
            size[v] = len(pre) - pos[v]
    return pos, size, np.asarray(pre, dtype=np.int64)


def dual_network_simplex(tail, head, cost, cap, b, vbasis, root: int, feas_tol: float = 1e-9,
                         max_iter: int = 10_000_000, steepest: bool = True, bfrt: bool = True) -> Dict[str, object]:
    """vbasis: 0 tree arc, -1 at lower, -2 at upper.  Returns dict(status, x, y, vbasis, iters, flips, obj);
    status 0 optimal, 1 primal infeasible (dual unbounded), 3 iteration limit, 5 start not in the domain (the codes
    of sx_simplex_result)."""
    tail = np.asarray(tail, dtype=np.int64)
    head = np.asarray(head, dtype=np.int64)
    cost = np.asarray(cost, dtype=np.float64)
    cap = np.asarray(cap, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    V, E = b.size, tail.size
    state = np.where(vbasis == 0, TREE, np.where(vbasis == -2, UPPER, LOWER)).astype(np.int64)
    out = dict(status=5, iters=0, flips=0)
    if np.any((state == UPPER) & np.isinf(cap)):
        return out
    t = _tree_arrays(V, root, tail, head, np.flatnonzero(state == TREE))
    if t is None or np.count_nonzero(state == TREE) != V - 1:
        return out
    parent, pred, order = t
    y = np.zeros(V)
    for v in order[1:]:
        a, p = pred[v], parent[v]
        y[v] = cost[a] + y[p] if tail[a] == v else y[p] - cost[a]
    rc = (cost - y[tail]) + y[head]
    # start: dual feasibility by flipping
    wrong = ((state == LOWER) & (rc < 0)) | ((state == UPPER) & (rc > 0))
    if np.any(wrong & np.isinf(cap)):
        return out
    state[wrong] = -state[wrong]
    flips = int(np.count_nonzero(wrong))
    x = np.where(state == UPPER, cap, 0.0)
    x[state == TREE] = 0.0

    def tree_flows():
        beff = b.copy()
        nb = np.flatnonzero(state == UPPER)
        np.subtract.at(beff, tail[nb], x[nb])
        np.add.at(beff, head[nb], x[nb])
        exc = beff.copy()
        pos, size, pre = _subtree(parent, order)
        for v in pre[:0:-1]:            # children before parents
            exc[parent[v]] += exc[v]
        f = np.zeros(V)
        nz = pre[1:]
        f[nz] = np.where(tail[pred[nz]] == nz, exc[nz], -exc[nz])
        return f, pos, size, exc[root]

    iters = 0
    status = 3
    while iters < max_iter:
        f, pos, size, _ = tree_flows()
        nodes = np.flatnonzero(parent >= 0)
        a_of = pred[nodes]
        lo = -f[nodes]
        hi = f[nodes] - cap[a_of]
        viol = np.maximum(np.maximum(lo, hi), 0.0)
        viol[viol <= feas_tol] = 0.0
        if not np.any(viol > 0):
            status = 0
            break
        score = viol * viol / size[nodes] if steepest else viol
        k = int(np.argmax(score))       # first maximum: the smaller node
        v = int(nodes[k])
        a = int(a_of[k])
        to_lower = lo[k] >= hi[k]
        delta = float(viol[k])
        out_of_S = tail[a] == v
        tau = (-1 if out_of_S else 1) if to_lower else (1 if out_of_S else -1)
        inS_t = (pos[tail] >= pos[v]) & (pos[tail] < pos[v] + size[v])
        inS_h = (pos[head] >= pos[v]) & (pos[head] < pos[v] + size[v])
        cross_out = inS_t & ~inS_h
        cross_in = inS_h & ~inS_t
        nontree = state != TREE
        if tau > 0:
            elig = nontree & ((cross_out & (state == LOWER)) | (cross_in & (state == UPPER)))
        else:
            elig = nontree & ((cross_out & (state == UPPER)) | (cross_in & (state == LOWER)))
        cand = np.flatnonzero(elig)
        if cand.size == 0:
            status = 1
            break
        ratio = np.abs(rc[cand])
        srt = cand[np.lexsort((cand, ratio))]
        enter = int(srt[0])
        passed = []
        if bfrt:
            remaining = delta
            for j in srt:
                if cap[j] < remaining:   # passing j still leaves the arc infeasible
                    remaining -= cap[j]
                    passed.append(int(j))
                else:
                    enter = int(j)
                    break
            else:                        # every candidate passed: still infeasible -> no entering arc
                status = 1
                break
        theta = abs(rc[enter])
        # potentials of S, reduced costs of the crossing arcs
        inS = (pos >= pos[v]) & (pos < pos[v] + size[v])
        y[inS] += tau * theta
        rc = (cost - y[tail]) + y[head]
        for j in passed:
            state[j] = -state[j]
            x[j] = cap[j] if state[j] == UPPER else 0.0
        flips += len(passed)
        # the leaving arc lands on its bound, the entering arc joins the tree
        state[a] = LOWER if to_lower else UPPER
        x[a] = 0.0 if to_lower else cap[a]
        state[enter] = TREE
        x[enter] = 0.0
        rc[enter] = 0.0
        # S now hangs at the entering arc: the tree arrays follow from the arc set
        t = _tree_arrays(V, root, tail, head, np.flatnonzero(state == TREE))
        parent, pred, order = t
        iters += 1
    f, pos, size, _ = tree_flows()
    nodes = np.flatnonzero(parent >= 0)
    x[pred[nodes]] = f[nodes]
    vb = np.where(state == TREE, 0, np.where(state == UPPER, -2, -1)).astype(np.int8)
    out.update(status=status, x=x, y=y, vbasis=vb, iters=iters, flips=flips, obj=float(cost @ x))
    return out
