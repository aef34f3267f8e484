"""TNET tree basis identification -- API of the reference's ``network_methods/tree_BI.py``
(tree_basis_identify :12, max_weight_spanning_tree :32, push_tree_to_bfs :62).

K13 scans all S*D arcs and runs on the device; K14 and K15 touch only the S + D - 1 tree arcs (1.5 k at
the 784 x 784 configuration, a few hundred pushes) and stay on the host, like in the reference:

  K13  maximum-weight spanning tree of the arcs with a non-zero indicator: Boruvka on the device
       (csrc/sx_tree.hip; the reference calls scipy's minimum_spanning_tree on the negated weights,
       125 ms at 784 x 784 -- mostly its argsort of 614,656 weights; arcs with indicator 0 are not
       edges, exactly as there)
  K14  the tree system B x_B = b, solved by leaf elimination in O(S + D) instead of a sparse LU
  K15  the "push" loop that removes negative tree flows along 4-cycles
"""
from __future__ import annotations

from collections import deque
from typing import Tuple

import numpy as np
from smart_crossover.formats import OptTransport
from smart_crossover.network_methods.net_manager import OTManager
from smart_crossover.output import Basis


def tree_basis_identify(ot_manager: OTManager, flow_weights: np.ndarray) -> Tuple[Basis, int]:
    """Spanning-tree basis guided by the flow indicators, pushed to primal feasibility.
    Returns (basis, number of push iterations); cbasis marks the last node's row as basic."""
    tree = max_weight_spanning_tree(ot_manager.ot, flow_weights)
    vbasis, pushes = push_tree_to_bfs(ot_manager, tree)
    cbasis = np.concatenate([-np.ones(ot_manager.m - 1), np.array([0])])
    return Basis(vbasis, cbasis), pushes


def max_weight_spanning_tree(ot: OptTransport, flow_weights: np.ndarray) -> np.ndarray:
    """Linear arc indices (i*D + j, ascending) of a maximum-weight spanning tree of the bipartite
    graph suppliers x demanders weighted by ``flow_weights``."""
    from smart_crossover.hip.device import default_context
    S, D = ot.M.shape
    ctx = default_context()
    w = ctx.to_device(np.ascontiguousarray(flow_weights, dtype=np.float64).reshape(-1))
    tree = ctx.spanning_tree_ot(S, D, w)      # K13 on the device: Boruvka, sx_spanning_tree_ot_dev
    w.free()
    return tree


def _solve_tree_flows(S: int, D: int, tree: np.ndarray, s: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Flows on the tree arcs that meet every supply and demand (K14).  The tree system has +-1
    entries only; eliminating leaves first makes it triangular: a leaf's single arc must carry the
    leaf's whole remaining supply/demand."""
    if tree.size != S + D - 1:
        raise ValueError("the arcs with a non-zero flow indicator do not connect all suppliers and demanders; "
                         "TNET needs a connected support (e.g. a Sinkhorn plan)")
    ti, tj = np.divmod(tree, D)
    node_u, node_v = ti, S + tj
    incident = [[] for _ in range(S + D)]
    for e, (a, b) in enumerate(zip(node_u, node_v)):
        incident[a].append(e)
        incident[b].append(e)
    need = np.concatenate([np.asarray(s, float), np.asarray(d, float)])    # what still has to leave / arrive
    degree = np.array([len(v) for v in incident])
    done = np.zeros(tree.size, dtype=bool)
    flow = np.zeros(tree.size)
    leaves = deque(np.flatnonzero(degree == 1).tolist())
    while leaves:
        v = leaves.popleft()
        if degree[v] != 1:
            continue
        e = next(k for k in incident[v] if not done[k])
        other = node_v[e] if node_u[e] == v else node_u[e]
        flow[e] = need[v]
        need[other] -= flow[e]
        need[v] = 0.0
        done[e] = True
        degree[v] -= 1
        degree[other] -= 1
        if degree[other] == 1:
            leaves.append(other)
    return flow


def push_tree_to_bfs(ot_manager: OTManager, tree: np.ndarray) -> Tuple[np.ndarray, int]:
    """Turn the tree solution into a basic feasible one (K15).  Every negative arc (I1, J1) is
    repaired by pushing flow around the 4-cycle through the largest arc of its row (I1, J2) and of
    its column (I2, J1); the step is the smallest of the three flows involved."""
    ot = ot_manager.ot
    S, D = ot.s.size, ot.d.size
    flows = np.zeros(S * D)
    flows[tree] = _solve_tree_flows(S, D, np.asarray(tree, dtype=np.int64), ot.s, ot.d)
    T = flows.reshape(S, D)

    pushes = 0
    neg_i, neg_j = np.where(T < 0)
    for I1, J1 in zip(neg_i, neg_j):
        J2 = int(np.argmax(T[I1, :]))
        I2 = int(np.argmax(T[:, J1]))
        while T[I1, J1] < 0:
            assert T[I2, J1] > 0 and T[I1, J2] > 0
            assert T[I2, J2] == 0
            candidates = (-T[I1, J1], T[I1, J2], T[I2, J1])
            theta = min(candidates)
            which = candidates.index(theta)          # first minimum, as np.argmin
            T[I1, J1] += theta
            T[I2, J1] -= theta
            T[I1, J2] -= theta
            T[I2, J2] += theta
            if which == 1:
                J2 = int(np.argmax(T[I1, :]))
            elif which == 2:
                I2 = int(np.argmax(T[:, J1]))
            pushes += 1

    vbasis = -np.ones(ot_manager.n)
    vbasis[T.ravel() > 0] = 0
    return vbasis, pushes
