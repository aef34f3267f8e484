"""On-disk instance formats either side of the network path.

* DIMACS min-cost-flow ``.min`` -> :class:`MinCostFlow`   (reference: scripts/min2mcf.py:12-41)
* MNIST ``idx3-ubyte`` images -> :class:`OptTransport`     (reference: scripts/mnist2ot.py:12-61)

Same conventions as the reference's converters: node-arc incidence with +1 at the tail and -1 at the
head, arc lower bounds ignored, supplies as written in the file; OT instances keep only the non-zero
pixels of each image and use the Manhattan distance between pixel positions as cost.  Pure numpy (the
reference needs ``idx2numpy``); nothing here touches the device.
"""
from __future__ import annotations

import struct
from typing import List, Sequence

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import MinCostFlow, OptTransport


def read_dimacs_min(path: str, name: str = "mcf") -> MinCostFlow:
    """Parse a DIMACS ``.min`` file.  ``p min V E`` gives the sizes, ``n id supply`` the non-zero
    supplies, ``a tail head low cap cost`` one arc per line (1-based node ids, file order = column
    order).  A self loop keeps the single entry +1, as the reference's two assignments leave it."""
    n_nodes = n_arcs = None
    supplies, arcs = [], []
    with open(path, "r") as fh:
        for line in fh:
            tag = line[:1]
            if tag == "a":
                arcs.append(line.split()[1:6])
            elif tag == "n":
                supplies.append(line.split()[1:3])
            elif tag == "p" and n_nodes is None:
                parts = line.split()
                n_nodes, n_arcs = int(parts[2]), int(parts[3])
    if n_nodes is None:
        raise ValueError(f"{path}: no problem line ('p min <nodes> <arcs>')")
    arc = np.asarray(arcs, dtype=np.int64).reshape(-1, 5)
    if arc.shape[0] != n_arcs:
        raise ValueError(f"{path}: problem line announces {n_arcs} arcs, file holds {arc.shape[0]}")
    b = np.zeros(n_nodes)
    if supplies:
        sup = np.asarray(supplies, dtype=np.int64).reshape(-1, 2)
        b[sup[:, 0] - 1] = sup[:, 1]                    # later lines overwrite earlier ones, as in the reference
    tail, head = arc[:, 0] - 1, arc[:, 1] - 1
    col = np.arange(n_arcs, dtype=np.int64)
    loop = tail == head
    rows = np.concatenate([tail, head[~loop]])
    cols = np.concatenate([col, col[~loop]])
    vals = np.concatenate([np.ones(n_arcs), -np.ones(int(np.count_nonzero(~loop)))])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n_nodes, n_arcs))
    return MinCostFlow(A=A, b=b, c=arc[:, 4].astype(np.float64), u=arc[:, 3].astype(np.float64), name=name)


def read_idx_images(path: str) -> np.ndarray:
    """An ``idx3-ubyte`` file (magic 0x00000803) as a uint8 array [count, rows, cols]."""
    with open(path, "rb") as fh:
        head = fh.read(16)
        if len(head) < 16:
            raise ValueError(f"{path}: truncated idx header")
        magic, count, rows, cols = struct.unpack(">IIII", head)
        if magic != 0x00000803:
            raise ValueError(f"{path}: not an idx3-ubyte file (magic {magic:#010x})")
        data = np.frombuffer(fh.read(count * rows * cols), dtype=np.uint8)
    if data.size != count * rows * cols:
        raise ValueError(f"{path}: truncated idx payload")
    return data.reshape(count, rows, cols)


def amplify_and_normalise(image: np.ndarray, k: int = 1) -> np.ndarray:
    """Each pixel repeated k x k times, then scaled to total mass 1 (mnist2ot.py:23-27)."""
    big = np.repeat(np.repeat(image.astype(np.float64), k, axis=0), k, axis=1)
    return big / np.sum(big)


def manhattan_cost(side: int) -> np.ndarray:
    """|dy| + |dx| between all pairs of pixels of a side x side grid, row-major pixel order
    (mnist2ot.py:30-40 with side = 28 k)."""
    yy, xx = np.divmod(np.arange(side * side), side)
    return (np.abs(yy[:, None] - yy[None, :]) + np.abs(xx[:, None] - xx[None, :])).astype(np.int64)


def ot_instances_from_images(images: Sequence[np.ndarray], cost: np.ndarray, k: int = 1) -> List[OptTransport]:
    """Pairs (0,1), (2,3), ... of mass-1 images -> OT instances over their non-zero pixels
    (mnist2ot.py:43-61)."""
    out = []
    for i in range(0, len(images) - 1, 2):
        a, b = np.ravel(images[i]), np.ravel(images[i + 1])
        ia, ib = np.flatnonzero(a), np.flatnonzero(b)
        out.append(OptTransport(s=a[ia], d=b[ib], M=cost[np.ix_(ia, ib)], name=f"mnist_{k}_{i // 2 % 10}"))
    return out
