"""CPU restatement of the entropic OT warm start the reference's driver script runs *before* the OT
crossover (scripts/run_network_crossover.py:95-97: ``sinkhorn(ot.s, ot.d, ot.M, reg=10,
numItermax=1000)`` from the third-party package POT, module ``ot``).

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and nothing on the product path).

**Parity unpinned.**  POT is not vendored in the reference, not pinned in its environment.yml and not
installed here, so no golden vector can be produced.  This file restates POT's published algorithm for
``method='sinkhorn'`` (``ot.bregman.sinkhorn_knopp``, unchanged across the 0.7-0.9 releases the
reference could have used): scaling vectors start at 1/dim, ``K = exp(M / (-reg))``,
``Kp = (1/a)[:, None] * K``; per iteration ``v = b / (K^T u)``, ``u = 1 / (Kp v)``; a numerical
breakdown (zero denominator, nan, inf) restores the previous pair and stops; every 10th iteration
(0, 10, 20, ...) the marginal violation ``|| u * (K v-weighted columns) - b ||_2`` is tested against
``stopThr``; the plan is ``(u[:, None] * K) * v[None, :]``.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def sinkhorn_knopp(a: np.ndarray, b: np.ndarray, M: np.ndarray, reg: float, numItermax: int = 1000,
                   stopThr: float = 1e-9) -> Tuple[np.ndarray, dict]:
    """Returns (plan, log) with log = {"iters": completed iterations, "err": last tested violation,
    "u": u, "v": v, "breakdown": bool}."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    M = np.asarray(M, dtype=np.float64)
    dim_a, dim_b = a.shape[0], b.shape[0]
    u = np.ones(dim_a) / dim_a
    v = np.ones(dim_b) / dim_b
    K = np.exp(M / (-reg))
    Kp = (1.0 / a).reshape(-1, 1) * K
    err = 1.0
    iters = 0
    breakdown = False
    with np.errstate(all="ignore"):
        for ii in range(numItermax):
            uprev, vprev = u, v
            KtU = K.T @ u
            v = b / KtU
            u = 1.0 / (Kp @ v)
            if (np.any(KtU == 0) or np.any(np.isnan(u)) or np.any(np.isnan(v)) or np.any(np.isinf(u))
                    or np.any(np.isinf(v))):
                u, v = uprev, vprev
                breakdown = True
                break
            iters = ii + 1
            if ii % 10 == 0:
                tmp2 = np.einsum("i,ij,j->j", u, K, v)
                err = float(np.linalg.norm(tmp2 - b))
                if err < stopThr:
                    break
    plan = u.reshape((-1, 1)) * K * v.reshape((1, -1))
    return plan, {"iters": iters, "err": err, "u": u, "v": v, "breakdown": breakdown}
