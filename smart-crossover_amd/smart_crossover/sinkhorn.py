"""Entropic optimal-transport warm start on the device.

The reference's driver produces the inexact plan that TNET / CNET_OT start from with POT:
``sinkhorn(ot.s, ot.d, ot.M, reg=10, numItermax=1000)`` (scripts/run_network_crossover.py:95-97).  This
module offers the same call -- same argument names and defaults as ``ot.sinkhorn`` for the arguments the
reference uses -- computed by ``sx_sinkhorn_dev`` (csrc/sx_sinkhorn.hip).  POT is a third-party package
that is neither vendored nor pinned by the reference, so parity is unpinned; the tests compare with a
restatement of POT's published ``sinkhorn_knopp`` (oracle/sinkhorn.py).
"""
from __future__ import annotations

import warnings
from typing import Tuple, Union

import numpy as np


def sinkhorn(a: np.ndarray, b: np.ndarray, M: np.ndarray, reg: float, numItermax: int = 1000, stopThr: float = 1e-9,
             log: bool = False) -> Union[np.ndarray, Tuple[np.ndarray, dict]]:
    """Sinkhorn-Knopp plan ``diag(u) exp(-M/reg) diag(v)`` for marginals ``a`` (sources) and ``b``
    (targets); with ``log=True`` also a dict with ``u``, ``v``, ``niter`` and the last tested ``err``."""
    from smart_crossover.hip.device import default_context
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    M = np.ascontiguousarray(M, dtype=np.float64)
    if M.shape != (a.size, b.size):
        raise ValueError("M must have shape (len(a), len(b))")
    ctx = default_context()
    S, D = M.shape
    plan, u, v = ctx.empty(S * D, np.float64), ctx.empty(S, np.float64), ctx.empty(D, np.float64)
    res = ctx.sinkhorn(S, D, ctx.to_device(a), ctx.to_device(b), ctx.to_device(M.reshape(-1)), reg, numItermax, stopThr,
                       plan, u, v)
    if res.status == 2:
        warnings.warn(f"Warning: numerical errors at iteration {int(res.iters)}")
    out = plan.download().reshape(S, D)
    if log:
        return out, {"u": u.download(), "v": v.download(), "niter": int(res.iters), "err": float(res.err)}
    return out
