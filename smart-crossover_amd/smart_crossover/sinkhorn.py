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


def sinkhorn_batch(a: np.ndarray, b: np.ndarray, M: np.ndarray, reg: float, numItermax: int = 1000,
                   stopThr: float = 1e-9, log: bool = False):
    """The warm starts of several instance pairs that share one cost matrix -- the reference's driver loops over
    ten MNIST image pairs on the same pixel grid (scripts/run_network_crossover.py:95-101) -- in one go:
    ``a`` is (B, S), ``b`` is (B, D), ``M`` is (S, D); a grid point an instance does not use carries mass 0 and
    drops out of that instance exactly as if it had been removed from its support.  Returns the (B, S, D) plans;
    with ``log=True`` also a list of dicts (``u``, ``v``, ``niter``, ``err``) per instance.  Batches of up to 16
    instances run as dense products on the fp64 matrix cores (``sx_sinkhorn_batch_dev``)."""
    from smart_crossover.hip.device import default_context
    a = np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
    b = np.ascontiguousarray(np.atleast_2d(b), dtype=np.float64)
    M = np.ascontiguousarray(M, dtype=np.float64)
    B = a.shape[0]
    if b.shape[0] != B or M.shape != (a.shape[1], b.shape[1]):
        raise ValueError("a must be (B, S), b (B, D) and M (S, D)")
    ctx = default_context()
    S, D = M.shape
    dM = ctx.to_device(M.reshape(-1))
    plans = np.empty((B, S, D))
    logs = []
    for lo in range(0, B, 16):
        nb = min(16, B - lo)
        d_plan, d_u, d_v = ctx.empty(nb * S * D, np.float64), ctx.empty(nb * S, np.float64), ctx.empty(nb * D, np.float64)
        res = ctx.sinkhorn_batch(S, D, nb, ctx.to_device(a[lo:lo + nb].reshape(-1)), ctx.to_device(b[lo:lo + nb].reshape(-1)),
                                 dM, reg, numItermax, stopThr, d_plan, d_u, d_v)
        plans[lo:lo + nb] = d_plan.download().reshape(nb, S, D)
        u, v = d_u.download().reshape(nb, S), d_v.download().reshape(nb, D)
        for k, r in enumerate(res):
            if r.status == 2:
                warnings.warn(f"Warning: numerical errors at iteration {int(r.iters)} (instance {lo + k})")
            logs.append({"u": u[k], "v": v[k], "niter": int(r.iters), "err": float(r.err)})
    return (plans, logs) if log else plans
