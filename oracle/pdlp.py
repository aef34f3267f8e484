"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the first-order stage of the LP re-solve.

CPU statement, in numpy, of what csrc/sx_pdlp.hip runs on the device (kernel group K16p): restarted, diagonally
preconditioned primal-dual hybrid gradient for  min c^T x, A x (= | <=) b, l <= x <= u.  It stands where the
reference's backends run their barrier before the crossover of the perturbed sub-problem
(lp_methods/algorithms.py:50-54 -> solver_caller/gurobi.py:111-115); that arithmetic is inside Gurobi, i.e.
PARITY UNPINNED -- this file is the algorithm's own statement (PDLP: Applegate et al., NeurIPS 2021), checked
against HiGHS' optimal value in tests/test_oracle_pdlp.py; the device is compared with it step for step.

Sign convention: reduced cost = c - A^T y, dual of a '<' row <= 0.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp

PERIOD = 64


def scalings(A: sp.csr_matrix, sweeps: int = 10):
    """Row / column scalings: ``sweeps`` of Ruiz (divide by the square root of the max norm), then one of
    Pock-Chambolle (divide by the square root of the 1-norm).  Empty rows / columns keep their scale."""
    A = sp.csr_matrix(A)
    absA = abs(A)
    m, n = A.shape
    dr, dc = np.ones(m), np.ones(n)
    for sweep in range(sweeps + 1):
        S = sp.diags(dr) @ absA @ sp.diags(dc)
        if sweep < sweeps:
            rn = np.asarray(S.max(axis=1).todense()).ravel()
            cn = np.asarray(S.max(axis=0).todense()).ravel()
        else:
            rn = np.asarray(S.sum(axis=1)).ravel()
            cn = np.asarray(S.sum(axis=0)).ravel()
        dr = np.where(rn > 0, dr / np.sqrt(np.where(rn > 0, rn, 1.0)), dr)
        dc = np.where(cn > 0, dc / np.sqrt(np.where(cn > 0, cn, 1.0)), dc)
    return dr, dc


def operator_norm(A: sp.csr_matrix, dr, dc, iters: int = 40) -> float:
    """sigma_max(D_r A D_c) by power iteration from the all-ones vector (the device's start)."""
    M = sp.diags(dr) @ sp.csr_matrix(A) @ sp.diags(dc)
    u = np.ones(A.shape[1])
    lam = 0.0
    for _ in range(iters):
        t = M.T @ (M @ u)
        lam = float(np.linalg.norm(t))
        if lam == 0.0:
            return 1.0
        u = t / lam
    return float(np.sqrt(lam))


def kkt(A, AT, b, c, l, u, lt, x, y):
    r = b - A @ x
    r = np.where(lt, np.minimum(r, 0.0), r)
    rc = c - AT @ y
    viol = np.where(rc > 0, np.where(np.isinf(l), rc, 0.0), np.where(rc < 0, np.where(np.isinf(u), rc, 0.0), 0.0))
    with np.errstate(invalid="ignore"):
        bound = np.where(rc > 0, np.where(np.isinf(l), 0.0, l * rc), np.where(rc < 0, np.where(np.isinf(u), 0.0, u * rc), 0.0))
    pobj = float(c @ x)
    dobj = float(b @ y) + float(bound.sum())
    pr, du, gap = float(np.linalg.norm(r)), float(np.linalg.norm(viol)), abs(pobj - dobj)
    return {"pr": pr, "du": du, "gap": gap, "pobj": pobj, "dobj": dobj, "err": float(np.sqrt(pr * pr + du * du + gap * gap)),
            "rcnorm": float(np.linalg.norm(rc))}


def pdlp(A, b, c, l, u, row_is_lt, x0: Optional[np.ndarray] = None, y0: Optional[np.ndarray] = None,
         max_iter: int = 20000, tol: float = 1e-8) -> Dict[str, object]:
    """The iteration of sx_pdlp_dev, same constants, same order of decisions.  Returns x, y and the record the
    device fills (status 0 converged / 3 iteration limit, iters, restarts, residuals, step, primal weight)."""
    A = sp.csr_matrix(A)
    AT = A.T.tocsr()
    m, n = A.shape
    lt = np.asarray(row_is_lt, dtype=bool)
    b, c, l, u = (np.asarray(v, dtype=np.float64) for v in (b, c, l, u))
    dr, dc = scalings(A)
    dr2, dc2 = dr * dr, dc * dc
    eta = 0.9 / (1.02 * operator_norm(A, dr, dc))
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    y = np.zeros(m) if y0 is None else np.array(y0, dtype=np.float64)
    max_iter = (max(int(max_iter), 1) + PERIOD - 1) // PERIOD * PERIOD
    bnorm, cnorm = float(np.linalg.norm(b)), float(np.linalg.norm(c))
    k0 = kkt(A, AT, b, c, l, u, lt, x, y)
    omega = 1.0
    if k0["rcnorm"] > 0 and bnorm > 0:
        omega = k0["rcnorm"] / bnorm
    omega = min(max(omega, 1e-8), 1e8)
    err_restart = err_prev = k0["err"]
    xr, yr = x.copy(), y.copy()
    xsum, ysum = np.zeros(n), np.zeros(m)
    k_since = total = restarts = 0
    status, last = 0, k0
    while status == 0:
        for _ in range(PERIOD):
            tau, sig = (eta / omega) * dc2, (eta * omega) * dr2
            xn = np.minimum(np.maximum(x - tau * (c - AT @ y), l), u)
            yn = y + sig * (b - A @ (2.0 * xn - x))
            yn = np.where(lt, np.minimum(yn, 0.0), yn)
            x, y = xn, yn
            xsum += x
            ysum += y
        k_since += PERIOD
        total += PERIOD
        xa, ya = xsum / k_since, ysum / k_since
        e0, e1 = kkt(A, AT, b, c, l, u, lt, x, y), kkt(A, AT, b, c, l, u, lt, xa, ya)
        cand_avg = e1["err"] < e0["err"]
        e = e1 if cand_avg else e0
        go = False
        if e["err"] <= 0.2 * err_restart:
            go = True
        elif e["err"] <= 0.8 * err_restart and e["err"] > err_prev:
            go = True
        elif k_since >= 0.36 * total and total > 1000:
            go = True
        err_prev = e["err"]
        conv = (e["pr"] <= tol * (1 + bnorm) and e["du"] <= tol * (1 + cnorm)
                and e["gap"] <= tol * (1 + abs(e["pobj"]) + abs(e["dobj"])))
        if conv:
            status = 1
        elif total >= max_iter:
            status = 2
        if status:
            go = True
        last = e
        if go:
            # (an average of points of the box can leave it by a rounding: put it back)
            xc, yc = (np.minimum(np.maximum(xa, l), u), np.where(lt, np.minimum(ya, 0.0), ya)) if cand_avg else (x, y)
            dx, dy = float(np.linalg.norm(xc - xr)), float(np.linalg.norm(yc - yr))
            if dx > 1e-300 and dy > 1e-300 and not conv:
                omega = float(np.exp(0.5 * np.log(dy / dx) + 0.5 * np.log(omega)))
            omega = min(max(omega, 1e-10), 1e10)
            x, y = xc.copy(), yc.copy()
            xr, yr = x.copy(), y.copy()
            xsum[:] = 0.0
            ysum[:] = 0.0
            k_since = 0
            restarts += 1
            err_restart = e["err"]
    return {"x": x, "y": y, "status": 0 if status == 1 else 3, "iters": total, "restarts": restarts,
            "primal_residual": last["pr"], "dual_residual": last["du"], "gap": last["gap"],
            "primal_obj": last["pobj"], "dual_obj": last["dobj"], "step": eta, "primal_weight": omega}
