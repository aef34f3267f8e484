"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): network-crossover host path.

numpy/scipy restatement of the flow-indicator scoring, ranking, big-M
bookkeeping, pricing test and column-generation schedule of the reference's
``network_methods`` (paths relative to ``/root/reference/src/smart_crossover``).

The MCF indicator is restated per arc / per node on the canonical CSR and CSC
arrays instead of through scipy's sparse ``multiply/maximum/find`` chain; the
rounding order is the same (products rounded separately, node sums taken in
ascending arc order, reciprocal then two multiplies per entry) and the golden
vectors pin it bit for bit.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp

ART_TOL = 1e-8    # TOLERANCE_FOR_ARTIFICIAL_VARS (parameters.py:7)
RC_TOL = 1e-6     # TOLERANCE_FOR_REDUCED_COSTS   (parameters.py:8)
CG_RATIO = 2      # COLUMN_GENERATION_RATIO       (parameters.py:16)


# --------------------------------------------------------------------------
# K7: MCF flow indicators (network_methods/net_manager.py:165-182)
# --------------------------------------------------------------------------
def mcf_signed_residual_flow(x: np.ndarray, u: np.ndarray):
    """x_hat and the large-flow mask (net_manager.py:166-168).

    x_hat = x*(~mask) + u*mask - x*mask with mask = x > u/2, then zeroed where
    x is outside [0, u].  The three-term expression is kept verbatim because
    it decides the rounding (and the NaN the reference produces for u = inf)."""
    with np.errstate(invalid="ignore"):
        mask = x > u / 2
        keep = (~mask).astype(np.float64)
        flip = mask.astype(np.float64)
        x_hat = x * keep + u * flip - x * flip
    x_hat[(x < 0) | (x > u)] = 0
    return x_hat, mask


def mcf_node_throughput(A: sp.csr_matrix, x_hat: np.ndarray, mask: np.ndarray):
    """f_i = max(sum of positive a_bar*x_hat, sum of |negative a_bar|*x_hat) per
    node, sums taken entry by entry in ascending arc order (net_manager.py:169-176);
    a_bar = -a on large-flow arcs.  Returns (f1, f2)."""
    A = sp.csr_matrix(A).copy()
    A.sum_duplicates()
    A.sort_indices()
    m = A.shape[0]
    f1 = np.zeros(m)
    f2 = np.zeros(m)
    sgn = np.where(mask, -1.0, 1.0)
    abar = A.data * sgn[A.indices]
    rows = np.repeat(np.arange(m), np.diff(A.indptr))
    pos = abar > 0
    neg = abar < 0
    # sequential per-row accumulation == CSR matvec of the filtered matrices
    Ap = sp.csr_matrix((abar[pos], (rows[pos], A.indices[pos])), shape=A.shape)
    An = sp.csr_matrix((-abar[neg], (rows[neg], A.indices[neg])), shape=A.shape)
    f1 = Ap @ x_hat
    f2 = An @ x_hat
    return f1, f2


def mcf_flow_indicators(A: sp.csr_matrix, x: np.ndarray, u: np.ndarray):
    """ind_j = max_i | (f_inv_i * x_hat_j) * a_bar_ij |  (net_manager.py:177-182).
    Returns (ind, dict of intermediates)."""
    x_hat, mask = mcf_signed_residual_flow(x, u)
    f1, f2 = mcf_node_throughput(A, x_hat, mask)
    f = np.maximum(f1, f2)
    f_inv = np.zeros_like(f)
    nz = f != 0
    f_inv[nz] = 1 / f[nz]
    C = sp.csc_matrix(A)
    C.sum_duplicates()
    n = C.shape[1]
    cols = np.repeat(np.arange(n), np.diff(C.indptr))
    abar = C.data * np.where(mask, -1.0, 1.0)[cols]
    r = np.abs((f_inv[C.indices] * x_hat[cols]) * abar)
    keep = abar != 0
    ind = np.zeros(n)
    np.maximum.at(ind, cols[keep], r[keep])
    return ind, dict(x_hat=x_hat, mask=mask, f1=f1, f2=f2, f=f, f_inv=f_inv)


# --------------------------------------------------------------------------
# K8: OT flow indicators (net_manager.py:377-379)
# --------------------------------------------------------------------------
def ot_flow_indicators(x: np.ndarray, s: np.ndarray, d: np.ndarray) -> np.ndarray:
    """max(X_ij / s_i, X_ij / d_j), row-major flattening."""
    X = x.reshape(s.size, d.size)
    return np.maximum(X / s[:, None], X / d[None, :]).ravel()


# --------------------------------------------------------------------------
# K9: ranking (net_manager.py:184, :379) and tie classes
# --------------------------------------------------------------------------
def rank_desc(ind: np.ndarray) -> np.ndarray:
    """The build's deterministic ranking: descending key and, inside a run of
    equal keys, descending index -- i.e. a *stable* ascending argsort read
    backwards.  The reference uses numpy's default unstable sort, whose tie
    order is not reproducible; parity is therefore defined on tie classes."""
    return np.argsort(ind, kind="stable")[::-1].astype(np.int64)


def same_up_to_ties(ind: np.ndarray, queue_a: np.ndarray, queue_b: np.ndarray) -> bool:
    """True when both queues are permutations of range(n) that order the keys
    identically (key sequence equal, and each run of equal keys holds the same
    index set)."""
    if queue_a.shape != queue_b.shape:
        return False
    ka, kb = ind[queue_a], ind[queue_b]
    if not np.array_equal(ka, kb):
        return False
    # boundaries of tie classes
    cut = np.flatnonzero(np.diff(ka) != 0) + 1
    start = 0
    for end in list(cut) + [ka.size]:
        if end - start > 1:
            if not np.array_equal(np.sort(queue_a[start:end]), np.sort(queue_b[start:end])):
                return False
        elif queue_a[start] != queue_b[start]:
            return False
        start = end
    return True


# --------------------------------------------------------------------------
# K10: pricing / optimality test
# --------------------------------------------------------------------------
def mcf_reduced_cost(A: sp.csr_matrix, c, y, vbasis) -> np.ndarray:
    """c - A^T y with the sign flipped on columns that sit at their upper
    bound (vbasis == -2)  (net_manager.py:302-303)."""
    rc = c - A.T @ y
    at_up = vbasis == -2
    rc[at_up] = -rc[at_up]
    return rc


def mcf_is_optimal(A, c, y, vbasis, x, artificial) -> bool:
    """net_manager.py:316-319."""
    art_ok = bool(np.all(x[artificial] < ART_TOL)) if len(artificial) else True
    rc_ok = bool(np.all(mcf_reduced_cost(A, c, y, vbasis) >= -RC_TOL))
    return art_ok and rc_ok


def ot_reduced_cost(M: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Reduced cost on the OT incidence structure of formats.py:156-159:
    column (i, j) holds -1 in row i and +1 in row S+j, so
    (A^T y)_ij = (0 + (-1)*y_i) + (+1)*y_{S+j}  (net_manager.py:483)."""
    S, D = M.shape
    aty = (0.0 + (-1.0) * y[:S, None]) + (1.0) * y[None, S:S + D]
    return (M - aty).ravel()


def ot_is_optimal(M, y, x, artificial) -> bool:
    """net_manager.py:495-497 (the last artificial, the corner arc, is exempt)."""
    art_ok = bool(np.all(x[artificial][:-1] < ART_TOL)) if len(artificial) else True
    return art_ok and bool(np.all(ot_reduced_cost(M, y) >= -RC_TOL))


# --------------------------------------------------------------------------
# K11 / K12: cnet_mcf set-up (network_methods/algorithms.py:64-68)
# --------------------------------------------------------------------------
def mcf_initial_partition(x, u):
    """Every arc starts fixed: up when x >= u/2, low otherwise
    (network_methods/algorithms.py:65)."""
    up = np.flatnonzero(x >= u / 2).astype(np.int64)
    low = np.flatnonzero(x < u / 2).astype(np.int64)
    return low, up


def mcf_bigM_extension(A: sp.csr_matrix, b, c_scaled, u, fix_up, bigM) -> Dict[str, object]:
    """One artificial node plus one artificial arc per original node
    (net_manager.py:142-154).  ``u_1`` carries n (not m) infinities -- quirk Q8."""
    m, n = A.shape
    mask_up = np.zeros(n, dtype=bool)
    mask_up[fix_up] = True
    b_true = b - A.multiply(mask_up) @ (u * mask_up)
    sgn = np.sign(b_true)
    sgn[sgn == 0] = 1
    c1 = np.concatenate([c_scaled, bigM * np.ones(m)])
    u1 = np.concatenate([u, np.inf * np.ones(n)])
    top = sp.hstack((A, sp.diags(sgn)))
    A1 = sp.vstack((top, sp.csr_matrix(np.concatenate([np.zeros(n), -sgn])))).tocsr()
    b1 = np.concatenate([b, [0.0]])
    art = np.arange(n, n + m, dtype=np.int64)
    return dict(A=A1, b=b1, c=c1, u=u1, b_true=b_true, b_sign=sgn, artificial=art)


def mcf_initial_basis(n, m, fix_up):
    """Artificials basic, everything else non-basic at a bound
    (net_manager.py:189-192).  Returns (vbasis[n+m], cbasis[m+1])."""
    vb = np.concatenate([-np.ones(n), np.zeros(m)]).astype(int)
    vb[fix_up] = -2
    cb = np.concatenate([-np.ones(m), np.zeros(1)]).astype(int)
    return vb, cb


def mcf_sub_problem(A1: sp.csr_matrix, b1, c1, u1, non_fix, fix_up):
    """net_manager.py:204-209."""
    return dict(A=sp.csr_matrix(A1[:, non_fix]), b=b1 - A1[:, fix_up] @ u1[fix_up], c=c1[non_fix], u=u1[non_fix])


def release_columns(non_fix, fix, fix_low, fix_up, new):
    """add_free_variables (net_manager.py:242-245): append in queue order,
    remove from the three fixed sets."""
    return (np.append(non_fix, new), np.setdiff1d(fix, new), np.setdiff1d(fix_low, new), np.setdiff1d(fix_up, new))


# --------------------------------------------------------------------------
# OT big-M (net_manager.py:388-400) and initial basis (:506-509)
# --------------------------------------------------------------------------
def ot_bigM_extension(s, d, M, bigM):
    S, D = M.shape
    s1 = np.append(s, np.sum(d))
    d1 = np.append(d, np.sum(s))
    M1 = np.empty((S + 1, D + 1))
    M1[:S, :D] = M
    M1[:S, D] = bigM
    M1[S, :D] = bigM
    M1[S, D] = 0
    mask = np.zeros((S + 1, D + 1), dtype=bool)
    mask[:, D] = True
    mask[S, :] = True
    return dict(s=s1, d=d1, M=M1, mask=mask, artificial=np.flatnonzero(mask.ravel()))


def ot_incidence(S: int, D: int) -> sp.csr_matrix:
    """formats.py:154-161: rows 0..S-1 hold -1 on the arcs leaving supplier i,
    rows S..S+D-1 hold +1 on the arcs entering demander j."""
    n = S * D
    arc = np.arange(n)
    rows = np.concatenate([arc // D, S + arc % D])
    cols = np.concatenate([arc, arc])
    vals = np.concatenate([-np.ones(n), np.ones(n)])
    return sp.csr_matrix((vals, (rows, cols)), shape=(S + D, n))


# --------------------------------------------------------------------------
# column-generation pointer schedule (network_methods/algorithms.py:102-136)
# --------------------------------------------------------------------------
def cg_schedule(m: int, n: int, queue_len: int, rounds: int) -> List[Tuple[int, int]]:
    """(left, right) slices of the queue released in each of the first
    ``rounds`` rounds, or fewer when the queue runs out (the reference then
    prints 'Column generation fails!').  ``right`` is an absolute position:
    min(target, len(queue)), target doubling each round."""
    target = int(10 * m) if n / m > 1000 else int(1.2 * m)
    left = 0
    out = []
    for _ in range(rounds):
        if left >= queue_len:
            break
        right = min(target, queue_len)
        out.append((left, right))
        target = int(CG_RATIO * target)
        left = right
    return out
