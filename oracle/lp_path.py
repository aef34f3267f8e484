"""Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): perturbation-crossover host path.

numpy/scipy restatement, as pure functions on arrays, of the reference's
``get_perturb_problem`` pipeline and its helpers.  Every function cites the
reference lines it follows (paths relative to
``/root/reference/src/smart_crossover``).  All arithmetic is fp64, index sets
are int64, exactly as in the reference.

Rounding-order contract, relied on by the HIP kernels:
  * ``c - A^T y`` : per column, products a_ij*y_i are rounded separately and
    added to a running sum that starts at +0.0, in the order the entries of
    the column appear when the CSR matrix is walked row by row
    (= stored order of the stable CSR->CSC transposition).
  * ``b - A x``   : per row, same rule in stored CSR order.
``seq_segment_sums`` states this with explicit Python loops; the tests pin
scipy's kernels (and hence the reference) to it.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np
import scipy.sparse as sp

# constants, lp side (parameters.py:22-28)
GAMMA0 = 1e-3              # OPTIMAL_FACE_ESTIMATOR
GAMMA_UPDATE = 1e-5        # OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO
X_FLOOR = 1e-6             # PERTURB_THRESHOLD (and the literal 1e-6 at algorithms.py:131)
SF_DIVISOR = 1e-2          # CONSTANT_SCALE_FACTOR
GAP_TOL = 1e-8             # PRIMAL_DUAL_GAP_THRESHOLD
PROJ_TOL = 1e-8            # PROJECTOR_THRESHOLD
P_CAP = 1e6                # PERTURB_UPPER_BOUND

CODE_LOW = 1   # bit 0: column goes to its lower bound
CODE_UP = 2    # bit 1: column goes to its upper bound


# --------------------------------------------------------------------------
# K1 / K2: slacks and indicator tests
# --------------------------------------------------------------------------
def dual_slack(A: sp.csr_matrix, c: np.ndarray, y: np.ndarray) -> np.ndarray:
    """s_d = c - A^T y  (formats.py:70-72, called at lp_methods/algorithms.py:99).

    ``A.transpose()`` of a CSR matrix is a CSC view, so scipy runs its
    ``csc_matvec`` scatter loop: row-major walk, one rounded multiply and one
    rounded add per stored entry."""
    return c - A.transpose() @ y


def primal_slack(A: sp.csr_matrix, b: np.ndarray, x: np.ndarray) -> np.ndarray:
    """s_p = b - A x  (formats.py:74-76, called at lp_methods/algorithms.py:100)."""
    return b - A @ x


def seq_segment_sums(ptr: np.ndarray, idx: np.ndarray, val: np.ndarray, vec: np.ndarray) -> np.ndarray:
    """Explicit-loop statement of the rounding order: for every segment
    ``s`` (a CSR row or a CSC column) ``out[s] = (((0 + v0*w0) + v1*w1) + ...)``
    with each product and each sum rounded to fp64.  Small inputs only."""
    nseg = len(ptr) - 1
    out = np.zeros(nseg, dtype=np.float64)
    for s in range(nseg):
        acc = np.float64(0.0)
        for k in range(int(ptr[s]), int(ptr[s + 1])):
            prod = np.float64(val[k]) * np.float64(vec[idx[k]])
            acc = acc + prod
        out[s] = acc
    return out


def csc_in_walk_order(A: sp.csr_matrix) -> sp.csc_matrix:
    """CSC arrays whose per-column entry order equals the order in which a
    row-major walk of ``A`` (CSR) meets them -- the order scipy's scatter loop
    and therefore the reference accumulate in.  ``tocsc`` is a stable counting
    sort by column, which is exactly that order (duplicates are kept)."""
    A = sp.csr_matrix(A)
    return A.tocsc(copy=True)


def column_codes(x, l, u, s_d, gamma) -> np.ndarray:
    """Primal-dual indicator of the columns (lp_methods/algorithms.py:104-105).

    bit0 set  <=>  x - l < gamma * s_d      (fix to lower bound)
    bit1 set  <=>  u - x < gamma * (-s_d)   (fix to upper bound)"""
    low = (x - l) < (gamma * s_d)
    up = (u - x) < (gamma * -s_d)
    return (low.astype(np.uint8) * CODE_LOW) | (up.astype(np.uint8) * CODE_UP)


def row_flags(s_p, y, gamma_dual) -> np.ndarray:
    """Rows whose slack is dominated by the dual: s_p < gamma_dual * (-y)
    (lp_methods/algorithms.py:106)."""
    return (s_p < (gamma_dual * -y)).astype(np.uint8)


def index_sets(code: np.ndarray, rowflag: np.ndarray):
    """int64 index arrays in the form the reference feeds to its manager."""
    fix_low = np.flatnonzero(code & CODE_LOW).astype(np.int64)
    fix_up = np.flatnonzero(code & CODE_UP).astype(np.int64)
    fixed_rows = np.flatnonzero(rowflag).astype(np.int64)
    return fix_low, fix_up, fixed_rows


# --------------------------------------------------------------------------
# K3: perturbed cost
# --------------------------------------------------------------------------
def free_index(l, u) -> np.ndarray:
    """formats.py:30-32."""
    return np.flatnonzero((l == -np.inf) & (u == np.inf))


def x_perturb_val(x, l, u) -> np.ndarray:
    """Distance to the nearer bound, free columns keep x; then the floor and
    the free->1 overwrite (lp_methods/algorithms.py:196-202 followed by :130-132).
    Order matters: the floor is applied while free entries still hold x."""
    free = free_index(l, u)
    xr = np.minimum(x - l, u - x)
    xr[free] = x[free]
    xr[xr < X_FLOOR] = 1e-6
    xr[free] = 1.0
    return xr


def xi_raw(n: int) -> np.ndarray:
    """The unnormalised direction: n draws U(0.9, 1) from the legacy MT19937
    stream seeded with 42 (lp_methods/algorithms.py:135-136).  The legacy stream is
    frozen by NEP 19, so ``RandomState(42)`` reproduces ``np.random.seed(42)``
    without touching the caller's global generator."""
    return np.random.RandomState(42).uniform(0.9, 1, n)


def xi_vector(n: int) -> np.ndarray:
    """xi / ||xi||_2  (lp_methods/algorithms.py:137)."""
    p = xi_raw(n)
    return p / np.linalg.norm(p)


def perturb_cost(c, x, l, u, xi, scale_factor: Optional[float], is_feas: bool) -> np.ndarray:
    """c_pt (lp_methods/algorithms.py:139-151) given the normalised direction ``xi``
    and, for non-feasibility problems, the projector scale factor."""
    if is_feas:
        return c + xi
    xr = x_perturb_val(x, l, u)
    p = np.minimum(xi / xr * scale_factor / SF_DIVISOR, P_CAP)
    p[free_index(l, u)] = 0
    return c + p


# --------------------------------------------------------------------------
# standard form pieces (formats.py:46-68)
# --------------------------------------------------------------------------
def slack_rows(sense) -> np.ndarray:
    return np.flatnonzero(sense == "<")


def standard_A(A: sp.csr_matrix, sense) -> sp.csr_matrix:
    """[A, I[:, rows with '<']]  (formats.py:46-50)."""
    m = A.shape[0]
    eye = sp.eye(m, format="csr")[:, slack_rows(sense)]
    return sp.hstack([A, eye]).tocsr()


def standard_c(c, sense) -> np.ndarray:
    """formats.py:52-54."""
    return np.concatenate([c, np.zeros(int(np.sum(sense == "<")))])


def standard_x(A, b, sense, x) -> np.ndarray:
    """[x, b_< - A_< x]  (formats.py:56-68)."""
    rows = slack_rows(sense)
    return np.concatenate([x, b[rows] - A[rows, :] @ x])


# --------------------------------------------------------------------------
# K4: projector norm / scale factor
# --------------------------------------------------------------------------
def cg_legacy(matvec: Callable[[np.ndarray], np.ndarray], b: np.ndarray, tol: float = 1e-8,
              maxiter: int = 1000) -> Tuple[np.ndarray, int, bool]:
    """Unpreconditioned CG from x0 = 0 with the stopping rule the reference's
    pinned scipy (1.7.3, ``cg(tol=..., atol=None)``, lp_methods/algorithms.py:186)
    applies: return x0 at once when ||b|| <= tol (absolute); otherwise stop as
    soon as ||r|| < tol*||b||, tested before each iteration; give up after
    ``maxiter`` iterations.  The recurrence is scipy's (rho/beta/p/q/alpha).
    Returns (z, iterations done, converged)."""
    xk = np.zeros_like(b)
    r = b.copy()
    bnrm = float(np.linalg.norm(b))
    if bnrm <= tol:
        return xk, 0, True
    atol = tol * bnrm
    rho_prev = None
    p = None
    for it in range(maxiter):
        if np.linalg.norm(r) < atol:
            return xk, it, True
        rho = np.dot(r, r)
        if it > 0:
            p *= rho / rho_prev
            p += r
        else:
            p = r.copy()
        q = matvec(p)
        alpha = rho / np.dot(p, q)
        xk += alpha * p
        r -= alpha * q
        rho_prev = rho
    return xk, maxiter, False


def projector_explicit(Y: sp.spmatrix, v: np.ndarray, tol: float = 1e-8, maxiter: int = 1000):
    """(I - Y^T (Y Y^T)^+ Y) v the way the reference does it: form Y Y^T
    explicitly, CG on it (lp_methods/algorithms.py:183-187).  Returns (proj, iters)."""
    Yv = Y @ v
    G = (Y @ Y.T).tocsr()
    z, iters, _ = cg_legacy(lambda p: G @ p, Yv, tol, maxiter)
    return v - Y.T @ z, iters


def projector_matrix_free(A: sp.csr_matrix, sense, xx: np.ndarray, c_std: np.ndarray,
                          tol: float = 1e-8, maxiter: int = 1000):
    """Same projection without forming Y Y^T: Y = [A, I_<] diag(xx);
    (Y Y^T) p = A (xx_A^2 * (A^T p)) + [row is '<'] xx_s^2 * p.
    This is the algorithm the HIP path runs (kernel K4); the CPU
    statement lives here so the device result can be compared like for like.
    Returns (proj, iters)."""
    m, n = A.shape
    rows = slack_rows(sense)
    xa, xs = xx[:n], xx[n:]
    v = xx * c_std
    va, vs = v[:n], v[n:]
    AT = A.transpose().tocsr()

    def Yt(p):
        return np.concatenate([xa * (AT @ p), xs * p[rows]])

    def Ymul(w):
        out = A @ (xa * w[:n])
        out[rows] += xs * w[n:]
        return out

    Yv = Ymul(v)
    z, iters, _ = cg_legacy(lambda p: Ymul(Yt(p)), Yv, tol, maxiter)
    return v - Yt(z), iters


def scale_factor_from_projection(proj: np.ndarray, n_std: int) -> float:
    """||proj||_2 / (n + #'<')  (lp_methods/algorithms.py:144,190-193)."""
    return float(np.linalg.norm(proj) / n_std)


def projector_qp_exact(Y, v: np.ndarray, A_f=None) -> np.ndarray:
    """What apply_projector_qp defines (lp_methods/algorithms.py:240-265):
    argmin ||x - v||^2  s.t.  Y x (+ A_f f) = 0, stated through its KKT system and solved densely by
    least squares.  Small inputs only.  **Parity unpinned**: the reference solves this QP inside Gurobi
    (BarQCPConvTol = 1e-1), which cannot run here; this is the exact minimiser of the same problem."""
    Y = np.asarray(Y.todense()) if sp.issparse(Y) else np.asarray(Y, dtype=float)
    m = Y.shape[0]
    if A_f is None:
        K, rhs = Y @ Y.T, Y @ v
    else:
        F = np.asarray(A_f.todense()) if sp.issparse(A_f) else np.asarray(A_f, dtype=float)
        nf = F.shape[1]
        K = np.block([[Y @ Y.T, F], [F.T, np.zeros((nf, nf))]])
        rhs = np.concatenate([Y @ v, np.zeros(nf)])
    lam = np.linalg.lstsq(K, rhs, rcond=None)[0][:m]
    return v - Y.T @ lam


def projector_Xc_free(A, b, c, l, u, sense, x_real):
    """Free-variable branch of get_projector_Xc (lp_methods/algorithms.py:173-180): the cost of the
    free columns is first removed by a least-squares step (cg on A_2^T A_2 there; solved densely here),
    then X_1 c' is projected with the free columns as unweighted extra unknowns.  Returns the projection
    over the non-free standard columns (structural ones in order, then the slack columns)."""
    free = free_index(l, u)
    n = A.shape[1]
    xx = standard_x(A, b, sense, x_real)
    c_std = standard_c(c, sense)
    A_std = sp.csr_matrix(standard_A(A, sense))
    nonfree = np.setdiff1d(np.arange(A_std.shape[1]), free)          # formats.py:34-36
    A1, A2 = A_std[:, nonfree], sp.csr_matrix(A)[:, free]
    G = (A2.T @ A2).toarray()
    trans = np.linalg.lstsq(G, c_std[free], rcond=None)[0]
    c_nonfree = c_std[nonfree] - A1.T @ (A2 @ trans)
    assert nonfree.size == n - free.size + int(np.count_nonzero(np.asarray(sense) == "<"))
    return projector_qp_exact(A1 @ sp.diags(xx[nonfree]), xx[nonfree] * c_nonfree, A2)


def projector_Xc(A, b, c, l, u, sense, x_real, tol=1e-8, maxiter=1000, explicit=True):
    """lp_methods/algorithms.py:162-180 (quirk Q4: the slack block is built from the clipped
    ``x_real``).  With free variables the reference goes through a Gurobi QP: see
    ``projector_Xc_free`` (exact minimiser, parity unpinned)."""
    if free_index(l, u).size:
        return projector_Xc_free(A, b, c, l, u, sense, x_real), 0
    xx = standard_x(A, b, sense, x_real)
    c_std = standard_c(c, sense)
    if explicit:
        Y = standard_A(A, sense) @ sp.diags(xx)
        return projector_explicit(Y, sp.diags(xx) @ c_std, tol, maxiter)
    return projector_matrix_free(A, sense, xx, c_std, tol, maxiter)


def perturbed_cost_full(A, b, c, l, u, sense, x, is_feas, explicit=True):
    """perturb_c end to end (lp_methods/algorithms.py:114-151).
    Returns (c_pt, dict of intermediates)."""
    n = len(x)
    xi = xi_vector(n)
    if is_feas:
        return c + xi, {"xi": xi}
    xr = x_perturb_val(x, l, u)
    proj, iters = projector_Xc(A, b, c, l, u, sense, xr, explicit=explicit)
    sf = scale_factor_from_projection(proj, n + int(np.count_nonzero(sense == "<")))
    c_pt = perturb_cost(c, x, l, u, xi, sf, False)
    return c_pt, {"xi": xi, "x_real": xr, "proj_norm": float(np.linalg.norm(proj)), "sf": sf, "cg_iters": iters}


# --------------------------------------------------------------------------
# K6: sub-problem bookkeeping (lp_manager.py)
# --------------------------------------------------------------------------
def fix_partition(n: int, fix_low: np.ndarray, fix_up: np.ndarray):
    """non_fix / fix as sorted unique int64 arrays (lp_manager.py:40-50)."""
    fixed = np.union1d(fix_low, fix_up).astype(np.int64)
    keep = np.ones(n, dtype=bool)
    keep[fixed] = False
    return np.flatnonzero(keep).astype(np.int64), fixed


def sub_problem(A: sp.csr_matrix, b, c, l, u, sense, fix_low, fix_up, fixed_rows) -> Dict[str, object]:
    """Restricted LP after fixing (lp_manager.py:52-66): column slice of A,
    right-hand side moved by the fixed columns at their bounds (upper bounds
    first, then lower bounds, two separate subtractions), gathered c/l/u, and
    the sense of dual-dominated rows turned into '='."""
    n = A.shape[1]
    non_fix, fixed = fix_partition(n, fix_low, fix_up)
    sense_sub = sense.copy()
    if fixed.size == 0:
        A_sub, b_sub, c_sub, l_sub, u_sub = A, b, c, l, u
    else:
        A_sub = A[:, non_fix]
        b_sub = b - A[:, fix_up] @ u[fix_up] - A[:, fix_low] @ l[fix_low]
        c_sub, l_sub, u_sub = c[non_fix], l[non_fix], u[non_fix]
    if fixed_rows.size:
        sense_sub[fixed_rows] = "="
    return dict(non_fix=non_fix, fix=fixed, A=sp.csr_matrix(A_sub), b=b_sub, c=c_sub, l=l_sub, u=u_sub,
                sense=sense_sub)


def recover_x(n, non_fix, fix_up, u, x_sub) -> np.ndarray:
    """lp_manager.py:68-77 -- quirk Q1: columns fixed low are left at 0, not l."""
    x = np.zeros(n)
    x[non_fix] = x_sub
    x[fix_up] = u[fix_up]
    return x


def original_x(n, non_fix, fix_low, fix_up, l, u, x_sub) -> np.ndarray:
    """lp_manager.py:99-109 (both bounds restored)."""
    x = recover_x(n, non_fix, fix_up, u, x_sub)
    x[fix_low] = l[fix_low]
    return x


def recover_vbasis(n, non_fix, fix_up, vbasis_sub) -> np.ndarray:
    """lp_manager.py:79-89: default -1, kept columns take the sub-basis,
    columns fixed up get -2; cbasis passes through unchanged."""
    vb = -np.ones(n, dtype=int)
    vb[non_fix] = vbasis_sub
    vb[fix_up] = -2
    return vb


def relative_gap(c_ori, x_full, barrier_obj) -> float:
    """lp_methods/algorithms.py:217-220."""
    mine = float(c_ori @ x_full)
    return abs(mine - barrier_obj) / (abs(mine) + abs(barrier_obj) + 1)


def gap_ok(c_ori, x_full, barrier_obj):
    """True or None -- never False (quirk Q3, lp_methods/algorithms.py:223-224)."""
    return True if relative_gap(c_ori, x_full, barrier_obj) < GAP_TOL else None


# --------------------------------------------------------------------------
# whole scoring pass (what bench.py times as the CPU baseline)
# --------------------------------------------------------------------------
def scoring_pass(A: sp.csr_matrix, b, c, l, u, x, y, gamma=GAMMA0, gamma_dual=GAMMA0):
    """K1 + K2 + index sets exactly as get_perturb_problem does them
    (lp_methods/algorithms.py:99-106)."""
    s_d = dual_slack(A, c, y)
    s_p = primal_slack(A, b, x)
    code = column_codes(x, l, u, s_d, gamma)
    rf = row_flags(s_p, y, gamma_dual)
    fix_low, fix_up, fixed_rows = index_sets(code, rf)
    return dict(s_d=s_d, s_p=s_p, code=code, rowflag=rf, fix_low=fix_low, fix_up=fix_up, fixed_rows=fixed_rows)
