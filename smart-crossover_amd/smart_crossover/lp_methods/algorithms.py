"""Perturbation crossover for general LPs -- API of the reference's ``lp_methods/algorithms.py``
(run_perturb_algorithm :18, get_perturb_problem :79, perturb_c :114, get_projector_c :154,
get_projector_Xc :162, apply_projector :183, get_scale_factor :190, get_x_perturb_val :196,
check_perturb_output_precision :205, check_feasibility_problem :227, apply_projector_qp :240).

Everything between the solver calls runs on the MI355X through libsxhip.so:

    K1  column scoring      s_d = c - A^T y, column codes          sx_score_columns_dev
    K2  row scoring         s_p = b - A x, row flags               sx_score_rows_dev
        index sets          np.where of the three tests            sx_select_indices_dev
    K3  perturbed cost      c + min(xi/x_real*sf/1e-2, 1e6)        sx_x_real_dev, sx_perturb_cost_dev
    K4  projector           ||(I - Y^T (YY^T)^+ Y) X c||, matrix-free CG   sx_projector_dev
    K6  sub-problem         A[:, non_fix], right-hand side, gathers   (LPManager.update_subproblem)

The LP re-solves go through the ``SolverCaller`` seam exactly as in the reference (``solver=``).
Printed messages are the reference's (its log scrapers key on them).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import GeneralLP
from smart_crossover.lp_methods.lp_manager import LPManager
from smart_crossover.output import Output
from smart_crossover.parameters import (CONSTANT_SCALE_FACTOR, OPTIMAL_FACE_ESTIMATOR,
                                        OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO, PERTURB_THRESHOLD,
                                        PERTURB_UPPER_BOUND, PRIMAL_DUAL_GAP_THRESHOLD, PROJECTOR_THRESHOLD)
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.solver_caller.solving import solve_lp

# the device kernels hard-code these three constants (include/sxhip.h, K3)
assert (PERTURB_THRESHOLD, CONSTANT_SCALE_FACTOR, PERTURB_UPPER_BOUND) == (1e-6, 1e-2, 1e6)


# ------------------------------------------------------------------------------------------------
# entry point
# ------------------------------------------------------------------------------------------------
def run_perturb_algorithm(lp: GeneralLP, solver: str = "GRB", barrierTol: float = 1e-8,
                          optimalityTol: float = 1e-6, log_file: str = "") -> Output:
    """Interior point -> primal-dual indicator fixing -> perturbed objective -> re-solve (with a
    widening retry) -> gap check -> warm-started primal simplex on the original LP if needed."""
    print("*** Running the perturbation crossover algorithm... ***")
    barrier_output = solve_lp(lp, solver, method="barrier",
                              settings=SolverSettings(barrierTol=barrierTol, presolve="on", crossover="off",
                                                      log_file=log_file))
    is_feas_problem = check_feasibility_problem(lp)

    gamma = gamma_dual = OPTIMAL_FACE_ESTIMATOR
    while True:
        print("*** Getting and solving a perturb subproblem... ***")
        manager = get_perturb_problem(lp, barrier_output.x, barrier_output.y, gamma, gamma_dual, is_feas=is_feas_problem)
        # NB a 'barrier' run ignores the warm start (quirk Q6); it is passed because the reference passes it
        perturb_output = solve_lp(manager.lp_sub, solver=solver, method="barrier",
                                  settings=SolverSettings(presolve="on", log_file=log_file),
                                  warm_start_solution=(manager.get_subx(barrier_output.x), barrier_output.y))
        if perturb_output.status in ("INFEASIBLE", "UNBOUNDED"):
            gamma *= OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO
            gamma_dual *= OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO ** 2
            print("*** The perturbation is infeasible or unbounded. Increasing the optimal face and try again... ***")
            continue
        break

    if check_perturb_output_precision(manager, perturb_output.x, lp.c, barrier_output.obj_val):
        print("*** A primal optimal BFS is found. ***")
        return perturb_output            # quirk Q2: sub-problem sized x, perturbed objective

    return solve_lp(lp, solver=solver, method="simplex" if solver == "MSK" else "primal_simplex",
                    settings=SolverSettings(presolve="on", optimalityTol=optimalityTol, log_file=log_file),
                    warm_start_solution=(manager.recover_x_from_sub_x(perturb_output.x), perturb_output.y),
                    warm_start_basis=manager.recover_basis_from_sub_basis(perturb_output.basis))


# ------------------------------------------------------------------------------------------------
# device pipeline
# ------------------------------------------------------------------------------------------------
class _Resident:
    """Device handles shared by the steps of one get_perturb_problem call."""

    def __init__(self, lp: GeneralLP, x: Optional[np.ndarray] = None, y: Optional[np.ndarray] = None):
        from smart_crossover.hip.resident import resident_for
        self.res = resident_for(lp)
        self.ctx = self.res.ctx
        self.m, self.n = self.res.m, self.res.n
        put = self.res.put
        self.b, self.c, self.l, self.u = put(lp.b), put(lp.c), put(lp.l), put(lp.u)
        self.x = put(x) if x is not None else None
        self.y = put(y) if y is not None else None


_XI_CACHE: Dict[int, Tuple[np.ndarray, tuple]] = {}


def _perturb_direction(n: int) -> np.ndarray:
    """xi / ||xi||: n draws U(0.9, 1) of numpy's *global* legacy generator re-seeded with 42
    (lp_methods/algorithms.py:135-137).  The side effect on the caller's global generator (quirk Q5:
    state = seed 42 advanced by n draws) is reproduced, also on cache hits."""
    hit = _XI_CACHE.get(n)
    if hit is not None:
        np.random.seed(42)
        np.random.set_state(hit[1])
        return hit[0]
    np.random.seed(42)
    p = np.random.uniform(0.9, 1, n)
    p = p / np.linalg.norm(p)
    if len(_XI_CACHE) > 4:
        _XI_CACHE.clear()
    _XI_CACHE[n] = (p, np.random.get_state())
    return p


def _projector_device(dv: _Resident, xa, xs, c, tol: float, max_iter: int, want_vector: bool):
    ctx = dv.ctx
    pc = ctx.empty(dv.n, np.float64) if want_vector else None
    pr = ctx.empty(dv.m, np.float64) if want_vector else None
    info = ctx.projector_norm(dv.res.A, xa, xs, c, tol, max_iter, pc, pr)
    return info, pc, pr


def _projector_free_device(lp: GeneralLP, dv: _Resident, x_real, xs, want_vector: bool):
    """Free-variable branch of get_projector_Xc (lp_methods/algorithms.py:173-180) on the device
    (``sx_projector_free_dev``, csrc/sx_cg.hip).

    Reference: t = cg(A_2^T A_2, c_free); c' = c_std[nonfree] - A_1^T A_2 t; then the Gurobi QP
    min ||x - X_1 c'||^2 s.t. A_1 X_1 x + A_2 f = 0.  The library solves the normal equations of the free block
    with its CG kernels, forms the adjusted cost with one column pass (K1) and takes the QP as the limit of the
    ordinary projector in which the free columns carry a large scale tau and zero cost (penalty form).  Parity
    with the reference is unpinned (no Gurobi); tests compare with the exact minimiser of the QP
    (oracle.projector_Xc_free)."""
    ctx, m, n = dv.ctx, dv.m, dv.n
    free = lp.get_free_ind()
    pc, pr = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    info = ctx.projector_free(dv.res.A, ctx.to_device(np.ascontiguousarray(free, dtype=np.int64)), x_real, xs, dv.c,
                              dv.res.lt, pc, pr)
    proj = None
    if want_vector:
        # the free block of the penalised projection is f / tau, not part of the QP's x: it is left out
        proj = np.concatenate([np.delete(pc.download(), free), pr.download()[np.asarray(lp.sense) == "<"]])
    return info, proj, None


def _scaled_slack(dv: _Resident, x_real_dev):
    """Slack block of the standard-form vector built from x_real (quirk Q4): b_< - A_< x_real on the
    '<' rows, 0 on the '=' rows."""
    ctx = dv.ctx
    s_p = ctx.empty(dv.m, np.float64)
    ctx.score_rows(dv.res.A, x_real_dev, dv.b, None, 0.0, s_p, None)
    xs = ctx.empty(dv.m, np.float64)
    ctx.mask(s_p, dv.res.lt, xs)
    return xs


def _perturb_c_device(lp: GeneralLP, dv: _Resident, is_feas: bool):
    """c_pt on the device; returns (device vector, dict of diagnostics)."""
    ctx, n = dv.ctx, dv.n
    xi = ctx.to_device(_perturb_direction(n))
    c_pt = ctx.empty(n, np.float64)
    if is_feas:
        ctx.perturb_cost(n, None, None, None, dv.c, xi, 0.0, True, c_pt)
        return c_pt, {}
    x_real = ctx.empty(n, np.float64)
    ctx.x_real(n, dv.x, dv.l, dv.u, x_real, True)
    xs = _scaled_slack(dv, x_real)
    if lp.get_free_ind().size:
        info, _, _ = _projector_free_device(lp, dv, x_real, xs, False)
    else:
        info, _, _ = _projector_device(dv, x_real, xs, dv.c, 1e-8, 1000, False)
    scale_factor = get_scale_factor_from_norm(info.proj_norm, n + dv.res.n_lt)
    ctx.perturb_cost(n, dv.x, dv.l, dv.u, dv.c, xi, scale_factor, False, c_pt)
    return c_pt, dict(proj_norm=info.proj_norm, scale_factor=scale_factor, cg_iters=int(info.iters),
                      cg_converged=bool(info.converged))


def get_perturb_problem(lp: GeneralLP, x: np.ndarray, y: np.ndarray, gamma: float, gamma_dual: float,
                        is_feas: bool) -> LPManager:
    """Approximate optimal face from the interior point (x, y) and the LP restricted to it, with the
    perturbed objective (lp_methods/algorithms.py:79-111)."""
    dv = _Resident(lp, x, y)
    ctx, m, n = dv.ctx, dv.m, dv.n
    code, flag = ctx.empty(n, np.uint8), ctx.empty(m, np.uint8)
    ctx.score_columns(dv.res.A, dv.y, dv.c, dv.x, dv.l, dv.u, gamma, None, code)
    ctx.score_rows(dv.res.A, dv.x, dv.b, dv.y, gamma_dual, None, flag)

    manager = LPManager(lp._copy_sharing_matrix())
    c_pt, info = _perturb_c_device(lp, dv, is_feas)
    manager.lp.c = c_pt.download()
    manager.perturb_info = info

    fix_low, fix_up = ctx.where(code, 1), ctx.where(code, 2)
    fixed = ctx.where(code, 3)
    compacted = None
    if fixed.size:
        # K6 now: the compaction hands back the kept columns' indices (device stream compaction), so the
        # complement of the fixed set is not recomputed on the host
        sub_matrix, non_fix_dev = ctx.compact_columns(dv.res.A, code)
        compacted = (sub_matrix, non_fix_dev)
        non_fix = non_fix_dev.download()
    else:
        non_fix = np.arange(n, dtype=np.int64)
    manager._adopt_partition(code, fix_low, fix_up, non_fix, fixed, compacted)
    manager.fix_constraints(ctx.where(flag))
    print("  The number of fixed variables is %d." % manager.get_num_fixed_variables())
    print("  The number of fixed constraints is %d." % manager.get_num_fixed_constraints())
    manager.update_subproblem()
    return manager


def perturb_c(lp: GeneralLP, x: np.ndarray, is_feas: bool) -> np.ndarray:
    """Perturbed cost vector (lp_methods/algorithms.py:114-151)."""
    c_pt, _ = _perturb_c_device(lp, _Resident(lp, x), is_feas)
    return c_pt.download()


# ------------------------------------------------------------------------------------------------
# projector helpers
# ------------------------------------------------------------------------------------------------
def get_scale_factor_from_norm(proj_norm: float, n: int) -> float:
    return proj_norm / n


def get_scale_factor(projector: np.ndarray, n: int) -> float:
    """||projector|| / n (lp_methods/algorithms.py:190-193)."""
    return float(np.linalg.norm(projector)) / n


def get_x_perturb_val(lp: GeneralLP, x: np.ndarray) -> np.ndarray:
    """min(x - l, u - x), free columns keep x (lp_methods/algorithms.py:196-202)."""
    dv = _Resident(lp, x)
    out = dv.ctx.empty(dv.n, np.float64)
    dv.ctx.x_real(dv.n, dv.x, dv.l, dv.u, out, False)
    return out.download()


def _assemble_std(lp: GeneralLP, proj_cols, proj_rows) -> np.ndarray:
    rows = np.flatnonzero(np.asarray(lp.sense) == "<")
    return np.concatenate([proj_cols.download(), proj_rows.download()[rows]])


def get_projector_Xc(lp: GeneralLP, x: np.ndarray) -> np.ndarray:
    """[I - (A X)^T (A X X A^T)^+ (A X)] X c in standard form, X = diag(standard_x(x))
    (lp_methods/algorithms.py:162-180).  With free variables the result covers the non-free standard
    columns only, like the reference's QP solution (see _projector_free_device)."""
    dv = _Resident(lp, x)
    xs = _scaled_slack(dv, dv.x)
    if lp.get_free_ind().size:
        _, proj, _ = _projector_free_device(lp, dv, dv.x, xs, True)
        return proj
    _, pc, pr = _projector_device(dv, dv.x, xs, dv.c, 1e-8, 1000, True)
    return _assemble_std(lp, pc, pr)


def apply_projector(Y, v, tol: float = 1e-8, max_iter: int = 1000) -> np.ndarray:
    """(I - Y^T (Y Y^T)^+ Y) v for an explicit sparse Y (lp_methods/algorithms.py:183-187), by CG on
    the device without forming Y Y^T."""
    from smart_crossover.hip.device import default_context
    ctx = default_context()
    Y = sp.csr_matrix(Y)
    m, n = Y.shape
    dY = ctx.matrix(Y)
    pc = ctx.empty(n, np.float64)
    ctx.projector_norm(dY, ctx.to_device(np.ones(n)), ctx.to_device(np.zeros(m)),
                       ctx.to_device(np.asarray(v, dtype=np.float64)), tol, max_iter, pc, None)
    out = pc.download()
    dY.free()
    return out


def apply_projector_qp(A: sp.csr_matrix, v: np.ndarray, A_f: Optional[sp.csr_matrix] = None) -> np.ndarray:
    """Projection of v onto {x : A x = 0} (lp_methods/algorithms.py:240-265).  The reference solves a
    least-squares QP with Gurobi (BarQCPConvTol 1e-1); here it is the same CG projector as
    ``apply_projector`` (own counterpart, parity unpinned: no Gurobi).  Free columns ``A_f`` (the
    constraint becomes A x + A_f f = 0) enter the same projector with a large scale and zero cost, the
    penalty form of the QP (see _projector_free_device)."""
    if A_f is None:
        return apply_projector(A, v)
    from smart_crossover.hip.device import default_context
    ctx = default_context()
    A, A_f = sp.csr_matrix(A), sp.csr_matrix(A_f)
    m, n = A.shape
    nf = A_f.shape[1]
    # the same device routine with unit scales: columns [A, A_f], the last nf of them free, cost v on the first n
    both = sp.hstack([A, A_f], format="csr")
    dY = ctx.matrix(both)
    pc, pr = ctx.empty(n + nf, np.float64), ctx.empty(m, np.float64)
    ctx.projector_free(dY, ctx.to_device(np.arange(n, n + nf, dtype=np.int64)), ctx.to_device(np.ones(n + nf)),
                       ctx.to_device(np.zeros(m)), ctx.to_device(np.concatenate([np.asarray(v, dtype=np.float64), np.zeros(nf)])),
                       ctx.to_device(np.zeros(m, dtype=np.uint8)), pc, pr)
    out = pc.download()[:n]
    dY.free()
    return out


def get_projector_c(lp: GeneralLP) -> np.ndarray:
    """Projection of c_std onto {A_std x = 0} (lp_methods/algorithms.py:154-159)."""
    dv = _Resident(lp)
    ctx = dv.ctx
    ones = ctx.to_device(np.ones(dv.n))
    xs = ctx.empty(dv.m, np.float64)
    ctx.mask(ctx.to_device(np.ones(dv.m)), dv.res.lt, xs)
    _, pc, pr = _projector_device(dv, ones, xs, dv.c, 1e-8, 1000, True)
    return _assemble_std(lp, pc, pr)


def check_feasibility_problem(lp: GeneralLP) -> bool:
    """True when the cost is (numerically) orthogonal to the feasible directions:
    ||P c_std|| / ||c|| < 1e-8 (lp_methods/algorithms.py:227-237)."""
    dv = _Resident(lp)
    ctx = dv.ctx
    ones = ctx.to_device(np.ones(dv.n))
    xs = ctx.empty(dv.m, np.float64)
    ctx.mask(ctx.to_device(np.ones(dv.m)), dv.res.lt, xs)
    info, _, _ = _projector_device(dv, ones, xs, dv.c, 1e-8, 1000, False)
    with np.errstate(all="ignore"):
        ratio = np.float64(info.proj_norm) / np.float64(np.linalg.norm(lp.c))   # 0/0 -> nan -> False, as in numpy
    if ratio < PROJECTOR_THRESHOLD:
        print("*** The problem is a feasibility problem. ***")
        return True
    return False


def check_perturb_output_precision(sublp_manager: LPManager, x_ptb: np.ndarray, c_ori: np.ndarray,
                                   barrier_obj: float) -> Optional[bool]:
    """Relative gap between c_ori^T x (x lifted back) and the barrier objective; True below 1e-8,
    otherwise None -- never False (quirk Q3, lp_methods/algorithms.py:205-224)."""
    x = sublp_manager.get_orix(x_ptb)
    mine = float(c_ori @ x)
    rel = abs(mine - barrier_obj) / (abs(mine) + abs(barrier_obj) + 1)
    print()
    print("*** Primal-dual gap: %(gap).2e ***" % {"gap": rel})
    return True if rel < PRIMAL_DUAL_GAP_THRESHOLD else None
