"""Managers of the network crossover -- API of the reference's
``network_methods/net_manager.py`` (NetworkManager :14, MCFManagerStd :116, OTManager :322).

On the MI355X (libsxhip.so): flow indicators (K7 / K8), their ranking (K9), the reduced-cost
optimality test (K10, on the sparse incidence matrix for MCF and on the implicit OT structure for
OT), the sub-problem matrices ``A[:, non_fix]`` (K12, released columns keep queue order) and the
fixed-arc right-hand sides.  On the host: the int64 index bookkeeping of ``var_info`` / the OT mask
and the one-off big-M assembly, as in the reference.

Ranking ties: the reference's queue comes from numpy's unstable argsort; here ties are ordered by
descending index (see include/sxhip.h, K9).  Indicators themselves are bit-exact.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
from scipy import sparse as sp

try:  # Python >= 3.8
    from typing import Protocol
except ImportError:  # pragma: no cover
    from typing_extensions import Protocol

from smart_crossover.formats import MinCostFlow, OptTransport
from smart_crossover.output import Basis, Output
from smart_crossover.parameters import TOLERANCE_FOR_ARTIFICIAL_VARS, TOLERANCE_FOR_REDUCED_COSTS
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.solver_caller.solving import solve_mcf

_I64 = np.int64


class NetworkManager(Protocol):
    """What ``column_generation`` needs from a manager."""

    m: int
    n: int
    basis: Basis

    def get_sorted_flows(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]: ...

    def recover_x_from_sub_x(self, x_sub: np.ndarray) -> np.ndarray: ...

    def recover_basis_from_sub_basis(self, basis_sub: Basis) -> Basis: ...

    def solve_subproblem(self, solver: str, solver_settings: SolverSettings) -> Output: ...

    def recover_obj_val(self, obj_val: float) -> float: ...

    def check_optimality_condition(self, x: np.ndarray, y: np.ndarray) -> bool: ...

    def add_free_variables(self, ind_free: np.ndarray) -> None: ...

    def update_subproblem(self) -> None: ...

    def set_basis(self, basis: Basis) -> None: ...


def _ctx():
    from smart_crossover.hip.device import default_context
    return default_context()


def _canonical(A) -> sp.csr_matrix:
    """A as canonical CSR; the *same object* when it already is one (the device copy is cached per object:
    a fresh wrapper on every call would re-upload the matrix each time)."""
    if sp.isspmatrix_csr(A) and A.has_canonical_format:
        return A
    A = sp.csr_matrix(A)
    if not A.has_canonical_format:
        A = A.copy()
        A.sum_duplicates()
        A.sort_indices()
    return A


class _ResidentMatrix:
    """Device copy of a host matrix, rebuilt when the host object changes."""

    def __init__(self):
        self.key = None
        self.dev = None

    def get(self, A: sp.csr_matrix):
        from smart_crossover.hip.resident import matrix_fingerprint
        key = matrix_fingerprint(A)      # addresses and sizes of the three arrays, not the wrapper's id
        if self.key != key or self.dev is None or self.dev.handle is None:
            if self.dev is not None:
                self.dev.free()
            self.dev = _ctx().matrix(A)
            self.key = key
        return self.dev


# ====================================================================================== MCF
class _SessionHolder:
    """Where a device-resident backend may keep state between the sub-problem solves of one manager
    (the HIP simplex keeps its basis inverse here); other backends ignore it."""
    session = None


def _tag_sub_problem(sub: MinCostFlow, col_ids: np.ndarray, holder: _SessionHolder, dev=None) -> None:
    """Stable identifiers of the sub-problem's columns (their index in the full problem), the state
    holder and the device copy of the sub-matrix the gather kernel has just built (so a device backend
    need not upload the host copy again), as plain attributes the reference's dataclass does not have;
    only backends that look for them (solver_caller/hip.py) are affected."""
    sub.col_ids = np.asarray(col_ids, dtype=_I64)
    sub._sx_device_matrix = dev
    sub.hip_session = holder


class MCFManagerStd:
    """min-cost-flow manager (reference net_manager.py:116-319)."""

    mcf: MinCostFlow

    def __init__(self, mcf: MinCostFlow) -> None:
        self.mcf = mcf                       # kept by reference, not copied (quirk Q7 relies on it)
        self.m = self.mcf.b.size
        self.n = self.mcf.c.size
        self.var_info = {"non_fix": np.arange(self.n, dtype=_I64)}
        self.artificial_vars = np.array([])
        self.c_rescaling_factor = None
        self._resident = _ResidentMatrix()
        self._spx_holder = _SessionHolder()

    def _dev_matrix(self):
        self.mcf.A = _canonical(self.mcf.A)
        return self._resident.get(self.mcf.A)

    # -- scoring -------------------------------------------------------------------------------
    def get_sorted_flows(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(queue, indicators): arcs ranked by how much of a node's throughput they carry
        (net_manager.py:156-184) -- kernels K7 and K9."""
        ctx = _ctx()
        ind = ctx.empty(self.n, np.float64)
        ctx.flow_indicator_mcf(self._dev_matrix(), ctx.to_device(np.asarray(x, float)),
                               ctx.to_device(np.asarray(self.mcf.u, float)), ind)
        queue = ctx.argsort_desc(ind)
        return queue.download(), ind.download()

    # -- big-M extension (one-off host assembly, net_manager.py:135-154) ---------------------------
    def extend_by_bigM(self, bigM: float) -> None:
        ctx = _ctx()
        n, m = self.n, self.m
        code = np.zeros(n, dtype=np.uint8)
        code[self.var_info["fix_up"]] = 2
        b_true_dev = ctx.empty(m, np.float64)
        zeros = ctx.zeros(n, np.float64)
        # b - A[:, fix_up] u[fix_up]: per node, arcs in ascending order (same sums as the reference's
        # masked product for finite capacities; an infinite capacity never multiplies a zero here)
        ctx.fixed_rhs(self._dev_matrix(), ctx.to_device(code), ctx.to_device(np.asarray(self.mcf.u, float)), zeros,
                      ctx.to_device(np.asarray(self.mcf.b, float)), b_true_dev)
        b_sign = np.sign(b_true_dev.download())
        b_sign[b_sign == 0] = 1
        c_1 = np.concatenate([self.mcf.c, bigM * np.ones(m)])
        u_1 = np.concatenate([self.mcf.u, np.inf * np.ones(n)])          # n, not m, infinities: quirk Q8
        top = sp.hstack((self.mcf.A, sp.diags(b_sign)))
        A_1 = sp.vstack((top, sp.csr_matrix(np.concatenate([np.zeros(n), -b_sign])))).tocsr()
        b_1 = np.concatenate([self.mcf.b, [0.0]])
        self.mcf = MinCostFlow(A_1, b_1, c_1, u_1)
        self.artificial_vars = np.arange(n, n + m, dtype=int)
        self.var_info["non_fix"] = np.append(self.var_info["non_fix"], np.arange(n, n + m, dtype=_I64))

    def set_initial_basis(self) -> None:
        """Artificial arcs basic, original arcs at a bound (net_manager.py:186-192)."""
        vbasis = np.concatenate((-np.ones(self.n), np.zeros(self.m)))
        vbasis[self.var_info["fix_up"]] = -2
        self.set_basis(Basis(vbasis, np.concatenate([-np.ones(self.m), np.zeros(1)])))

    def set_basis(self, basis: Basis) -> None:
        self.basis = basis

    # -- sub-problem (K12) ----------------------------------------------------------------------
    def update_subproblem(self) -> None:
        """mcf_sub = columns non_fix (in that order), b moved by the arcs fixed at capacity
        (net_manager.py:202-209)."""
        ctx = _ctx()
        dA = self._dev_matrix()
        ncol = self.mcf.c.size
        non_fix = np.asarray(self.var_info["non_fix"], dtype=_I64)
        sub = ctx.gather_columns(dA, ctx.to_device(non_fix))
        code = np.zeros(ncol, dtype=np.uint8)
        code[self.var_info["fix_up"]] = 2
        u_dev = ctx.to_device(np.asarray(self.mcf.u, float))
        b_sub = ctx.empty(self.mcf.b.size, np.float64)
        ctx.fixed_rhs(dA, ctx.to_device(code), u_dev, ctx.zeros(ncol, np.float64), ctx.to_device(np.asarray(self.mcf.b, float)),
                      b_sub)
        self.mcf_sub = MinCostFlow(A=sub.to_scipy(), b=b_sub.download(), c=self.mcf.c[non_fix], u=self.mcf.u[non_fix])
        _tag_sub_problem(self.mcf_sub, non_fix, self._spx_holder, sub)      # `sub` is freed with the sub-problem object

    def solve_subproblem(self, solver: str, solver_settings: SolverSettings) -> Output:
        method = "network_simplex" if solver == "CPL" else "default"
        warm = Basis(self.basis.vbasis[self.var_info["non_fix"]], self.basis.cbasis)
        return solve_mcf(self.mcf_sub, solver=solver, method=method, warm_start_basis=warm, settings=solver_settings)

    # -- partition bookkeeping (host index arrays, as in the reference) -----------------------------
    def fix_variables(self, ind_fix_to_low: np.ndarray, ind_fix_to_up: np.ndarray) -> None:
        low, up = np.asarray(ind_fix_to_low, dtype=_I64), np.asarray(ind_fix_to_up, dtype=_I64)
        self.var_info["fix_low"] = low
        self.var_info["fix_up"] = up
        keep = np.ones(len(self.mcf.c), dtype=bool)
        keep[low] = False
        keep[up] = False
        self.var_info["non_fix"] = np.flatnonzero(keep).astype(_I64)
        self.var_info["fix"] = np.flatnonzero(~keep).astype(_I64)

    def add_free_variables(self, ind_free_new: np.ndarray) -> None:
        """Release columns: appended to non_fix in the given (queue) order (net_manager.py:236-245)."""
        new = np.asarray(ind_free_new, dtype=_I64)
        self.var_info["non_fix"] = np.append(self.var_info["non_fix"], new)
        gone = np.zeros(len(self.mcf.c), dtype=bool)
        gone[new] = True
        for key in ("fix", "fix_low", "fix_up"):
            cur = self.var_info[key]
            self.var_info[key] = cur[~gone[cur]]            # == np.setdiff1d for the sorted unique sets kept here

    def recover_x_from_sub_x(self, x_sub: np.ndarray) -> np.ndarray:
        x = np.zeros(self.mcf.c.size)
        x[self.var_info["non_fix"]] = x_sub
        x[self.var_info["fix_up"]] = self.mcf.u[self.var_info["fix_up"]]
        return x

    def recover_basis_from_sub_basis(self, basis_sub: Basis) -> Basis:
        vbasis = np.full(self.mcf.c.size, -1, dtype=int)
        vbasis[self.var_info["non_fix"]] = basis_sub.vbasis
        vbasis[self.var_info["fix_up"]] = -2
        return Basis(vbasis, basis_sub.cbasis)

    def rescale_cost(self, factor: float) -> None:
        """c <- c / factor on the *caller's* MinCostFlow object (quirk Q7, net_manager.py:276-283)."""
        self.mcf.c = self.mcf.c / factor
        self.c_rescaling_factor = factor

    def recover_obj_val(self, obj_val: float) -> float:
        return obj_val * self.c_rescaling_factor

    # -- pricing (K10) --------------------------------------------------------------------------
    def _price(self, y: np.ndarray, want_rc: bool):
        ctx = _ctx()
        n = self.mcf.c.size
        rc = ctx.empty(n, np.float64) if want_rc else None
        vb = np.clip(np.asarray(self.basis.vbasis), -128, 127).astype(np.int8)
        res = ctx.price(self._dev_matrix(), ctx.to_device(np.asarray(y, float)), ctx.to_device(np.asarray(self.mcf.c, float)),
                        ctx.to_device(vb), TOLERANCE_FOR_REDUCED_COSTS, rc)
        return rc, ctx.read_price(res)

    def get_reduced_cost_for_original_mcf(self, y: np.ndarray) -> np.ndarray:
        """c - A^T y, sign flipped on arcs at capacity (net_manager.py:293-304)."""
        rc, _ = self._price(y, True)
        return rc.download()

    def check_optimality_condition(self, x: np.ndarray, y: np.ndarray) -> bool:
        """No flow on artificial arcs and no reduced cost below -1e-6 (net_manager.py:306-319)."""
        art_ok = bool(np.all(x[self.artificial_vars] < TOLERANCE_FOR_ARTIFICIAL_VARS)) if self.artificial_vars.size > 0 else True
        _, (_, _, n_bad) = self._price(y, False)
        return art_ok and n_bad == 0


# ====================================================================================== OT
class OTManager:
    """optimal-transport manager (reference net_manager.py:322-509)."""

    ot: OptTransport
    mask_sub_ot: np.ndarray
    basis: Basis
    artificial_vars: np.ndarray
    mcf: MinCostFlow

    def __init__(self, ot: OptTransport) -> None:
        self.ot = ot
        self.m = ot.s.size + ot.d.size
        self.n = ot.s.size * ot.d.size
        self.mask_sub_ot = np.zeros(self.n, dtype=bool)
        self.artificial_vars = np.array([])
        self._resident = _ResidentMatrix()
        self._spx_holder = _SessionHolder()

    def get_mcf(self) -> None:
        self.mcf = self.ot.to_MCF()

    def get_X(self, x: np.ndarray) -> np.ndarray:
        return x.reshape((self.ot.s.size, self.ot.d.size))

    def get_sorted_flows(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """max(X_ij / s_i, X_ij / d_j) and its ranking (net_manager.py:368-379) -- kernels K8, K9."""
        ctx = _ctx()
        S, D = self.ot.s.size, self.ot.d.size
        ind = ctx.empty(S * D, np.float64)
        ctx.flow_indicator_ot(S, D, ctx.to_device(np.asarray(x, float).ravel()), ctx.to_device(np.asarray(self.ot.s, float)),
                              ctx.to_device(np.asarray(self.ot.d, float)), ind)
        queue = ctx.argsort_desc(ind)
        return queue.download(), ind.download()

    def extend_by_bigM(self, bigM: float) -> None:
        """One artificial supplier and one artificial demander (net_manager.py:381-400)."""
        S, D = self.ot.s.size, self.ot.d.size
        M = np.asarray(self.ot.M.toarray() if sp.issparse(self.ot.M) else self.ot.M, dtype=float)
        M1 = np.empty((S + 1, D + 1))
        M1[:S, :D] = M
        M1[:S, D] = bigM
        M1[S, :D] = bigM
        M1[S, D] = 0
        mask = np.zeros((S + 1, D + 1), dtype=np.bool_)
        mask[:, D] = True
        mask[S, :] = True
        self.mask_sub_ot = mask
        self.artificial_vars = np.where(mask.ravel())[0]
        self.ot = OptTransport(np.append(self.ot.s, np.sum(self.ot.d)), np.append(self.ot.d, np.sum(self.ot.s)), M1)

    def add_free_variables(self, ind_free: np.ndarray) -> None:
        """Release arcs: linear indices of the *original* S x D block once the problem is extended;
        indices or a boolean mask of the flat problem before (quirk Q10, net_manager.py:402-414)."""
        if self.artificial_vars.size > 0:
            inner = self.mask_sub_ot[:-1, :-1]
            rows, cols = np.unravel_index(ind_free, inner.shape)
            inner[rows, cols] = True
            self.mask_sub_ot[:-1, :-1] = inner
        else:
            self.mask_sub_ot[np.asarray(ind_free).ravel()] = True

    def set_basis(self, basis: Basis) -> None:
        self.basis = basis

    def recover_x_from_sub_x(self, x_sub: np.ndarray) -> np.ndarray:
        x = np.zeros(self.ot.s.size * self.ot.d.size)
        x[self.mask_sub_ot.ravel()] = x_sub
        return x

    def recover_basis_from_sub_basis(self, basis_sub: Basis) -> Basis:
        vbasis = -np.ones(self.ot.s.size * self.ot.d.size)
        vbasis[self.mask_sub_ot.ravel()] = basis_sub.vbasis
        return Basis(vbasis, basis_sub.cbasis)

    def get_sub_problem(self) -> MinCostFlow:
        """Columns of the incidence matrix selected by the mask (net_manager.py:450-455), gathered on
        the device."""
        ctx = _ctx()
        mask = self.mask_sub_ot.ravel()
        cols = np.flatnonzero(mask).astype(_I64)
        sub = ctx.gather_columns(self._resident.get(self.mcf.A), ctx.to_device(cols))
        M = np.asarray(self.ot.M.toarray() if sp.issparse(self.ot.M) else self.ot.M)
        out = MinCostFlow(A=sub.to_scipy(), b=self.mcf.b, c=M.flatten()[mask], u=self.mcf.u[mask])
        _tag_sub_problem(out, cols, self._spx_holder, sub)                   # `sub` is freed with `out`
        return out

    def solve_subproblem(self, solver: str, solver_settings: SolverSettings) -> Output:
        method = "network_simplex" if solver == "CPL" else "default"
        warm = Basis(self.basis.vbasis[self.mask_sub_ot.ravel()], self.basis.cbasis)
        return solve_mcf(self.get_sub_problem(), solver=solver, method=method, warm_start_basis=warm,
                         settings=solver_settings)

    def recover_obj_val(self, obj_val):
        return obj_val

    def _price(self, y: np.ndarray, want_rc: bool):
        ctx = _ctx()
        S, D = self.ot.s.size, self.ot.d.size
        M = np.asarray(self.ot.M.toarray() if sp.issparse(self.ot.M) else self.ot.M, dtype=float)
        rc = ctx.empty(S * D, np.float64) if want_rc else None
        res = ctx.price_ot(S, D, ctx.to_device(M.ravel()), ctx.to_device(np.asarray(y, float)),
                           TOLERANCE_FOR_REDUCED_COSTS, rc)
        return rc, ctx.read_price(res)

    def get_reduced_cost_for_original_OT(self, y: np.ndarray) -> np.ndarray:
        """M_ij - (y_{S+j} - y_i) for every arc (net_manager.py:474-483), without the kron matrix."""
        rc, _ = self._price(y, True)
        return rc.download()

    def check_optimality_condition(self, x: np.ndarray, y: np.ndarray) -> bool:
        """net_manager.py:485-497 (the corner artificial arc is exempt from the flow test)."""
        art_ok = bool(np.all(x[self.artificial_vars][:-1] < TOLERANCE_FOR_ARTIFICIAL_VARS)) if self.artificial_vars.size > 0 else True
        _, (_, _, n_bad) = self._price(y, False)
        return art_ok and n_bad == 0

    def update_subproblem(self):
        """Nothing to rebuild: the mask is the sub-problem."""

    def set_initial_basis(self) -> None:
        vbasis = -np.ones(self.ot.s.size * self.ot.d.size)
        vbasis[self.artificial_vars] = 0
        self.basis = Basis(vbasis, np.concatenate([-np.ones(self.m + 1), np.zeros(1)]))
