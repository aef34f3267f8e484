"""``LPManager``: bookkeeping of a restricted LP (API of the reference's
``lp_methods/lp_manager.py:8-134``) with the heavy steps on the MI355X.

What stays on the host: the index arrays of ``var_info`` (int64, like the reference) and the small
recover/scatter helpers.  What runs in libsxhip.so: the column partition (stream compaction of the
per-column codes), the sub-matrix ``A[:, non_fix]`` in CSR and CSC, the right-hand side
``b - A[:, fix_up] u - A[:, fix_low] l`` (bit-exact with the reference's two sliced products) and
the gathers of c, l, u.  ``lp_sub`` is materialised as a host ``GeneralLP`` because the re-solve
behind the ``SolverCaller`` seam takes host arrays; the compacted matrix also stays resident
(``lp_sub._sx_resident``) for device consumers.

Reference quirks kept: Q1 ``recover_x_from_sub_x`` leaves columns fixed at their lower bound at 0;
``lp_sub is lp`` when nothing is fixed; ``lp_sub.sense`` aliases ``lp.sense``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import GeneralLP
from smart_crossover.output import Basis

_I64 = np.int64


def _idx(a) -> np.ndarray:
    return np.asarray(a, dtype=_I64).ravel()


class LPManager:
    m: int
    n: int
    lp: GeneralLP
    var_info: Dict[str, np.ndarray]
    lp_sub: GeneralLP
    basis: Basis

    def __init__(self, lp: GeneralLP) -> None:
        self.lp = lp
        self.m = self.lp.b.size
        self.n = self.lp.c.size
        empty = np.array([], dtype=_I64)
        self.var_info = {"non_fix": np.arange(self.n, dtype=_I64), "fix_low": empty, "fix_up": empty, "fix": empty}
        self.fixed_constraints = empty
        self._dev_code = None          # per-column code already on the device (set by get_perturb_problem)
        self._compacted = None         # (sub-matrix, kept columns) when get_perturb_problem has compacted already

    # ------------------------------------------------------------------ partition
    def _code_host(self) -> np.ndarray:
        code = np.zeros(self.n, dtype=np.uint8)
        code[self.var_info["fix_low"]] |= 1
        code[self.var_info["fix_up"]] |= 2
        return code

    def fix_variables(self, ind_fix_to_low: np.ndarray, ind_fix_to_up: np.ndarray) -> None:
        """Fix columns at their lower / upper bound in the sub-problem (lp_manager.py:40-50)."""
        low, up = _idx(ind_fix_to_low), _idx(ind_fix_to_up)
        self.var_info["fix_low"] = low
        self.var_info["fix_up"] = up
        keep = np.ones(self.n, dtype=bool)
        keep[low] = False
        keep[up] = False
        self.var_info["non_fix"] = np.flatnonzero(keep).astype(_I64)
        self.var_info["fix"] = np.flatnonzero(~keep).astype(_I64)
        self._dev_code = None
        self._compacted = None

    def _adopt_partition(self, dev_code, fix_low, fix_up, non_fix, fix, compacted=None) -> None:
        """Partition computed on the device by get_perturb_problem (same sets as fix_variables); ``compacted`` =
        (sub-matrix, kept column indices) when the compaction has already run."""
        self.var_info.update(fix_low=fix_low, fix_up=fix_up, non_fix=non_fix, fix=fix)
        self._dev_code = dev_code
        self._compacted = compacted

    def fix_constraints(self, ind_fix_to_up: np.ndarray) -> None:
        self.fixed_constraints = _idx(ind_fix_to_up)

    def get_num_fixed_variables(self) -> int:
        return int(self.var_info["fix"].size)

    def get_num_fixed_constraints(self) -> int:
        return int(self.fixed_constraints.size)

    # ------------------------------------------------------------------ sub-problem (device)
    def update_subproblem(self) -> None:
        """Build ``lp_sub`` (lp_manager.py:52-66)."""
        if self.var_info["fix"].size == 0:
            self.lp_sub = self.lp
        else:
            from smart_crossover.hip.resident import ResidentLP, resident_for
            res = resident_for(self.lp)
            ctx = res.ctx
            code = self._dev_code if self._dev_code is not None else ctx.to_device(self._code_host())
            d_b, d_c, d_l, d_u = (res.put(v) for v in (self.lp.b, self.lp.c, self.lp.l, self.lp.u))
            sub_matrix, non_fix = self._compacted if self._compacted is not None else ctx.compact_columns(res.A, code)
            self._compacted = None
            if non_fix.size != self.var_info["non_fix"].size:
                raise RuntimeError("device partition disagrees with var_info")      # cannot happen; cheap guard
            b_sub = ctx.empty(self.m, np.float64)
            ctx.fixed_rhs(res.A, code, d_u, d_l, d_b, b_sub)
            c_sub, l_sub, u_sub = (ctx.gather(non_fix, d).download() for d in (d_c, d_l, d_u))
            self.lp_sub = GeneralLP(A=sub_matrix.to_scipy(), b=b_sub.download(), c=c_sub, l=l_sub, u=u_sub,
                                    sense=self.lp.sense)
            # keep the compacted matrix resident for device consumers of lp_sub
            self.lp_sub._sx_resident = ResidentLP.adopt(ctx, sub_matrix, self.lp_sub.A, self.lp.sense)
        if self.fixed_constraints.size > 0:
            self.lp_sub.sense[self.fixed_constraints] = "="

    # ------------------------------------------------------------------ recovery (host index glue)
    def recover_x_from_sub_x(self, x_sub: np.ndarray) -> np.ndarray:
        x = np.zeros(self.lp.c.size)
        x[self.var_info["non_fix"]] = x_sub
        x[self.var_info["fix_up"]] = self.lp.u[self.var_info["fix_up"]]
        return x

    def recover_basis_from_sub_basis(self, basis_sub: Basis) -> Basis:
        vbasis = np.full(self.lp.c.size, -1, dtype=int)
        vbasis[self.var_info["non_fix"]] = basis_sub.vbasis
        vbasis[self.var_info["fix_up"]] = -2
        return Basis(vbasis, basis_sub.cbasis)

    def get_subx(self, x: np.ndarray) -> np.ndarray:
        return x[self.var_info["non_fix"]]

    def get_orix(self, x_sub: np.ndarray) -> np.ndarray:
        x = self.recover_x_from_sub_x(x_sub)
        x[self.var_info["fix_low"]] = self.lp.l[self.var_info["fix_low"]]
        return x

    def update_c(self, c_sub_new: np.ndarray) -> None:
        self.lp.c[self.var_info["non_fix"]] = c_sub_new
        self.lp_sub.c = c_sub_new
