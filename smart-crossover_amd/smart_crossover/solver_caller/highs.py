"""``HgsCaller``: the open-source HiGHS solver (bundled with scipy as
``scipy.optimize._highspy``) behind the ``SolverCaller`` seam.

The reference drives Gurobi / CPLEX / Mosek through this seam (solver_caller/gurobi.py etc.); none
of them can be installed here, so HiGHS stands in for *the third-party solver* -- the re-solves of
the restricted problems -- on both the CPU baseline and the GPU path.  It plays no part in the
scoring / projector / compaction arithmetic, which is libsxhip.so's.  Mapping of the reference's
Gurobi parameter choices (gurobi.py:111-146,212-242):

  run_barrier               solver=ipm, run_crossover=on (x_bar not exposed by HiGHS: None)
  run_barrier_no_crossover  solver=ipm, run_crossover=off
  run_default / run_simplex solver=choose
  run_primal_simplex        solver=simplex, simplex_strategy=4 (primal)
  run_dual_simplex          solver=simplex, simplex_strategy=1 (dual)
  run_network_simplex       HiGHS has no network simplex: dual simplex (documented substitution)
  warm start                VBasis/CBasis codes -> HighsBasis; PStart -> setSolution(col_value)
  status                    kOptimal/OPTIMAL, kInfeasible + kUnboundedOrInfeasible/INFEASIBLE,
                            kUnbounded/UNBOUNDED, anything else/UNKNOWN
"""
from __future__ import annotations

import datetime
import time
from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import GeneralLP, StandardLP
from smart_crossover.output import Basis
from smart_crossover.solver_caller.caller import SolverCaller, SolverSettings


def _core():
    try:
        import scipy.optimize._highspy._core as hc
    except Exception as exc:  # pragma: no cover - depends on the scipy build
        raise ImportError("the HGS backend needs scipy's bundled HiGHS (scipy >= 1.15)") from exc
    return hc


class HgsCaller(SolverCaller):
    solver_name = "HGS"

    def __init__(self, solver_settings: Optional[SolverSettings] = None) -> None:
        super().__init__(solver_settings)
        self.hc = _core()
        self.h = self.hc._Highs()
        self.h.setOptionValue("output_flag", False)
        self._x_bar: Optional[np.ndarray] = None
        self._runtime = 0.0
        self._used_ipm = False
        self._crossover_off = False
        self._iters = 0
        self._bar_iters = 0
        self._ran = False

    # -- loading -------------------------------------------------------------------------------
    def _pass(self, A, c, l, u, row_lower, row_upper) -> None:
        hc = self.hc
        A = sp.csc_matrix(A)
        m, n = A.shape
        lp = hc.HighsLp()
        lp.num_col_, lp.num_row_ = n, m
        inf = hc.kHighsInf
        lp.col_cost_ = np.asarray(c, dtype=np.float64)
        lp.col_lower_ = np.clip(np.asarray(l, dtype=np.float64), -inf, inf)
        lp.col_upper_ = np.clip(np.asarray(u, dtype=np.float64), -inf, inf)
        lp.row_lower_ = np.clip(row_lower, -inf, inf)
        lp.row_upper_ = np.clip(row_upper, -inf, inf)
        lp.a_matrix_.format_ = hc.MatrixFormat.kColwise
        lp.a_matrix_.num_col_, lp.a_matrix_.num_row_ = n, m
        lp.a_matrix_.start_ = A.indptr.astype(np.int32)
        lp.a_matrix_.index_ = A.indices.astype(np.int32)
        lp.a_matrix_.value_ = A.data.astype(np.float64)
        status = self.h.passModel(lp)
        if status == hc.HighsStatus.kError:
            raise ValueError("HiGHS rejected the model")
        self._A, self._c, self._l, self._u = A, np.asarray(c, float), np.asarray(l, float), np.asarray(u, float)
        self._row_lower, self._row_upper = np.asarray(row_lower, float), np.asarray(row_upper, float)
        self._ran = False

    def read_genlp(self, genlp: GeneralLP) -> None:
        b = np.asarray(genlp.b, dtype=np.float64)
        lower = np.where(np.asarray(genlp.sense) == "=", b, -np.inf)
        self._pass(genlp.A, genlp.c, genlp.l, genlp.u, lower, b)

    def read_stdlp(self, stdlp: StandardLP) -> None:
        # deviation from quirk Q11 (GrbCaller.read_stdlp drops l): the lower bounds are honoured
        b = np.asarray(stdlp.b, dtype=np.float64)
        self._pass(stdlp.A, stdlp.c, stdlp.l, stdlp.u, b, b)

    def read_model_from_file(self, path: str) -> None:
        """MPS / LP file -> continuous relaxation (reference: gurobipy.read + relax + presolve,
        gurobi.py:25-29; the HiGHS presolve is applied at solve time instead of materialised)."""
        hc = self.hc
        if self.h.readModel(path) == hc.HighsStatus.kError:
            raise ValueError(f"cannot read {path}")
        lp = self.h.getLp()
        n, m = lp.num_col_, lp.num_row_
        if n and len(lp.integrality_):
            self.h.changeColsIntegrality(n, np.arange(n, dtype=np.int32),
                                         np.full(n, hc.HighsVarType.kContinuous))
            lp = self.h.getLp()
        mat = lp.a_matrix_
        if mat.format_ == hc.MatrixFormat.kColwise:
            A = sp.csc_matrix((np.array(mat.value_), np.array(mat.index_), np.array(mat.start_)), shape=(m, n))
        else:
            A = sp.csr_matrix((np.array(mat.value_), np.array(mat.index_), np.array(mat.start_)), shape=(m, n)).tocsc()
        sign = -1.0 if lp.sense_ == hc.ObjSense.kMaximize else 1.0
        self._pass(A, sign * np.array(lp.col_cost_), np.array(lp.col_lower_), np.array(lp.col_upper_),
                   np.array(lp.row_lower_), np.array(lp.row_upper_))

    # -- model data in the reference's '=' / '<' row form -----------------------------------------
    def _row_form(self):
        """Rows as (A, b, sense) with only '=' and '<': a '>' row is negated, a ranged row is split in
        two, a free row is dropped (what return_genlp of the reference assumes of a presolved model)."""
        inf = self.hc.kHighsInf
        lo, up = self._row_lower, self._row_upper
        A = sp.csr_matrix(self._A)
        eq = (lo == up) & (np.abs(lo) < inf)
        le = ~eq & (up < inf)
        ge = ~eq & (lo > -inf)
        blocks, rhs, sense = [], [], []
        for mask, sgn, vals, tag in ((eq, 1.0, up, "="), (le, 1.0, up, "<"), (ge, -1.0, -lo, "<")):
            idx = np.flatnonzero(mask)
            if idx.size:
                blocks.append(A[idx] * sgn)
                rhs.append(vals[idx])
                sense.append(np.full(idx.size, tag))
        if not blocks:
            return sp.csr_matrix((0, A.shape[1])), np.zeros(0), np.zeros(0, dtype="<U1")
        return sp.vstack(blocks, format="csr"), np.concatenate(rhs), np.concatenate(sense)

    def get_A(self) -> sp.csr_matrix:
        return self._row_form()[0]

    def get_b(self) -> np.ndarray:
        return self._row_form()[1]

    def get_sense(self) -> np.ndarray:
        return self._row_form()[2]

    def get_c(self) -> np.ndarray:
        return self._c.copy()

    def get_l(self) -> np.ndarray:
        inf = self.hc.kHighsInf
        return np.where(self._l <= -inf, -np.inf, self._l)

    def get_u(self) -> np.ndarray:
        inf = self.hc.kHighsInf
        return np.where(self._u >= inf, np.inf, self._u)

    # -- warm starts ---------------------------------------------------------------------------
    def add_warm_start_basis(self, basis: Basis) -> None:
        hc = self.hc
        st = hc.HighsBasisStatus
        hb = hc.HighsBasis()
        # codes 0 / -1 / -2 / -3 -> table slots 0..3 (anything else: at lower); one C-level pass each
        col_table = (st.kBasic, st.kLower, st.kUpper, st.kZero)
        vb = np.asarray(basis.vbasis).astype(np.int64)
        slot = np.where((vb <= 0) & (vb >= -3), -vb, 1)
        hb.col_status = list(map(col_table.__getitem__, slot.tolist()))
        row_table = (st.kBasic, st.kUpper, st.kLower)
        cb = np.asarray(basis.cbasis).astype(np.int64)
        rslot = np.where(cb == 0, 0, np.where(self._row_upper < hc.kHighsInf, 1, 2))
        hb.row_status = list(map(row_table.__getitem__, rslot.tolist()))
        hb.valid = True
        hb.alien = True      # let HiGHS repair a basis that is not exactly m x m non-singular
        self.h.setBasis(hb)

    def add_warm_start_solution(self, start_solution: Tuple[np.ndarray, np.ndarray]) -> None:
        sol = self.hc.HighsSolution()
        sol.col_value = np.asarray(start_solution[0], dtype=np.float64)
        self.h.setSolution(sol)

    # -- runs ----------------------------------------------------------------------------------
    def _apply_settings(self) -> None:
        s, h = self.settings, self.h
        h.setOptionValue("presolve", "off" if s.presolve == "off" else "choose")
        h.setOptionValue("ipm_optimality_tolerance", float(s.barrierTol))
        h.setOptionValue("dual_feasibility_tolerance", float(s.optimalityTol))
        h.setOptionValue("time_limit", float(s.timeLimit))
        h.setOptionValue("ipm_iteration_limit", int(s.iterLimit))
        h.setOptionValue("output_flag", bool(s.log_console) or bool(s.log_file))
        h.setOptionValue("log_to_console", bool(s.log_console))
        if s.log_file:
            h.setOptionValue("log_file", s.log_file)
        if s.simplexPricing == "SE":
            h.setOptionValue("simplex_dual_edge_weight_strategy", 2)
            h.setOptionValue("simplex_primal_edge_weight_strategy", 2)
        elif s.simplexPricing == "PP":
            h.setOptionValue("simplex_dual_edge_weight_strategy", 0)
            h.setOptionValue("simplex_primal_edge_weight_strategy", 0)

    def _run(self, solver: str, crossover: str = "on", strategy: Optional[int] = None) -> None:
        self._apply_settings()
        self.h.setOptionValue("solver", solver)
        self.h.setOptionValue("run_crossover", crossover)
        if solver == "ipm" and crossover == "off":
            # An interior point without crossover cannot be postsolved reliably by this HiGHS: when the
            # run ends "Unknown" (IPX optimal, tiny residual corrections) the returned column values belong
            # to the presolved model (c^T x != reported objective).  The interior point is the *input* of
            # the perturbation crossover, so it is computed on the unpresolved model.
            self.h.setOptionValue("presolve", "off")
        if strategy is not None:
            self.h.setOptionValue("simplex_strategy", strategy)
        self._ipm_certificate = None
        t0 = time.perf_counter()
        self.h.run()
        self._runtime = time.perf_counter() - t0
        info = self.h.getInfo()
        self._iters = int(info.simplex_iteration_count) + max(int(info.crossover_iteration_count), 0)
        self._bar_iters = max(int(info.ipm_iteration_count), 0)
        self._used_ipm = solver == "ipm"
        self._crossover_off = self._used_ipm and crossover == "off"
        self._x_bar = None
        self._ran = True
        self._log_summary(self._runtime, None if self._crossover_off else self._iters,
                          self._bar_iters if self._used_ipm else None)

    def run_barrier_no_crossover(self) -> None:
        self._run("ipm", "off")
        if self.return_status() == "OPTIMAL":
            self._x_bar = np.array(self.h.getSolution().col_value)

    def run_barrier(self) -> None:
        """Interior point followed by HiGHS's own crossover in one run.  HiGHS does not expose the
        interior iterate of such a run (and IPX rejects an external crossover start), so x_bar is
        None here; a crossover-off run returns x == x_bar."""
        self._run("ipm", "on")

    def run_default(self) -> None:
        self._run("choose")

    def run_simplex(self) -> None:
        self._run("choose")

    def run_primal_simplex(self) -> None:
        self._run("simplex", strategy=4)

    def run_dual_simplex(self) -> None:
        self._run("simplex", strategy=1)

    def run_network_simplex(self) -> None:
        self._run("simplex", strategy=1)

    def reset_model(self) -> None:
        self.h.clearSolver()
        self._ran = False

    # -- results -------------------------------------------------------------------------------
    def return_status(self) -> str:
        ms = self.hc.HighsModelStatus
        st = self.h.getModelStatus()
        if st == ms.kOptimal:
            return "OPTIMAL"
        if st in (ms.kInfeasible, ms.kUnboundedOrInfeasible):
            return "INFEASIBLE"
        if st == ms.kUnbounded:
            return "UNBOUNDED"
        if st == ms.kUnknown and getattr(self, "_crossover_off", False):
            # An interior-point run without crossover that IPX itself reports optimal comes back as
            # "Unknown" when the postsolved point needs tiny residual corrections ("basis is not valid;
            # solution is valid").  Gurobi reports such a barrier-only run as OPTIMAL; so does this.
            if self._bar_iters > 0 and self._interior_point_is_optimal():
                return "OPTIMAL"
        return "UNKNOWN"

    def _interior_point_is_optimal(self, tol: float = 1e-6) -> bool:
        """KKT certificate of an interior (x, y) that carries no basis: primal residuals, sign of the
        reduced costs on infinite bounds, and the duality gap against the bound-aware dual objective.
        (HiGHS' own infeasibility counts are meaningless here: they assume non-basic variables sit on
        a bound.)"""
        cached = getattr(self, "_ipm_certificate", None)
        if cached is not None:
            return cached
        sol = self.h.getSolution()
        ok = False
        if sol.value_valid and sol.dual_valid:
            inf = self.hc.kHighsInf
            x, y = np.array(sol.col_value), np.array(sol.row_dual)
            A = sp.csr_matrix(self._A)
            c = np.asarray(self._c, dtype=float)
            l, u = np.asarray(self._l, dtype=float), np.asarray(self._u, dtype=float)
            lo, up = self._row_lower, self._row_upper
            ax = A @ x
            scale_b = 1.0 + max(float(np.max(np.abs(lo[np.abs(lo) < inf]), initial=0.0)),
                                float(np.max(np.abs(up[np.abs(up) < inf]), initial=0.0)))
            p_res = max(float(np.max(np.maximum(lo - ax, 0.0), initial=0.0)), float(np.max(np.maximum(ax - up, 0.0), initial=0.0)),
                        float(np.max(np.maximum(l - x, 0.0), initial=0.0)), float(np.max(np.maximum(x - u, 0.0), initial=0.0)))
            rc = c - A.T @ y
            scale_c = 1.0 + float(np.max(np.abs(c), initial=0.0))
            d_res = max(float(np.max(np.where(l <= -inf, np.maximum(rc, 0.0), 0.0), initial=0.0)),
                        float(np.max(np.where(u >= inf, np.maximum(-rc, 0.0), 0.0), initial=0.0)),
                        float(np.max(np.where(lo <= -inf, np.maximum(y, 0.0), 0.0), initial=0.0)),
                        float(np.max(np.where(up >= inf, np.maximum(-y, 0.0), 0.0), initial=0.0)))
            with np.errstate(invalid="ignore"):
                dual_obj = (float(np.sum(np.where(y > 0, np.where(lo > -inf, lo, 0.0) * y, np.where(up < inf, up, 0.0) * y)))
                            + float(np.sum(np.where(rc > 0, np.where(l > -inf, l, 0.0) * rc, np.where(u < inf, u, 0.0) * rc))))
            primal_obj = float(c @ x)
            gap = abs(primal_obj - dual_obj) / (1.0 + abs(primal_obj))
            ok = p_res <= 1e-4 * scale_b and d_res <= 1e-4 * scale_c and gap <= tol
        self._ipm_certificate = ok
        return ok

    def return_x(self) -> np.ndarray:
        assert self.return_status() == "OPTIMAL", "The model is not solved to optimal!"
        return np.array(self.h.getSolution().col_value)

    def return_y(self) -> np.ndarray:
        return np.array(self.h.getSolution().row_dual)

    def return_barx(self) -> Optional[np.ndarray]:
        return self._x_bar if self._used_ipm else None

    def return_obj_val(self) -> float:
        return float(self.h.getInfo().objective_function_value)

    def return_runtime(self) -> datetime.timedelta:
        return datetime.timedelta(seconds=self._runtime)

    def return_iter_count(self) -> int:
        return self._iters

    def return_bar_iter_count(self) -> int:
        return self._bar_iters

    def return_reduced_cost(self) -> np.ndarray:
        return np.array(self.h.getSolution().col_dual)

    def return_basis(self) -> Optional[Basis]:
        if self._crossover_off:
            return None
        hb = self.h.getBasis()
        if not hb.valid:
            return None
        st = self.hc.HighsBasisStatus
        # enum -> int in one C-level pass, then a table lookup (codes of output.py: 0 / -1 / -2 / -3)
        code = np.full(1 + max(int(v) for v in (st.kBasic, st.kLower, st.kUpper, st.kZero, st.kNonbasic)), -1)
        for status, value in ((st.kBasic, 0), (st.kLower, -1), (st.kUpper, -2), (st.kZero, -3), (st.kNonbasic, -1)):
            code[int(status)] = value
        cols, rows = hb.col_status, hb.row_status
        vb = code[np.fromiter(map(int, cols), dtype=np.int64, count=len(cols))]
        rb = np.fromiter(map(int, rows), dtype=np.int64, count=len(rows))
        cb = np.where(rb == int(st.kBasic), 0, -1)
        return Basis(vb.astype(int), cb.astype(int))
