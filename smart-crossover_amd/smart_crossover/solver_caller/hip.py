"""``HipCaller``: the device solvers of libsxhip.so (kernel groups K16*) behind the ``SolverCaller``
seam -- ``solver="HIP"`` in ``solve_lp`` / ``solve_mcf`` / ``solve_ot`` / ``network_crossover``.

It serves the *simplex family* of methods ('default', 'simplex', 'primal_simplex', 'dual_simplex',
'network_simplex') with Gurobi-style warm bases, i.e. the re-solves the crossover algorithms issue with a warm
start (lp_methods/algorithms.py:69-74, network_methods/net_manager.py:222,468): network sub-problems go to the dual /
primal network simplex (K16d / K16n), general LPs from ``BAND_MIN_ROWS`` rows on to the sparse crossover started from
the given basis (K16s: bordered band factorisation of that very basis, ``sx_crossover_band_basis_dev``), everything
else -- and whatever K16s refuses -- to the bounded primal simplex on a dense inverse (K16, 8 m^2 bytes: about 1.7e5
rows on an empty MI355X).

'barrier' with crossover -- the re-solve of the perturbed sub-problem, lp_methods/algorithms.py:50-54 -- is served
too, by a substitute: the reference's solvers run an interior-point method plus their own crossover there and ignore
the warm start they are handed (quirk Q6); all the caller consumes is the optimal *vertex* and its basis, and the
perturbed LP has a unique one.  The device takes the interior point it is handed (``crash_from_warm_start``:
``solve_problem`` passes the warm start on to backends that ask for it), carries it towards the perturbed optimum by a
first-order stage (K16p, ``sx_pdlp_dev``) and crosses over from there: K16s from ``BAND_MIN_ROWS`` rows on (basis guessed
from the margins of the point, bordered band factorisation, tableau of the columns that can still move), the dense
K16 (``sx_simplex_crossover_dev`` from a crash basis: columns strictly between their bounds and slacks of inactive rows
basic, the rest superbasic) below that size or when K16s answers "unsupported".  'barrier' *without* crossover (the
initial solve, whose output is the interior point itself) is not available: pair the device with a barrier-capable
backend through the composite name ``"<barrier backend>+HIP"`` (e.g. ``"HGS+HIP"``).
"""
from __future__ import annotations

import datetime
import os
import time
from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import GeneralLP, StandardLP
from smart_crossover.output import Basis
from smart_crossover.solver_caller.caller import SolverCaller, SolverSettings
from smart_crossover.hip.lib import SxError as _SxError

_STATUS = {0: "OPTIMAL", 1: "INFEASIBLE", 2: "UNBOUNDED"}


PDLP_ITERS = 20000       # first-order stage in front of the crossover ('barrier' runs): iteration limit (0: skip it)
PDLP_TOL = 1e-9          # ... and its relative KKT tolerance


def _pdlp_iterations(m: int, to_band: bool) -> int:
    """Iterations of the first-order stage.  In front of the dense crossover the full budget: below ~2e4 iterations its
    crash basis is no better than without the stage (config 2: 238,949 pivots after 5,000 iterations, 692 after 20,000).
    In front of the sparse crossover a pivot is cheap at 1e5 rows (0.1 ms against 16 us per iteration: 5,000 iterations
    and 1,105 pivots beat 20,000 and 837 by 0.17 s) and dear at 1e6 (1-2 ms per pivot: 5,000 iterations leave 13,022
    pivots, 70 s against 17 s) -- so the budget grows with the rows: m / 20 between 5,000 and PDLP_ITERS.  Measured on
    workloads.netlib_lp only (profiles/r03/lp_1e6.md); SX_PDLP_ITERS overrides."""
    if not to_band:
        return PDLP_ITERS
    return max(5000, min(PDLP_ITERS, m // 20))
BAND_MIN_ROWS = 2000     # from this many rows on the crossover behind the first-order stage (and the warm-started simplex) is the
                         # sparse one (K16s: bordered band factorisation + tableau of the tracked columns) when the basis has
                         # that structure: 0.054 s against 0.078 s for the dense K16 at 3,000 rows, 0.12 s against 1.03 s at
                         # 20,000 (netlib_lp, profiles/r04/lp_1e6.md); an unstructured basis is refused after the matching
NETDUAL_FEAS_TOL = 1e-9  # bound violation of a tree arc the dual network simplex still calls feasible
CRASH_MARGIN = 1e-6      # a column this far (relative) inside its bounds / a slack this large counts as basic


class HipCaller(SolverCaller):
    solver_name = "HIP"
    crash_from_warm_start = True     # solve_problem hands a 'barrier' run its warm start (see module docstring)

    def __init__(self, solver_settings: Optional[SolverSettings] = None) -> None:
        super().__init__(solver_settings)
        self._warm: Optional[Basis] = None
        self._warm_point: Optional[Tuple[np.ndarray, np.ndarray]] = None
        self._res = None
        self._runtime = 0.0
        self._x = self._y = self._vb = self._cb = None

    # -- loading -------------------------------------------------------------------------------
    def _load(self, A, b, c, l, u, row_lt) -> None:
        self._A = sp.csr_matrix(A)
        self._b, self._c = np.asarray(b, float), np.asarray(c, float)
        self._l, self._u = np.asarray(l, float), np.asarray(u, float)
        self._row_lt = np.asarray(row_lt, dtype=np.uint8)
        self._warm = None
        self._warm_point = None
        self._res = None
        self._network = False

    def read_genlp(self, genlp: GeneralLP) -> None:
        self._load(genlp.A, genlp.b, genlp.c, genlp.l, genlp.u, np.asarray(genlp.sense) == "<")
        # the LP's device residency, validated against the matrix it holds NOW: resident_for() rebuilds it
        # when lp.A was rebound since the last device call (a stale copy would silently solve the old matrix)
        from smart_crossover.hip.resident import resident_for
        self._resident = resident_for(genlp) if getattr(genlp, "_sx_resident", None) is not None else None
        self._col_ids = self._session_holder = self._dev_matrix = None

    def read_stdlp(self, stdlp: StandardLP) -> None:
        self._load(stdlp.A, stdlp.b, stdlp.c, stdlp.l, stdlp.u, np.zeros(np.asarray(stdlp.b).size, dtype=bool))
        self._resident = None
        # column-generation sequences (network managers) tag their sub-problems: stable column identifiers
        # and a holder in which the device simplex keeps its basis inverse from one round to the next
        self._col_ids = getattr(stdlp, "col_ids", None)
        self._session_holder = getattr(stdlp, "hip_session", None)
        self._dev_matrix = getattr(stdlp, "_sx_device_matrix", None)

    def read_mcf(self, mcf) -> None:
        # a MinCostFlow is a network by construction (formats.py:104-121); sx_netsimplex_dev checks the columns
        self.read_stdlp(mcf)
        self._network = True

    def get_A(self) -> sp.csr_matrix:
        return self._A

    def get_b(self) -> np.ndarray:
        return self._b

    def get_sense(self) -> np.ndarray:
        return np.where(self._row_lt.astype(bool), "<", "=")

    def get_c(self) -> np.ndarray:
        return self._c

    def get_l(self) -> np.ndarray:
        return self._l

    def get_u(self) -> np.ndarray:
        return self._u

    # -- warm starts ---------------------------------------------------------------------------
    def add_warm_start_basis(self, basis: Basis) -> None:
        self._warm = basis

    def add_warm_start_solution(self, start_solution: Tuple[np.ndarray, np.ndarray]) -> None:
        # a primal simplex starts from a basis, not from a point: the point only serves the crash of run_barrier
        self._warm_point = start_solution

    def _crash_basis(self, ctx, dA) -> Optional[Basis]:
        """Basis guess from the interior point handed over as warm start.  Candidates are the variables that
        sit strictly inside their bounds: structural j with margin min(x_j - l_j, u_j - x_j), and the slack of a
        '<' row with margin b_i - (A x)_i (row scoring kernel K2 on the device).  A vertex has room for m of
        them; the perturbed objective of the crossover weighs a column by 1 / margin
        (lp_methods/algorithms.py:148), so the ones it drives to a bound are, by and large, those with the
        smallest margins: the m candidates with the largest margins are proposed as basic (codes 0), the other
        interior ones stay superbasic at their value (-3 / a free row), everything else sits at its nearer bound."""
        if self._warm_point is None:
            return None
        x = np.asarray(self._warm_point[0], dtype=np.float64)
        m, n = self._A.shape
        if x.size != n:
            return None
        with np.errstate(invalid="ignore"):
            lo_gap, up_gap = x - self._l, self._u - x
        margin = np.minimum(lo_gap, up_gap)
        free = np.isinf(self._l) & np.isinf(self._u)
        interior = (margin > CRASH_MARGIN * (1.0 + np.abs(x))) | free
        s_p = ctx.empty(m, np.float64)
        ctx.score_rows(dA, ctx.to_device(x), ctx.to_device(self._b), None, 0.0, s_p, None)
        slack = s_p.download()
        slack_in = self._row_lt.astype(bool) & (slack > CRASH_MARGIN * (1.0 + np.abs(self._b)))
        score = np.concatenate([np.where(free, np.inf, np.where(interior, margin, -np.inf)),
                                np.where(slack_in, slack, -np.inf)])
        n_cand = int(np.count_nonzero(score > -np.inf))
        thr = -np.inf
        if n_cand > m:                     # more interior variables than a vertex can hold: keep the m largest
            thr = np.partition(score, score.size - m)[score.size - m]
        vb = np.where(lo_gap <= up_gap, -1, -2).astype(np.int64)
        vb[interior] = -3
        vb[interior & (score[:n] >= thr)] = 0
        cb = np.where(slack_in & (score[n:] >= thr), 0, -1).astype(np.int64)
        return Basis(vb, cb)

    # -- runs ----------------------------------------------------------------------------------
    def _solve(self) -> None:
        from smart_crossover.hip.device import default_context
        from smart_crossover.hip.resident import matrix_fingerprint
        ctx = default_context()
        m, n = self._A.shape
        res = getattr(self, "_resident", None)
        given = getattr(self, "_dev_matrix", None)
        if given is not None and given.shape == (m, n) and given.handle is not None and given.ctx is ctx:
            dA, own = given, False                      # built on the device by the manager's gather
        elif (res is not None and res.A.shape == (m, n) and res.A.handle is not None
              and res.fingerprint == matrix_fingerprint(self._A)):
            dA, own = res.A, False
        else:
            dA, own = ctx.matrix(self._A), True
        put = lambda v: ctx.to_device(np.ascontiguousarray(v, dtype=np.float64))   # noqa: E731
        d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
        d_vb, d_cb = ctx.empty(n, np.int8), ctx.empty(m, np.int8)
        vb_in = cb_in = x_start = None
        self._res = None
        t0 = time.perf_counter()
        d_b, d_c, d_l, d_u = put(self._b), put(self._c), put(self._l), put(self._u)
        self.pdlp = None
        if self._warm is None and getattr(self, "_want_crash", False):
            mode = os.environ.get("SX_LP_CROSSOVER", "auto")
            to_band = mode == "band" or (mode == "auto" and m >= BAND_MIN_ROWS)
            if (to_band and mode == "auto" and m < 200_000 and self._warm_point is not None
                    and np.asarray(self._warm_point[0]).size == n):
                # the dense crossover needs the full first-order budget in front of it: ask the sparse one now whether it
                # will take this LP (one kernel when no column of A is taller than a band the LU takes; else its matching on
                # the point at hand: ~20 ms at 1e5 rows; at 1e6 rows the question is not asked -- a refusal there lets the
                # first-order stage go on to the full budget instead)
                x_probe = put(np.clip(np.asarray(self._warm_point[0], dtype=np.float64), self._l, self._u))
                to_band = ctx.crossover_band_takes(dA, d_b, d_c, d_l, d_u, ctx.to_device(self._row_lt), x_probe)
            iters = int(os.environ.get("SX_PDLP_ITERS", _pdlp_iterations(m, to_band)))
            if self._warm_point is not None and iters > 0 and np.asarray(self._warm_point[0]).size == n:
                # what the reference's backends do with 'barrier' before their crossover: carry the interior point
                # of the original LP next to the optimum of THIS (perturbed) LP -- first-order stage K16p
                x0 = np.clip(np.asarray(self._warm_point[0], dtype=np.float64), self._l, self._u)
                y0 = np.asarray(self._warm_point[1], dtype=np.float64)
                y0 = np.where(self._row_lt.astype(bool), np.minimum(y0, 0.0), y0) if y0.size == m else np.zeros(m)
                d_px, d_py = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
                if to_band and m >= 200_000:
                    # the sparse crossover's eta file (<= 16 GB) + tableau: allocated while the first-order stage runs
                    # (hipMalloc of 28 GB takes 0.5-1.4 s; sx_crossover_band_dev finds the block in the context)
                    mp = 1.012 * m + 20000
                    ctx.prefetch_block(int(min(16e9, 8.0 * mp * 20000) + 8.0 * mp * (0.0016 * m + 700)))
                self.pdlp = ctx.pdlp(dA, d_b, d_c, d_l, d_u, ctx.to_device(self._row_lt), put(x0), put(y0), iters,
                                     float(os.environ.get("SX_PDLP_TOL", PDLP_TOL)), d_px, d_py)
                self._warm_point = (d_px.download(), d_py.download())
                self.pdlp_seconds = time.perf_counter() - t0
                if to_band:      # ("dense": K16 whatever the size)
                    self._res = self._band(ctx, dA, d_b, d_c, d_l, d_u, d_px, None, None, d_x, d_y, d_vb, d_cb)
                    full = int(os.environ.get("SX_PDLP_ITERS", PDLP_ITERS))
                    if self._res is None and iters < full:
                        # the sparse crossover does not take this LP (no band structure: config 2) and the dense one needs the
                        # point the FULL first-order budget leaves (239,796 pivots behind 5,000 iterations, 692 behind 20,000):
                        # the stage goes on from where it stopped
                        self.pdlp = ctx.pdlp(dA, d_b, d_c, d_l, d_u, ctx.to_device(self._row_lt), put(self._warm_point[0]),
                                             put(self._warm_point[1]), full - iters,
                                             float(os.environ.get("SX_PDLP_TOL", PDLP_TOL)), d_px, d_py)
                        self._warm_point = (d_px.download(), d_py.download())
                        self.pdlp_seconds = time.perf_counter() - t0
            if self._res is None:
                self._warm = self._crash_basis(ctx, dA)
            if self._warm is not None:          # crossover: start AT the interior point (superbasic columns)
                x_start = ctx.to_device(np.ascontiguousarray(self._warm_point[0], dtype=np.float64))
        if self._warm is not None and self._warm.vbasis.size == n and self._warm.cbasis.size == m:
            vb_in = ctx.to_device(np.clip(self._warm.vbasis, -3, 0).astype(np.int8))
            cb_in = ctx.to_device(np.clip(self._warm.cbasis, -1, 0).astype(np.int8))
            mode = os.environ.get("SX_LP_CROSSOVER", "auto")
            if (self._res is None and x_start is None and not self._network and self._session_holder is None
                    and (mode == "band" or (mode == "auto" and m >= BAND_MIN_ROWS))):
                # the warm-started simplex of the reference's last step (lp_methods/algorithms.py:69-74) at a size where the
                # dense inverse is out of reach: the given basis is factored as it is (bordered band form), superbasic
                # columns (-3) sit at the warm point's value
                xs = self._warm_point[0] if self._warm_point is not None and np.asarray(self._warm_point[0]).size == n else np.zeros(n)
                xs = np.clip(np.asarray(xs, dtype=np.float64), self._l, self._u)
                xs = np.where(np.isfinite(xs), xs, 0.0)
                self._res = self._band(ctx, dA, d_b, d_c, d_l, d_u, put(xs), vb_in, cb_in, d_x, d_y, d_vb, d_cb)
        session, col_ids = None, getattr(self, "_col_ids", None)
        holder = getattr(self, "_session_holder", None)
        if holder is not None and col_ids is not None:
            if getattr(holder, "session", None) is None:
                holder.session = ctx.simplex_session()
            session = holder.session
        if self._res is None:
            self.solved_by = "simplex"
        if self._res is None and self._network and vb_in is not None and x_start is None and not self._row_lt.any():
            # a network sub-problem with a warm tree basis.  First the dual method on the whole GPU (K16d: the
            # tree need not be primal feasible, new arcs at the wrong bound are flipped); status 5 = outside its
            # domain (an uncapacitated arc would have to flip, A is no incidence matrix, not a tree).  Then the
            # primal network simplex (K16n: needs a primal feasible tree); status 5 again: general simplex below
            res = ctx.net_dual(dA, d_b, d_c, d_l, d_u, vb_in, cb_in, 0, NETDUAL_FEAS_TOL, d_x, d_y, d_vb, d_cb)
            if int(res.status) in (0, 1):       # (3 = its iteration limit: the primal method takes over from vb_in)
                self._res, self.solved_by = res, "netdual"
            else:
                res = ctx.net_simplex(dA, d_b, d_c, d_l, d_u, vb_in, cb_in, 0, 1e-7, float(self.settings.optimalityTol),
                                      d_x, d_y, d_vb, d_cb)
                if int(res.status) in (0, 1, 2):     # (3, 4, 5: the general simplex below takes over from vb_in / cb_in)
                    self._res, self.solved_by = res, "netsimplex"
        if self._res is None:
            self._res = ctx.simplex(dA, d_b, d_c, d_l, d_u, ctx.to_device(self._row_lt),
                                    vb_in, cb_in, 0, 1e-7, float(self.settings.optimalityTol), d_x, d_y, d_vb, d_cb,
                                    session=None if x_start is not None else session,
                                    col_ids=None if x_start is not None else col_ids, x_start=x_start)
        self._runtime = time.perf_counter() - t0
        self._x, self._y = d_x.download(), d_y.download()
        self._vb, self._cb = d_vb.download().astype(int), d_cb.download().astype(int)
        if own:
            dA.free()
        if int(self._res.status) not in _STATUS:
            # no usable vertex: say so here instead of letting the caller trip over a missing x / basis
            why = {3: "hit the iteration limit", 4: "lost numerical accuracy (the basis inverse could not be repaired)"}
            raise RuntimeError(f"HIP simplex {why.get(int(self._res.status), 'failed')} after {int(self._res.iters)} pivots "
                               f"on a {m} x {n} problem (max bound violation {float(self._res.max_violation):.2e}); "
                               "use solver='HGS' for this instance")
        self._log_summary(self._runtime, int(self._res.iters))

    def _band(self, ctx, dA, d_b, d_c, d_l, d_u, d_start, vb_in, cb_in, d_x, d_y, d_vb, d_cb):
        """The sparse crossover (K16s); None when it does not apply or did not finish and the dense path may still do it."""
        m = self._A.shape[0]
        dense_fits = 8.0 * m * m < 1.2e11
        try:
            res = ctx.crossover_band(dA, d_b, d_c, d_l, d_u, ctx.to_device(self._row_lt), d_start, 0, 1e-7,
                                     float(self.settings.optimalityTol), d_x, d_y, d_vb, d_cb, vbasis_in=vb_in, cbasis_in=cb_in)
        except NotImplementedError:
            return None                  # no band structure / border too large: the dense crossover
        except (ValueError, MemoryError, _SxError) as exc:
            if dense_fits:
                return None
            raise RuntimeError(f"the sparse crossover failed on a {m}-row problem ({exc}) and the dense basis inverse "
                               "does not fit the device; use solver='HGS' for this instance") from exc
        self.solved_by = "crossover_band"
        if int(res.status) not in _STATUS and dense_fits:
            return None                  # iteration limit / numerical trouble: the dense crossover may still do it
        return res

    def run_default(self) -> None:
        self._solve()

    run_simplex = run_primal_simplex = run_dual_simplex = run_network_simplex = run_default

    def run_barrier(self) -> None:
        """'barrier' + crossover: first-order stage from the warm point, then the sparse (K16s) or dense (K16) crossover
        to the vertex and its basis (module docstring)."""
        self._want_crash = True
        try:
            self._solve()
        finally:
            self._want_crash = False

    def run_barrier_no_crossover(self) -> None:
        raise NotImplementedError("the HIP backend has no interior-point method: an interior point (barrier without "
                                  "crossover) has to come from another backend, e.g. solver='HGS+HIP'")

    def reset_model(self) -> None:
        self._res = None
        self._warm = None

    # -- results -------------------------------------------------------------------------------
    def return_status(self) -> str:
        return _STATUS.get(int(self._res.status), "UNKNOWN") if self._res is not None else "UNKNOWN"

    def return_x(self) -> np.ndarray:
        assert self.return_status() == "OPTIMAL", "The model is not solved to optimal!"
        return self._x

    def return_y(self) -> np.ndarray:
        return self._y

    def return_barx(self) -> Optional[np.ndarray]:
        return None

    def return_obj_val(self) -> float:
        return float(self._res.obj)

    def return_runtime(self) -> datetime.timedelta:
        return datetime.timedelta(seconds=self._runtime)

    def return_iter_count(self) -> int:
        return int(self._res.iters)

    def return_bar_iter_count(self) -> int:
        return 0

    def return_reduced_cost(self) -> np.ndarray:
        return self._c - self._A.T @ self._y

    def return_basis(self) -> Optional[Basis]:
        return Basis(self._vb, self._cb)


class SplitCaller(SolverCaller):
    """``"<barrier backend>+<simplex backend>"``: barrier runs on the first, everything else on the second."""

    def __init__(self, bar: SolverCaller, spx: SolverCaller, name: str) -> None:
        super().__init__(bar.settings)
        self.solver_name = name
        self._bar, self._spx = bar, spx
        self._active = spx

    def read_genlp(self, genlp):
        self._bar.read_genlp(genlp)
        self._spx.read_genlp(genlp)

    def read_stdlp(self, stdlp):
        self._bar.read_stdlp(stdlp)
        self._spx.read_stdlp(stdlp)

    def read_mcf(self, mcf):
        # both backends get to see that the problem is a network (HipCaller.read_mcf selects K16d / K16n)
        self._bar.read_mcf(mcf)
        self._spx.read_mcf(mcf)

    def read_ot(self, ot):
        self._bar.read_ot(ot)
        self._spx.read_ot(ot)

    def add_warm_start_basis(self, basis):
        self._spx.add_warm_start_basis(basis)

    def add_warm_start_solution(self, start_solution):
        self._spx.add_warm_start_solution(start_solution)

    @property
    def crash_from_warm_start(self):
        return getattr(self._spx, "crash_from_warm_start", False)

    def run_barrier(self):
        # vertex wanted: a simplex backend that can start from the interior point it is handed takes it
        if self.crash_from_warm_start:
            self._active = self._spx
            self._spx.run_barrier()
        else:
            self._active = self._bar
            self._bar.run_barrier()

    def run_barrier_no_crossover(self):
        self._active = self._bar
        self._bar.run_barrier_no_crossover()

    def _spx_run(self, name):
        self._active = self._spx
        getattr(self._spx, name)()

    def run_default(self):
        self._spx_run("run_default")

    def run_simplex(self):
        self._spx_run("run_simplex")

    def run_primal_simplex(self):
        self._spx_run("run_primal_simplex")

    def run_dual_simplex(self):
        self._spx_run("run_dual_simplex")

    def run_network_simplex(self):
        self._spx_run("run_network_simplex")

    def return_output(self):
        return self._active.return_output()

    def return_status(self):
        return self._active.return_status()
