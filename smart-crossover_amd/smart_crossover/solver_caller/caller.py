"""Backend-neutral solver seam (API of the reference's ``solver_caller/caller.py``).

``SolverSettings`` carries the same nine fields with the same defaults (caller.py:33-41).
``SolverCaller`` lists the operations ``solve_problem`` / ``solve_lp`` / ``solve_mcf`` drive; a
backend implements them.  Unlike the reference, this module imports no solver package: gurobipy,
cplex and mosek are commercial and optional, backends are imported lazily by
``solving.generate_solver_caller``.
"""
from __future__ import annotations

import datetime
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp

from smart_crossover.formats import GeneralLP, MinCostFlow, OptTransport, StandardLP
from smart_crossover.output import Basis, Output


@dataclass
class SolverSettings:
    presolve: str = "on"          # 'on' | 'off'
    crossover: str = "on"         # 'on' | 'off'  (barrier runs only)
    barrierTol: float = 1e-8
    optimalityTol: float = 1e-6
    timeLimit: int = 3600         # seconds
    log_file: str = ""
    log_console: int = 1
    iterLimit: int = 1000         # barrier iteration limit
    simplexPricing: str = ""      # '' | 'SE' (steepest edge) | 'PP' (partial pricing)


class SolverCaller:
    """Operations a backend provides.  Readers load a problem, ``add_warm_start_*`` seed the next
    run, ``run_*`` solve, ``return_*`` expose the result; ``return_output`` bundles them."""

    solver_name: str = "?"

    def __init__(self, solver_settings: Optional[SolverSettings] = None) -> None:
        self.settings = solver_settings if solver_settings is not None else SolverSettings()

    # -- loading -------------------------------------------------------------------------------
    def _log_summary(self, seconds: float, iters: int, bar_iters: Optional[int] = None) -> None:
        """Gurobi-style summary lines, so that the reference's log analysis keeps working on logs of any
        backend (visualization.py:31-34,345-346 parse exactly these two sentences)."""
        lines = []
        if bar_iters is not None:
            lines.append(f"Barrier solved model in {int(bar_iters)} iterations and {seconds:.2f} seconds")
        if iters is not None:
            lines.append(f"Solved in {int(iters)} iterations and {seconds:.2f} seconds")
        if self.settings.log_file:
            with open(self.settings.log_file, "a") as fh:
                fh.write("".join(line + "\n" for line in lines))
        if self.settings.log_console:
            for line in lines:
                print(line)

    def read_model_from_file(self, path: str) -> None:
        raise NotImplementedError

    def read_stdlp(self, stdlp: StandardLP) -> None:
        raise NotImplementedError

    def read_genlp(self, genlp: GeneralLP) -> None:
        raise NotImplementedError

    def read_mcf(self, mcf: MinCostFlow) -> None:
        self.read_stdlp(mcf)

    def read_ot(self, ot: OptTransport) -> None:
        self.read_mcf(ot.to_MCF())

    # -- model data ----------------------------------------------------------------------------
    def get_A(self) -> sp.csr_matrix:
        raise NotImplementedError

    def get_b(self) -> np.ndarray:
        raise NotImplementedError

    def get_sense(self) -> np.ndarray:
        raise NotImplementedError

    def get_c(self) -> np.ndarray:
        raise NotImplementedError

    def get_l(self) -> np.ndarray:
        raise NotImplementedError

    def get_u(self) -> np.ndarray:
        raise NotImplementedError

    def return_mcf(self) -> MinCostFlow:
        return MinCostFlow(self.get_A(), self.get_b(), self.get_c(), self.get_u())

    def return_stdlp(self) -> StandardLP:
        return StandardLP(A=self.get_A(), b=self.get_b(), c=self.get_c(), u=self.get_u())

    def return_genlp(self) -> GeneralLP:
        return GeneralLP(A=self.get_A(), b=self.get_b(), c=self.get_c(), l=self.get_l(), u=self.get_u(),
                         sense=self.get_sense())

    # -- warm starts ---------------------------------------------------------------------------
    def add_warm_start_basis(self, basis: Basis) -> None:
        raise NotImplementedError

    def add_warm_start_solution(self, start_solution: Tuple[np.ndarray, np.ndarray]) -> None:
        raise NotImplementedError

    # -- runs ----------------------------------------------------------------------------------
    def run_default(self) -> None:
        raise NotImplementedError

    def run_barrier(self) -> None:
        raise NotImplementedError

    def run_barrier_no_crossover(self) -> None:
        raise NotImplementedError

    def run_simplex(self) -> None:
        raise NotImplementedError

    def run_primal_simplex(self) -> None:
        raise NotImplementedError

    def run_dual_simplex(self) -> None:
        raise NotImplementedError

    def run_network_simplex(self) -> None:
        raise NotImplementedError

    def reset_model(self) -> None:
        raise NotImplementedError

    # -- results -------------------------------------------------------------------------------
    def return_x(self) -> np.ndarray:
        raise NotImplementedError

    def return_y(self) -> np.ndarray:
        raise NotImplementedError

    def return_barx(self) -> Optional[np.ndarray]:
        raise NotImplementedError

    def return_obj_val(self) -> float:
        raise NotImplementedError

    def return_runtime(self) -> datetime.timedelta:
        raise NotImplementedError

    def return_iter_count(self) -> int:
        raise NotImplementedError

    def return_bar_iter_count(self) -> int:
        raise NotImplementedError

    def return_reduced_cost(self) -> np.ndarray:
        raise NotImplementedError

    def return_basis(self) -> Optional[Basis]:
        raise NotImplementedError

    def return_status(self) -> str:
        raise NotImplementedError

    def return_output(self) -> Output:
        status = self.return_status()
        if status != "OPTIMAL":
            return Output(runtime=self.return_runtime(), status=status)
        return Output(x=self.return_x(), y=self.return_y(), x_bar=self.return_barx(), obj_val=self.return_obj_val(),
                      runtime=self.return_runtime(), iter_count=self.return_iter_count(),
                      bar_iter_count=self.return_bar_iter_count(), basis=self.return_basis(), status=status)
