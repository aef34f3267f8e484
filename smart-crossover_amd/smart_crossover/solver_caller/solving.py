"""String-dispatched solve helpers (API of the reference's ``solver_caller/solving.py``):
``generate_solver_caller``, ``solve_problem``, ``solve_lp``, ``solve_mcf``, ``solve_ot``.

Backends: 'HGS' (HiGHS, bundled with scipy -- the stand-in used wherever the reference would call a
commercial solver) and, when their Python packages are installed, the reference's own
'GRB' / 'CPL' / 'MSK' adaptors may be registered with ``register_backend``.  Error behaviour follows
the reference: unknown solver / method / LP type raise ValueError with the same messages.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np

from smart_crossover.formats import GeneralLP, MinCostFlow, OptTransport, StandardLP
from smart_crossover.output import Basis, Output
from smart_crossover.solver_caller.caller import SolverCaller, SolverSettings

_SIMPLEX_LIKE = ("default", "simplex", "network_simplex", "primal_simplex", "dual_simplex")
_BACKENDS: Dict[str, Callable[[SolverSettings], SolverCaller]] = {}


def register_backend(name: str, factory: Callable[[SolverSettings], SolverCaller]) -> None:
    """Make ``solver=name`` available to solve_lp / solve_mcf / the crossover entry points."""
    _BACKENDS[name] = factory


def _highs(settings: SolverSettings) -> SolverCaller:
    from smart_crossover.solver_caller.highs import HgsCaller
    return HgsCaller(settings)


def _hip(settings: SolverSettings) -> SolverCaller:
    from smart_crossover.solver_caller.hip import HipCaller
    return HipCaller(settings)


register_backend("HGS", _highs)
register_backend("HIP", _hip)


def generate_solver_caller(solver: str = "GRB", solver_settings: Optional[SolverSettings] = None) -> SolverCaller:
    settings = solver_settings if solver_settings is not None else SolverSettings()
    if solver in _BACKENDS:
        return _BACKENDS[solver](settings)
    if "+" in solver:      # "<barrier backend>+<simplex backend>", e.g. "HGS+HIP"
        bar_name, spx_name = solver.split("+", 1)
        from smart_crossover.solver_caller.hip import SplitCaller
        return SplitCaller(generate_solver_caller(bar_name, settings), generate_solver_caller(spx_name, settings), solver)
    if solver in ("GRB", "CPL", "MSK"):
        package = {"GRB": "gurobipy", "CPL": "cplex", "MSK": "mosek"}[solver]
        raise ImportError(f"solver '{solver}' needs the commercial package '{package}', which is not part of this "
                          f"build; use solver='HGS' or register an adaptor with register_backend('{solver}', ...)")
    raise ValueError("Invalid solver specified. Choose from 'GRB', 'CPL' and 'MSK'.")


def solve_problem(solver_caller: SolverCaller, method: str, settings: SolverSettings,
                  warm_start_basis: Optional[Basis] = None,
                  warm_start_solution: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> Output:
    """Run one method on a loaded problem.  Warm starts are applied for the simplex family only; a
    'barrier' run ignores them (quirk Q6 of the reference, solving.py:46-65)."""
    if method in _SIMPLEX_LIKE:
        if warm_start_solution is not None:
            solver_caller.add_warm_start_solution(warm_start_solution)
        if warm_start_basis is not None:
            solver_caller.add_warm_start_basis(warm_start_basis)
        {"default": solver_caller.run_default, "simplex": solver_caller.run_simplex,
         "network_simplex": solver_caller.run_network_simplex, "primal_simplex": solver_caller.run_primal_simplex,
         "dual_simplex": solver_caller.run_dual_simplex}[method]()
    elif method == "barrier":
        # extension for backends outside the reference's three: one that builds its vertex from the interior
        # point (solver_caller/hip.py) is handed the warm start; the reference's backends never are (Q6)
        if warm_start_solution is not None and getattr(solver_caller, "crash_from_warm_start", False):
            solver_caller.add_warm_start_solution(warm_start_solution)
        if settings.crossover == "on":
            solver_caller.run_barrier()
        else:
            solver_caller.run_barrier_no_crossover()
    else:
        raise ValueError("Invalid method specified. Choose from 'default' or 'barrier'.")
    return solver_caller.return_output()


def solve_lp(lp: Union[GeneralLP, StandardLP], solver: str = "GRB", method: str = "default",
             settings: Optional[SolverSettings] = None, warm_start_basis: Optional[Basis] = None,
             warm_start_solution: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> Output:
    settings = settings if settings is not None else SolverSettings()
    caller = generate_solver_caller(solver, settings)
    if isinstance(lp, StandardLP):
        caller.read_stdlp(lp)
    elif isinstance(lp, GeneralLP):
        caller.read_genlp(lp)
    else:
        raise ValueError("Invalid LP format.")
    return solve_problem(caller, method, settings, warm_start_basis, warm_start_solution)


def solve_mcf(mcf: MinCostFlow, solver: str = "GRB", method: str = "default",
              settings: Optional[SolverSettings] = None, warm_start_basis: Optional[Basis] = None) -> Output:
    settings = settings if settings is not None else SolverSettings()
    caller = generate_solver_caller(solver, settings)
    caller.read_mcf(mcf)
    return solve_problem(caller, method, settings, warm_start_basis)


def solve_ot(ot: OptTransport, solver: str = "GRB", method: str = "default",
             settings: Optional[SolverSettings] = None, warm_start_basis: Optional[Basis] = None) -> Output:
    settings = settings if settings is not None else SolverSettings()
    caller = generate_solver_caller(solver, settings)
    caller.read_ot(ot)
    return solve_problem(caller, method, settings, warm_start_basis)
