"""Network crossover (TNET / CNET_OT / CNET_MCF) -- API of the reference's
``network_methods/algorithms.py`` (network_crossover :14, column_generation :81).

The driver is host control flow, as in the reference; the arithmetic it triggers (flow indicators,
ranking, spanning tree, sub-problem assembly, pricing) runs on the MI355X inside the managers.  Timing
follows the reference's definition: Output.runtime = host time of set-up and bookkeeping + the
solver-reported runtimes of the sub-problem solves.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, Optional, Tuple

import numpy as np

from smart_crossover.formats import MinCostFlow, OptTransport
from smart_crossover.network_methods.net_manager import MCFManagerStd, NetworkManager, OTManager
from smart_crossover.network_methods.tree_BI import tree_basis_identify
from smart_crossover.output import Output
from smart_crossover.parameters import COLUMN_GENERATION_RATIO
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.timer import Timer


# ------------------------------------------------------------------------------------------------
# starting points of the three methods: each returns (manager, queue, push iterations)
# ------------------------------------------------------------------------------------------------
def _start_tnet(x: np.ndarray, ot: OptTransport, mcf: Optional[MinCostFlow]):
    """TNET: spanning-tree basis from the flow indicators, its arcs released at once."""
    manager = OTManager(ot)
    queue, indicators = manager.get_sorted_flows(x)
    manager.get_mcf()
    basis, pushes = tree_basis_identify(manager, indicators)
    manager.set_basis(basis)
    manager.add_free_variables(basis.vbasis == 0)            # quirk Q10: a boolean mask on the flat problem
    return manager, queue, pushes


def _start_cnet_ot(x: np.ndarray, ot: OptTransport, mcf: Optional[MinCostFlow]):
    """CNET on transport problems: big-M row/column, artificial basis."""
    manager = OTManager(ot)
    queue, _ = manager.get_sorted_flows(x)
    manager.extend_by_bigM(manager.m * np.max(ot.M))
    manager.get_mcf()
    manager.update_subproblem()
    manager.set_initial_basis()
    return manager, queue, 0


def _start_cnet_mcf(x: np.ndarray, ot: Optional[OptTransport], mcf: MinCostFlow):
    """CNET on min-cost flow: costs rescaled on the caller's object (quirk Q7), every arc fixed to the
    nearer bound first (quirk Q9), big-M artificials on top."""
    manager = MCFManagerStd(mcf)
    queue, _ = manager.get_sorted_flows(x)
    manager.rescale_cost(np.max(np.abs(mcf.c)))
    half = mcf.u / 2
    manager.fix_variables(ind_fix_to_up=np.where(x >= half)[0], ind_fix_to_low=np.where(x < half)[0])
    manager.extend_by_bigM(manager.m * np.max(mcf.c))        # mcf.c is the rescaled cost by now
    manager.update_subproblem()
    manager.set_initial_basis()
    return manager, queue, 0


_START: Dict[str, Callable] = {"tnet": _start_tnet, "cnet_ot": _start_cnet_ot, "cnet_mcf": _start_cnet_mcf}


def network_crossover(x: np.ndarray, ot: Optional[OptTransport] = None, mcf: Optional[MinCostFlow] = None,
                      method: str = "tnet", solver: str = "GRB",
                      solver_settings: SolverSettings = SolverSettings(log_console=0)) -> Output:
    """From an inexact flow ``x`` to an optimal basic solution of an OT ('tnet', 'cnet_ot') or MCF
    ('cnet_mcf') problem by column generation over arcs ranked by their flow indicators."""
    print(f"*** Running {method} algorithm. ***")
    clock = Timer()
    clock.start_timer()
    start = _START.get(method)
    if start is None:
        raise ValueError("Invalid method specified. Choose from 'tnet', 'cnet_ot', or 'cnet_mcf'.")
    manager, queue, pushes = start(x, ot, mcf)
    clock.end_timer()

    generated = column_generation(manager, queue, solver, solver_settings)
    runtime = clock.total_duration + generated.runtime
    iterations = generated.iter_count + pushes
    print(f"*** Optimal solution found with {iterations} simplex iterations in {runtime} seconds. ***")
    return Output(x=generated.x, obj_val=generated.obj_val, runtime=runtime, iter_count=iterations, basis=generated.basis)


# ------------------------------------------------------------------------------------------------
# column generation
# ------------------------------------------------------------------------------------------------
def _release_schedule(m: int, n: int, queue_len: int) -> Iterator[Tuple[int, int]]:
    """Prefixes of the queue released round by round (network_methods/algorithms.py:102,111-114,135):
    budgets are *absolute* positions -- 10 m when n/m > 1000 else 1.2 m for the first round, doubled
    after every round -- and a round whose left end has run off the queue ends the schedule."""
    budget = int(10 * m) if n / m > 1000 else int(1.2 * m)
    left = 0
    while left < queue_len:
        right = min(budget, queue_len)
        yield left, right
        left = right
        budget = int(COLUMN_GENERATION_RATIO * budget)


def column_generation(net_manager: NetworkManager, queue: np.ndarray, solver: str,
                      solver_settings: SolverSettings) -> Output:
    """Release the ranked arcs in geometrically growing prefixes of ``queue`` until the sub-problem's
    optimum prices out for the whole network (network_methods/algorithms.py:81-144)."""
    clock = Timer()
    clock.start_timer()
    x = obj_val = None
    simplex_iterations = 0
    priced_out = False
    for round_no, (left, right) in enumerate(_release_schedule(net_manager.m, net_manager.n, len(queue)), start=1):
        net_manager.add_free_variables(queue[left:right])
        net_manager.update_subproblem()

        clock.end_timer()                                   # the solve is accounted by the solver's own clock
        solved = net_manager.solve_subproblem(solver, solver_settings)
        obj_val = net_manager.recover_obj_val(solved.obj_val)
        clock.accumulate_time(solved.runtime)
        clock.start_timer()

        net_manager.set_basis(net_manager.recover_basis_from_sub_basis(solved.basis))
        x = net_manager.recover_x_from_sub_x(solved.x)
        priced_out = bool(net_manager.check_optimality_condition(x, solved.y))
        simplex_iterations += solved.iter_count
        print(f"***  CG iteration {round_no} completed. ***")
        if priced_out:
            break
    if not priced_out:
        print(" ##### Column generation fails! #####")
    clock.end_timer()
    return Output(x=x, obj_val=obj_val, runtime=clock.total_duration, iter_count=simplex_iterations,
                  basis=net_manager.basis)
