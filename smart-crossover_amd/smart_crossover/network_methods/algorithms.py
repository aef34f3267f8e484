"""Network crossover (TNET / CNET_OT / CNET_MCF) -- API of the reference's
``network_methods/algorithms.py`` (network_crossover :14, column_generation :81).

The driver is host control flow, as in the reference; the arithmetic it triggers (flow indicators,
ranking, sub-problem assembly, pricing) runs on the MI355X inside the managers.  Timing follows the
reference's definition: Output.runtime = host time of set-up and bookkeeping + the solver-reported
runtimes of the sub-problem solves.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from smart_crossover.formats import MinCostFlow, OptTransport
from smart_crossover.network_methods.net_manager import MCFManagerStd, NetworkManager, OTManager
from smart_crossover.network_methods.tree_BI import tree_basis_identify
from smart_crossover.output import Output
from smart_crossover.parameters import COLUMN_GENERATION_RATIO
from smart_crossover.solver_caller.caller import SolverSettings
from smart_crossover.timer import Timer


def network_crossover(x: np.ndarray, ot: Optional[OptTransport] = None, mcf: Optional[MinCostFlow] = None,
                      method: str = "tnet", solver: str = "GRB",
                      solver_settings: SolverSettings = SolverSettings(log_console=0)) -> Output:
    """From an inexact flow ``x`` to an optimal basic solution of an OT ('tnet', 'cnet_ot') or MCF
    ('cnet_mcf') problem by column generation over arcs ranked by their flow indicators."""
    print(f"*** Running {method} algorithm. ***")
    timer = Timer()
    timer.start_timer()
    push_iter = 0

    if method in ("tnet", "cnet_ot"):
        manager = OTManager(ot)
    elif method == "cnet_mcf":
        manager = MCFManagerStd(mcf)
    else:
        raise ValueError("Invalid method specified. Choose from 'tnet', 'cnet_ot', or 'cnet_mcf'.")

    queue, flow_indicators = manager.get_sorted_flows(x)

    if method == "tnet":
        manager.get_mcf()
        tree_basis, push_iter = tree_basis_identify(manager, flow_indicators)
        manager.set_basis(tree_basis)
        manager.add_free_variables(tree_basis.vbasis == 0)
    else:
        if method == "cnet_ot":
            manager.extend_by_bigM(manager.m * np.max(ot.M))
            manager.get_mcf()
        else:
            manager.rescale_cost(np.max(np.abs(mcf.c)))
            # every arc starts fixed (quirk Q9); mcf.c is already the rescaled cost here (quirk Q7)
            manager.fix_variables(ind_fix_to_up=np.where(x >= mcf.u / 2)[0], ind_fix_to_low=np.where(x < mcf.u / 2)[0])
            manager.extend_by_bigM(manager.m * np.max(mcf.c))
        manager.update_subproblem()
        manager.set_initial_basis()

    timer.end_timer()
    cg_output = column_generation(manager, queue, solver, solver_settings)
    total = timer.total_duration + cg_output.runtime
    print(f"*** Optimal solution found with {cg_output.iter_count + push_iter} simplex iterations in {total} seconds. ***")
    return Output(x=cg_output.x, obj_val=cg_output.obj_val, runtime=total, iter_count=cg_output.iter_count + push_iter,
                  basis=cg_output.basis)


def column_generation(net_manager: NetworkManager, queue: np.ndarray, solver: str,
                      solver_settings: SolverSettings) -> Output:
    """Release the ranked arcs in geometrically growing prefixes of ``queue`` until the sub-problem's
    optimum prices out for the whole network (network_methods/algorithms.py:81-144).

    The budget is an *absolute* position in the queue: the first round releases queue[:budget0] with
    budget0 = 10 m when n/m > 1000 else 1.2 m, every later round doubles the budget."""
    timer = Timer()
    timer.start_timer()
    left = 0
    budget = int(10 * net_manager.m) if net_manager.n / net_manager.m > 1000 else int(1.2 * net_manager.m)
    x = None
    obj_val = None
    iter_count = 0
    rounds = 1
    optimal = False
    while not optimal:
        if left >= len(queue):
            print(" ##### Column generation fails! #####")
            break
        right = min(budget, len(queue))
        net_manager.add_free_variables(queue[left:right])
        net_manager.update_subproblem()

        timer.end_timer()                                   # the solve is accounted by the solver's own clock
        sub_output = net_manager.solve_subproblem(solver, solver_settings)
        obj_val = net_manager.recover_obj_val(sub_output.obj_val)
        timer.accumulate_time(sub_output.runtime)
        timer.start_timer()

        net_manager.set_basis(net_manager.recover_basis_from_sub_basis(sub_output.basis))
        x = net_manager.recover_x_from_sub_x(sub_output.x)
        optimal = bool(net_manager.check_optimality_condition(x, sub_output.y))

        budget = int(COLUMN_GENERATION_RATIO * budget)
        left = right
        iter_count += sub_output.iter_count
        print(f"***  CG iteration {rounds} completed. ***")
        rounds += 1

    timer.end_timer()
    return Output(x=x, obj_val=obj_val, runtime=timer.total_duration, iter_count=iter_count, basis=net_manager.basis)
