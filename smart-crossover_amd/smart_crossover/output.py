"""Result containers (API of the reference's ``smart_crossover/output.py``).

Basis codes follow the Gurobi VBasis / CBasis convention the reference uses throughout
(solver_caller/gurobi.py:88-89,107-108): 0 basic, -1 non-basic at lower bound, -2 non-basic at
upper bound, -3 super-basic (free); constraints: 0 basic, -1 non-basic.
"""
from __future__ import annotations

import datetime
from dataclasses import dataclass
from typing import Optional

import numpy as np

BASIC, AT_LOWER, AT_UPPER, SUPERBASIC = 0, -1, -2, -3


@dataclass
class Basis:
    """vbasis[n] / cbasis[m] status codes; float input (e.g. ``-np.ones(n)``) is cast to int."""

    vbasis: np.ndarray
    cbasis: np.ndarray

    def __post_init__(self) -> None:
        self.vbasis = np.asarray(self.vbasis).astype(int)
        self.cbasis = np.asarray(self.cbasis).astype(int)


@dataclass(frozen=True)
class Output:
    """What a solve (or a whole crossover) hands back.  Every field is optional: a non-optimal
    solve only fills ``runtime`` and ``status`` (reference solver_caller/caller.py:164-169)."""

    x: Optional[np.ndarray] = None               # primal vertex
    y: Optional[np.ndarray] = None               # dual vertex
    x_bar: Optional[np.ndarray] = None           # primal interior point (barrier runs)
    obj_val: Optional[float] = None
    runtime: Optional[datetime.timedelta] = None
    iter_count: Optional[float] = None           # simplex (push) iterations
    bar_iter_count: Optional[int] = None
    rcost: Optional[np.ndarray] = None
    basis: Optional[Basis] = None
    status: Optional[str] = None                 # 'OPTIMAL' | 'INFEASIBLE' | 'UNBOUNDED' | 'UNKNOWN'

    def __str__(self) -> str:
        return (f"Output(obj_val={self.obj_val}, runtime={self.runtime}, iter_count={self.iter_count}, "
                f"bar_iter_count={self.bar_iter_count})")
