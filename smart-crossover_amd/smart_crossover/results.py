"""Result files and summary statistics either side of the crossover paths.

* pickle result files of the reference's drivers (``filehandling.py:101-111``: ``results/<path>`` under the
  project root; here the root is an explicit argument instead of a walk up from the current directory)
* the summary statistics its analysis derives from them: geometric means with the 3600 s penalty of the LP
  tables (``visualization.py:181-195``) and the shifted geometric mean by instance group of the network
  tables (``visualization.py:415,428``: ``exp(mean(log(x + 0.01)))``, rounded to two decimals)

Host-side bookkeeping only: nothing here touches the device.
"""
from __future__ import annotations

import os
import pickle
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import numpy as np

TIME_LIMIT_PENALTY = 3600.0     # seconds charged for a missing runtime (visualization.py:185-186)
GEOMEAN_SHIFT = 0.01            # visualization.py:415,428


def _results_dir(root: Optional[str]) -> str:
    return os.path.join(root if root is not None else os.getcwd(), "results")


def write_results_to_pickle(results: Any, path: str, root: Optional[str] = None) -> None:
    """``pickle.dump(results)`` into ``<root>/results/<path>`` (filehandling.py:108-111)."""
    full = os.path.join(_results_dir(root), path)
    os.makedirs(os.path.dirname(full), exist_ok=True)
    with open(full, "wb") as fh:
        pickle.dump(results, fh)


def read_results_from_pickle(path: str, root: Optional[str] = None) -> Any:
    """Inverse of :func:`write_results_to_pickle` (filehandling.py:101-105)."""
    with open(os.path.join(_results_dir(root), path), "rb") as fh:
        return pickle.load(fh)


def average_improvement_lp(ptime: Sequence[float], crossover_ori: Sequence[float]) -> Tuple[float, float, float, int]:
    """Geometric means of the perturbation-crossover time, of the solver's own crossover time and of the
    better of the two per instance, plus the number of instances the perturbation crossover wins
    (visualization.py:181-195).  A missing time (NaN) counts as 3600 s in ``ptime`` and in the per-instance
    minimum; in ``crossover_ori`` it is skipped by the product but still counted in the root's denominator, which
    is what pandas' ``Series.prod()`` over ``len(df)`` does in the reference."""
    p = np.asarray(ptime, dtype=np.float64)
    c = np.asarray(crossover_ori, dtype=np.float64)
    par = np.where(np.isnan(p), c, np.where(np.isnan(c), p, np.minimum(p, c)))   # DataFrame.min(axis=1) skips NaN
    par = np.where(np.isnan(par), TIME_LIMIT_PENALTY, par)
    improved = int(np.count_nonzero(p < c))          # counted before the penalty is filled in
    p = np.where(np.isnan(p), TIME_LIMIT_PENALTY, p)
    n = p.size
    return (float(np.prod(p) ** (1.0 / n)), float(np.nanprod(c) ** (1.0 / n)), float(np.prod(par) ** (1.0 / n)), improved)


def grouped_geometric_mean(rows: Mapping[str, Mapping[str, float]]) -> Dict[str, Dict[str, float]]:
    """Shifted geometric mean of every column by instance group -- the part of the instance name before the
    first underscore -- rounded to two decimals (visualization.py:413-417, 426-429)."""
    groups: Dict[str, Dict[str, list]] = {}
    for name, cols in rows.items():
        g = groups.setdefault(name.split("_")[0], {})
        for key, val in cols.items():
            g.setdefault(key, []).append(float(val))
    return {g: {k: float(np.round(np.exp(np.mean(np.log(np.asarray(v) + GEOMEAN_SHIFT))), 2)) for k, v in cols.items()}
            for g, cols in groups.items()}
