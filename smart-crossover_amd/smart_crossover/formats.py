"""Problem containers with the field names and method names of the reference's
``smart_crossover/formats.py`` (GeneralLP :10-80, StandardLP :83-101, MinCostFlow :104-121,
OptTransport :124-161), so that existing callers keep working.

Difference in substance: the arithmetic methods on the crossover hot path --
``get_dual_slack`` (c - A^T y), ``get_primal_slack`` (b - A x) and ``get_standard_x`` -- run in
libsxhip.so on the MI355X against a matrix that stays resident in HBM (``hip.resident``); they raise
when the library or a GPU is missing, there is no numpy fallback.  The purely structural helpers
(index sets, standard-form assembly, OT -> MCF incidence) are host-side format conversions.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Union

import numpy as np
import scipy.sparse as sp

Matrix = Union[sp.csr_matrix, np.ndarray]


def _as_csr(A: Matrix) -> sp.csr_matrix:
    return A if sp.isspmatrix_csr(A) else sp.csr_matrix(A)


@dataclass
class GeneralLP:
    """min c^T x  s.t.  A x (sense) b,  l <= x <= u,  sense[i] in {'=', '<'}."""

    A: Matrix
    b: np.ndarray
    c: np.ndarray
    l: np.ndarray
    u: np.ndarray
    sense: np.ndarray
    name: str = "lp_instance"

    def __post_init__(self) -> None:
        sense = np.asarray(self.sense)
        if not np.all((sense == "=") | (sense == "<")):
            raise AssertionError("GeneralLP only allows '=' and '<' constraints")

    # ---- structure (host) ---------------------------------------------------------------------
    def get_free_ind(self) -> np.ndarray:
        return np.flatnonzero(np.isneginf(self.l) & np.isposinf(self.u))

    def get_nonfree_ind(self) -> np.ndarray:
        keep = np.ones(self.get_standard_c().size, dtype=bool)
        keep[self.get_free_ind()] = False
        return np.flatnonzero(keep)

    def get_free_var_matrix(self) -> sp.csr_matrix:
        return _as_csr(self.A)[:, self.get_free_ind()]

    def get_nonfree_var_matrix(self) -> sp.csr_matrix:
        return self.get_standard_A()[:, self.get_nonfree_ind()]

    def _slack_rows(self) -> np.ndarray:
        return np.flatnonzero(np.asarray(self.sense) == "<")

    def get_standard_A(self) -> sp.csr_matrix:
        """[A | unit columns of the '<' rows]: the equality form A_std x_std = b."""
        m = self.b.size
        rows = self._slack_rows()
        unit = sp.csr_matrix((np.ones(rows.size), (rows, np.arange(rows.size))), shape=(m, rows.size))
        return sp.hstack([_as_csr(self.A), unit], format="csr")

    def get_standard_c(self) -> np.ndarray:
        return np.concatenate([self.c, np.zeros(self._slack_rows().size)])

    # ---- arithmetic on the hot path (device) ----------------------------------------------------
    def _resident(self):
        from .hip.resident import resident_for
        return resident_for(self)

    def invalidate_device(self) -> None:
        """Forget the HBM copy of A (needed only after modifying A's arrays in place)."""
        res = getattr(self, "_sx_resident", None)
        if res is not None:
            res.free()
            self._sx_resident = None

    def get_dual_slack(self, y: np.ndarray) -> np.ndarray:
        """c - A^T y, rounded exactly like scipy's ``A.transpose() @ y`` (kernel K1)."""
        return self._resident().dual_slack(self.c, y)

    def get_primal_slack(self, x: np.ndarray) -> np.ndarray:
        """b - A x (kernel K2)."""
        return self._resident().primal_slack(self.b, x)

    def get_standard_x(self, x: np.ndarray) -> np.ndarray:
        """[x, b_< - A_< x]: a primal vector extended by the slacks of the '<' rows."""
        s_p = self.get_primal_slack(x)
        return np.concatenate([x, s_p[self._slack_rows()]])

    def copy(self) -> "GeneralLP":
        return GeneralLP(self.A.copy(), self.b.copy(), self.c.copy(), self.l.copy(), self.u.copy(),
                         np.asarray(self.sense).copy(), self.name)

    def _copy_sharing_matrix(self) -> "GeneralLP":
        """Copy of the vectors, same (immutable by convention) matrix object and HBM residency."""
        twin = GeneralLP(self.A, self.b.copy(), self.c.copy(), self.l.copy(), self.u.copy(),
                         np.asarray(self.sense).copy(), self.name)
        res = getattr(self, "_sx_resident", None)
        if res is not None:
            twin._sx_resident = res
        return twin


@dataclass
class StandardLP:
    """min c^T x  s.t.  A x = b,  l <= x <= u  with l in {0, -inf} (default 0)."""

    A: Matrix
    b: np.ndarray
    c: np.ndarray
    u: np.ndarray
    name: str = "lp_instance"
    l: Optional[np.ndarray] = None

    def __post_init__(self) -> None:
        if self.l is None:
            self.l = np.zeros_like(self.u)

    def to_general(self) -> GeneralLP:
        """The lift SURVEY.md section 0 asks for: every row an equality, bounds carried over."""
        return GeneralLP(self.A, self.b, self.c, self.l, self.u, np.full(self.b.size, "="), self.name)


@dataclass
class MinCostFlow(StandardLP):
    """min c^T x  s.t.  A x = b (node-arc incidence), 0 <= x <= u.  A is kept in CSR; sum(b) must
    vanish (atol 1e-8) or ValueError is raised, as in the reference."""

    name: str = "mcf_instance"

    def __post_init__(self) -> None:
        if self.l is None:
            self.l = np.zeros_like(self.u)
        self.A = self.A.tocsr() if sp.issparse(self.A) else sp.csr_matrix(self.A)
        if not np.isclose(np.sum(self.b), 0, atol=1e-8):
            raise ValueError("The sum of the b array must be equal to 0.")


@dataclass
class OptTransport:
    """Suppliers s[S], demanders d[D], cost M[S, D]; sum(s) must equal sum(d) (atol 1e-8)."""

    s: np.ndarray
    d: np.ndarray
    M: Union[sp.csr_matrix, np.ndarray]
    name: str = "ot_instance"

    def __post_init__(self) -> None:
        if not np.isclose(np.sum(self.s), np.sum(self.d), atol=1e-8):
            raise ValueError("The sum of the s and d arrays must be the same.")

    def incidence(self) -> sp.csr_matrix:
        """(S+D) x (S*D) node-arc matrix: arc (i, j) = column i*D + j leaves supplier row i with -1
        and enters demander row S + j with +1."""
        S, D = self.s.size, self.d.size
        # the CSR arrays written down directly (canonical: sorted columns, no duplicates): supplier row i holds the
        # columns i*D .. i*D + D - 1, demander row S + j the columns j, j + D, j + 2D, ...
        it = np.int32 if 2 * S * D < 2 ** 31 else np.int64   # one index type: scipy would convert otherwise
        indptr = np.concatenate([np.arange(S + 1, dtype=it) * D, S * D + np.arange(1, D + 1, dtype=it) * S])
        indices = np.concatenate([np.arange(S * D, dtype=it),
                                  (np.arange(D, dtype=it)[:, None] + np.arange(S, dtype=it)[None, :] * D).ravel()])
        vals = np.concatenate([-np.ones(S * D), np.ones(S * D)])
        A = sp.csr_matrix((vals, indices, indptr), shape=(S + D, S * D))
        A.has_sorted_indices = True
        A.has_canonical_format = True
        return A

    def to_MCF(self) -> MinCostFlow:
        n = self.s.size * self.d.size
        M = self.M.toarray() if sp.issparse(self.M) else np.asarray(self.M)
        return MinCostFlow(A=self.incidence(), b=np.concatenate([-self.s, self.d]), c=M.flatten(),
                           u=np.full(n, np.inf))
