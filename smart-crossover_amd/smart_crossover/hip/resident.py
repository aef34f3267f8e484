"""Device residency of one LP: the matrix (CSR + CSC with tile tables) stays in HBM across calls,
vectors are uploaded per call.  Used by ``formats.GeneralLP`` and ``lp_methods``; everything that
computes runs in libsxhip.so."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from .device import Context, DeviceArray, DeviceMatrix, default_context


def matrix_fingerprint(A: sp.csr_matrix) -> Tuple:
    """Cheap identity of a scipy CSR matrix: addresses and sizes of its three arrays.  A matrix whose
    arrays were modified *in place* keeps its fingerprint -- call ``invalidate_device()`` then."""
    return (A.shape, A.nnz, A.data.ctypes.data, A.indices.ctypes.data, A.indptr.ctypes.data)


class ResidentLP:
    """A (m x n) resident on one device plus the row-sense mask."""

    def __init__(self, A: sp.spmatrix, sense: np.ndarray, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        A = sp.csr_matrix(A)
        self.fingerprint = matrix_fingerprint(A)
        self.m, self.n = A.shape
        self.A: DeviceMatrix = self.ctx.matrix(A)
        self.set_sense(sense)

    @classmethod
    def adopt(cls, ctx: Context, device_matrix: DeviceMatrix, host_csr: sp.csr_matrix, sense: np.ndarray) -> "ResidentLP":
        """Residency for a matrix that already lives on the device (e.g. a compacted sub-matrix) and
        whose host copy is ``host_csr``."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.A = device_matrix
        self.m, self.n = device_matrix.shape
        self.fingerprint = matrix_fingerprint(host_csr)
        self.set_sense(sense)
        return self

    def set_sense(self, sense: np.ndarray) -> None:
        lt = (np.asarray(sense) == "<")
        self.n_lt = int(np.count_nonzero(lt))
        self.lt_host = lt
        self.lt = self.ctx.to_device(lt.astype(np.uint8))

    # -- helpers -----------------------------------------------------------------------------
    def put(self, v: np.ndarray) -> DeviceArray:
        return self.ctx.to_device(np.ascontiguousarray(v, dtype=np.float64))

    # -- K1 / K2 as plain vector functions -----------------------------------------------------
    def dual_slack(self, c: np.ndarray, y: np.ndarray) -> np.ndarray:
        out = self.ctx.empty(self.n, np.float64)
        self.ctx.score_columns(self.A, self.put(y), self.put(c), None, None, None, 0.0, out, None)
        return out.download()

    def primal_slack(self, b: np.ndarray, x: np.ndarray) -> np.ndarray:
        out = self.ctx.empty(self.m, np.float64)
        self.ctx.score_rows(self.A, self.put(x), self.put(b), None, 0.0, out, None)
        return out.download()

    def free(self) -> None:
        self.A.free()


def resident_for(lp, ctx: Optional[Context] = None) -> ResidentLP:
    """Residency cached on the LP object (attribute ``_sx_resident``), rebuilt when A was rebound."""
    A = lp.A if sp.isspmatrix_csr(lp.A) else sp.csr_matrix(lp.A)
    if A is not lp.A:
        lp.A = A
    res: Optional[ResidentLP] = getattr(lp, "_sx_resident", None)
    if res is None or res.fingerprint != matrix_fingerprint(A) or getattr(res.A, "handle", None) is None:
        if res is not None:
            res.free()
        res = ResidentLP(A, lp.sense, ctx)
        lp._sx_resident = res
    elif not np.array_equal(res.lt_host, np.asarray(lp.sense) == "<"):
        res.set_sense(lp.sense)
    return res
