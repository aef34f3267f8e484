"""Thin object layer over the C ABI: device context, device vectors, resident
sparse matrices and one Python method per kernel entry point.

Everything here only marshals pointers and sizes; all arithmetic happens in
libsxhip.so.  Device memory comes from ``sx_malloc`` by default; a
``DeviceArray`` can also wrap foreign device memory (``torch.Tensor.data_ptr()``)
so that RCCL collectives issued through ``torch.distributed`` can operate on
the same buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp

from . import lib as _l


def _ptr(a) -> Optional[int]:
    """Raw address of a DeviceArray / numpy array / None."""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    raise TypeError(f"expected DeviceArray, ndarray or None, got {type(a)!r}")


class Context:
    """One HIP device + stream + scratch space (``sx_ctx``)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        from . import host_threads
        host_threads.fit_to_quota()   # (BLAS / OpenMP workers beyond the CPU quota get the launching thread throttled)
        self._lib = _l.load()
        h = C.c_void_p()
        _l.check(self._lib.sx_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._timer_depth = 0

    # -- lifetime ---------------------------------------------------------
    def prefetch_block(self, nbytes: int) -> None:
        """Start allocating the context's big block (eta file + tableau of the sparse crossover) on a helper thread."""
        _l.check(self._lib.sx_ctx_prefetch_block(self.handle, int(nbytes)))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.sx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        _l.check(self._lib.sx_ctx_sync(self.handle))

    def set_option(self, key: str, value: int) -> None:
        _l.check(self._lib.sx_ctx_set_option(self.handle, key.encode(), int(value)))

    def device_info(self) -> Tuple[str, int, int]:
        name = C.create_string_buffer(128)
        cus = C.c_int(0)
        hbm = C.c_uint64(0)
        _l.check(self._lib.sx_ctx_device_info(self.handle, name, 128, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value

    # -- memory -----------------------------------------------------------
    def empty(self, n: int, dtype) -> "DeviceArray":
        return DeviceArray(self, int(n), np.dtype(dtype))

    def zeros(self, n: int, dtype) -> "DeviceArray":
        a = self.empty(n, dtype)
        if a.nbytes:
            _l.check(self._lib.sx_memset(self.handle, a.ptr, 0, a.nbytes))
        return a

    def to_device(self, host: np.ndarray, dtype=None) -> "DeviceArray":
        host = np.ascontiguousarray(host, dtype=dtype)
        a = self.empty(host.size, host.dtype)
        a.upload(host)
        return a

    def wrap(self, ptr: int, n: int, dtype, owner=None) -> "DeviceArray":
        """Adopt foreign device memory (kept alive by ``owner``)."""
        return DeviceArray(self, int(n), np.dtype(dtype), ptr=int(ptr), owner=owner)

    # -- stopwatch --------------------------------------------------------
    def timer_start(self) -> None:
        _l.check(self._lib.sx_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        _l.check(self._lib.sx_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)

    def marker(self, ident: int) -> None:
        _l.check(self._lib.sx_marker_record(self.handle, int(ident)))

    def marker_elapsed(self, a: int, b: int) -> float:
        ms = C.c_float(0)
        _l.check(self._lib.sx_marker_elapsed(self.handle, int(a), int(b), C.byref(ms)))
        return float(ms.value)

    def sync_device(self) -> None:
        _l.check(self._lib.sx_ctx_sync_device(self.handle))

    # -- matrices ---------------------------------------------------------
    def matrix(self, A: sp.spmatrix, csc: Optional[sp.csc_matrix] = None) -> "DeviceMatrix":
        return DeviceMatrix(self, A, csc)

    def column_shard(self, csc: sp.csc_matrix) -> "DeviceMatrix":
        """CSC-only resident matrix (columns of one rank); entries must be in walk order."""
        m, n = csc.shape
        ptr = np.ascontiguousarray(csc.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(csc.indices, dtype=np.int32)
        val = np.ascontiguousarray(csc.data, dtype=np.float64)
        h = C.c_void_p()
        _l.check(self._lib.sx_matrix_create_single(self.handle, m, n, csc.nnz, 1, ptr.ctypes.data, idx.ctypes.data,
                                                   val.ctypes.data, C.byref(h)))
        return DeviceMatrix.from_handle(self, h)

    def row_shard(self, csr: sp.csr_matrix) -> "DeviceMatrix":
        """CSR-only resident matrix (rows of one rank)."""
        m, n = csr.shape
        ptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(csr.indices, dtype=np.int32)
        val = np.ascontiguousarray(csr.data, dtype=np.float64)
        h = C.c_void_p()
        _l.check(self._lib.sx_matrix_create_single(self.handle, m, n, csr.nnz, 0, ptr.ctypes.data, idx.ctypes.data,
                                                   val.ctypes.data, C.byref(h)))
        return DeviceMatrix.from_handle(self, h)

    # -- kernels (device pointers; asynchronous on the context's stream) ---
    def score_columns(self, A, y, c, x, l, u, gamma, s_d=None, code=None) -> None:
        _l.check(self._lib.sx_score_columns_dev(self.handle, A.handle, _ptr(y), _ptr(c), _ptr(x), _ptr(l), _ptr(u),
                                                float(gamma), _ptr(s_d), _ptr(code)))

    def score_rows(self, A, x, b, y, gamma_dual, s_p=None, flag=None) -> None:
        _l.check(self._lib.sx_score_rows_dev(self.handle, A.handle, _ptr(x), _ptr(b), _ptr(y), float(gamma_dual),
                                             _ptr(s_p), _ptr(flag)))

    def select_indices(self, flags: "DeviceArray", mask: int, idx_out: "DeviceArray", count_out: "DeviceArray") -> None:
        _l.check(self._lib.sx_select_indices_dev(self.handle, flags.size, flags.ptr, int(mask), idx_out.ptr,
                                                 count_out.ptr))

    def where(self, flags: "DeviceArray", mask: int = 0xFF) -> np.ndarray:
        """np.where(flags & mask)[0] computed on the device, returned as a host int64 array."""
        idx = self.empty(max(flags.size, 1), np.int64)
        cnt = self.empty(1, np.int64)
        self.select_indices(flags, mask, idx, cnt)
        k = int(cnt.download()[0])
        return idx.download(k)

    def perturb_cost(self, n, x, l, u, c, xi, scale_factor, is_feas, c_pt) -> None:
        _l.check(self._lib.sx_perturb_cost_dev(self.handle, int(n), _ptr(x), _ptr(l), _ptr(u), _ptr(c), _ptr(xi),
                                               float(scale_factor), int(bool(is_feas)), _ptr(c_pt)))

    def price(self, A, y, c, vbasis=None, tol=1e-6, rc=None, result: Optional["DeviceArray"] = None):
        """Enqueue pricing; returns the 24-byte device record (download with ``read_price``)."""
        if result is None:
            result = self.empty(C.sizeof(_l.PriceResult), np.uint8)
        _l.check(self._lib.sx_price_dev(self.handle, A.handle, _ptr(y), _ptr(c), _ptr(vbasis), float(tol), _ptr(rc),
                                        result.ptr))
        return result

    def compact_columns(self, A, code: "DeviceArray"):
        """K6 (blocking): returns (A_sub as DeviceMatrix, non_fix as DeviceArray[int64] of length n_sub)."""
        non_fix = self.empty(max(A.shape[1], 1), np.int64)
        h = C.c_void_p()
        nsub = C.c_int64(0)
        _l.check(self._lib.sx_compact_columns_dev(self.handle, A.handle, code.ptr, C.byref(h), non_fix.ptr,
                                                  C.byref(nsub)))
        non_fix.size = int(nsub.value)
        non_fix.nbytes = non_fix.size * 8
        return DeviceMatrix.from_handle(self, h), non_fix

    def gather_columns(self, A, idx: "DeviceArray") -> "DeviceMatrix":
        """A[:, idx] for distinct columns in any order (blocking)."""
        h = C.c_void_p()
        _l.check(self._lib.sx_gather_columns_dev(self.handle, A.handle, idx.ptr, idx.size, C.byref(h)))
        return DeviceMatrix.from_handle(self, h)

    def fixed_rhs(self, A, code, u, l, b, b_sub) -> None:
        _l.check(self._lib.sx_fixed_rhs_dev(self.handle, A.handle, _ptr(code), _ptr(u), _ptr(l), _ptr(b), _ptr(b_sub)))

    def gather(self, idx: "DeviceArray", src: "DeviceArray", dst: Optional["DeviceArray"] = None) -> "DeviceArray":
        if dst is None:
            dst = self.empty(idx.size, np.float64)
        _l.check(self._lib.sx_gather_f64_dev(self.handle, idx.size, idx.ptr, src.ptr, dst.ptr))
        return dst

    def flow_indicator_mcf(self, A, x, u, ind, xhat=None, f=None) -> None:
        _l.check(self._lib.sx_flow_indicator_mcf_dev(self.handle, A.handle, _ptr(x), _ptr(u), _ptr(ind), _ptr(xhat),
                                                     _ptr(f)))

    def flow_indicator_ot(self, S, D, X, s, d, ind) -> None:
        _l.check(self._lib.sx_flow_indicator_ot_dev(self.handle, int(S), int(D), _ptr(X), _ptr(s), _ptr(d), _ptr(ind)))

    def spanning_tree_ot(self, S, D, w: "DeviceArray") -> np.ndarray:
        """K13: arc indices (ascending) of the maximum-weight spanning forest of the S x D bipartite graph."""
        flags = self.empty(max(int(S) * int(D), 1), np.uint8)
        _l.check(self._lib.sx_spanning_tree_ot_dev(self.handle, int(S), int(D), _ptr(w), flags.ptr))
        return self.where(flags, 0xFF) if int(S) * int(D) else np.zeros(0, dtype=np.int64)

    def argsort_desc(self, key: "DeviceArray", out: Optional["DeviceArray"] = None) -> "DeviceArray":
        if out is None:
            out = self.empty(key.size, np.int64)
        _l.check(self._lib.sx_argsort_desc_dev(self.handle, key.size, key.ptr, out.ptr))
        return out

    def price_ot(self, S, D, M, y, tol=1e-6, rc=None, result: Optional["DeviceArray"] = None):
        if result is None:
            result = self.empty(C.sizeof(_l.PriceResult), np.uint8)
        _l.check(self._lib.sx_price_ot_dev(self.handle, int(S), int(D), _ptr(M), _ptr(y), float(tol), _ptr(rc),
                                           result.ptr))
        return result

    def projector_norm(self, A, xa, xs, c, tol=1e-8, maxiter=1000, proj_cols=None, proj_rows=None,
                       cs=None) -> "_l.CgResult":
        """K4 (blocking): ||(I - Y^T (YY^T)^+ Y) v|| by matrix-free CG; device pointers in; the
        projection itself lands in proj_cols[n] / proj_rows[m] when given.  ``cs`` (m) is an optional
        cost on the slack columns."""
        res = _l.CgResult()
        _l.check(self._lib.sx_projector_std_dev(self.handle, A.handle, _ptr(xa), _ptr(xs), _ptr(c), _ptr(cs),
                                                float(tol), int(maxiter), _ptr(proj_cols), _ptr(proj_rows),
                                                C.byref(res)))
        return res

    def projector_free(self, A, free_idx, xa, xs, c, row_lt, proj_cols, proj_rows) -> "_l.CgResult":
        """Free-variable branch of the projector (sx_projector_free_dev, blocking)."""
        res = _l.CgResult()
        _l.check(self._lib.sx_projector_free_dev(self.handle, A.handle, free_idx.size, free_idx.ptr, _ptr(xa), _ptr(xs),
                                                 _ptr(c), _ptr(row_lt), _ptr(proj_cols), _ptr(proj_rows), C.byref(res)))
        return res

    def simplex(self, A, b, c, l, u, row_is_lt, vbasis=None, cbasis=None, max_iter=0, feas_tol=1e-7, opt_tol=1e-7,
                x=None, y=None, vbasis_out=None, cbasis_out=None, session: Optional["SimplexSession"] = None,
                col_ids: Optional[np.ndarray] = None, x_start=None) -> "_l.SimplexResult":
        """K16 (blocking): bounded primal simplex on the device; device pointers in and out.  With a
        ``session`` and stable column identifiers ``col_ids`` (host int64) the basis inverse of the
        previous solve is reused when the warm basis is that solve's final basis.  With ``x_start`` (and a
        basis guess) the solve is a crossover from that interior point (sx_simplex_crossover_dev)."""
        res = _l.SimplexResult()
        if x_start is not None:
            _l.check(self._lib.sx_simplex_crossover_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u),
                                                        _ptr(row_is_lt), _ptr(vbasis), _ptr(cbasis), _ptr(x_start),
                                                        int(max_iter), float(feas_tol), float(opt_tol), _ptr(x), _ptr(y),
                                                        _ptr(vbasis_out), _ptr(cbasis_out), C.byref(res)))
            return res
        if session is None:
            _l.check(self._lib.sx_simplex_solve_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u),
                                                    _ptr(row_is_lt), _ptr(vbasis), _ptr(cbasis), int(max_iter),
                                                    float(feas_tol), float(opt_tol), _ptr(x), _ptr(y), _ptr(vbasis_out),
                                                    _ptr(cbasis_out), C.byref(res)))
            return res
        ids = None if col_ids is None else np.ascontiguousarray(col_ids, dtype=np.int64)
        if ids is not None and ids.size != A.shape[1]:
            raise ValueError("col_ids must name every structural column")
        _l.check(self._lib.sx_simplex_solve_session_dev(
            self.handle, session.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(row_is_lt), _ptr(vbasis),
            _ptr(cbasis), None if ids is None else ids.ctypes.data, int(max_iter), float(feas_tol), float(opt_tol),
            _ptr(x), _ptr(y), _ptr(vbasis_out), _ptr(cbasis_out), C.byref(res)))
        return res

    def pdlp(self, A, b, c, l, u, row_is_lt, x0=None, y0=None, max_iter=0, tol=1e-8, x=None, y=None) -> "_l.PdlpResult":
        """K16p (blocking): first-order stage of the LP re-solve (restarted PDHG) from (x0, y0); device pointers."""
        res = _l.PdlpResult()
        _l.check(self._lib.sx_pdlp_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(row_is_lt),
                                       _ptr(x0), _ptr(y0), int(max_iter), float(tol), _ptr(x), _ptr(y), C.byref(res)))
        return res

    def crossover_band(self, A, b, c, l, u, row_is_lt, x_start, max_iter=0, feas_tol=1e-9, opt_tol=1e-7, x=None, y=None,
                       vbasis_out=None, cbasis_out=None, vbasis_in=None, cbasis_in=None) -> "_l.SimplexResult":
        """K16s (blocking): sparse crossover from the first-order point -- or, with ``vbasis_in`` / ``cbasis_in``, from a given
        basis (the warm-started final solve) -- on a bordered band factorisation + a tableau of the tracked columns; raises
        NotImplementedError when the basis is no band matrix the band LU takes."""
        res = _l.SimplexResult()
        _l.check(self._lib.sx_crossover_band_basis_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(row_is_lt),
                                                       _ptr(x_start), _ptr(vbasis_in), _ptr(cbasis_in), int(max_iter), float(feas_tol),
                                                       float(opt_tol), _ptr(x), _ptr(y), _ptr(vbasis_out), _ptr(cbasis_out), C.byref(res)))
        return res

    def crossover_band_takes(self, A, b, c, l, u, row_is_lt, x_start) -> bool:
        """K16s' set-up up to its band-width check: would ``crossover_band`` take this LP from this point?"""
        rc = self._lib.sx_crossover_band_probe_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(row_is_lt), _ptr(x_start))
        if rc == _l.SX_ERR_UNSUPPORTED:
            return False
        _l.check(rc)
        return True

    def net_simplex(self, A, b, c, l, u, vbasis, cbasis, max_iter=0, feas_tol=1e-7, opt_tol=1e-7, x=None, y=None,
                    vbasis_out=None, cbasis_out=None) -> "_l.SimplexResult":
        """K16n (blocking): primal network simplex from a spanning-tree basis; status 5 = the problem or the
        basis is outside its domain (the caller then takes ``simplex``)."""
        res = _l.SimplexResult()
        _l.check(self._lib.sx_netsimplex_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(vbasis),
                                             _ptr(cbasis), int(max_iter), float(feas_tol), float(opt_tol), _ptr(x),
                                             _ptr(y), _ptr(vbasis_out), _ptr(cbasis_out), C.byref(res)))
        return res

    def net_dual(self, A, b, c, l, u, vbasis, cbasis, max_iter=0, feas_tol=1e-9, x=None, y=None, vbasis_out=None,
                 cbasis_out=None) -> "_l.SimplexResult":
        """K16d (blocking): dual network simplex from a spanning-tree basis that need not be primal feasible;
        status 5 = outside its domain (the caller then takes ``net_simplex`` / ``simplex``)."""
        res = _l.SimplexResult()
        _l.check(self._lib.sx_netdual_dev(self.handle, A.handle, _ptr(b), _ptr(c), _ptr(l), _ptr(u), _ptr(vbasis),
                                          _ptr(cbasis), int(max_iter), float(feas_tol), _ptr(x), _ptr(y),
                                          _ptr(vbasis_out), _ptr(cbasis_out), C.byref(res)))
        return res

    def simplex_session(self) -> "SimplexSession":
        return SimplexSession(self)

    def sinkhorn(self, S, D, a, b, M, reg, max_iter=1000, stop_thr=1e-9, plan=None, u=None, v=None) -> "_l.SinkhornResult":
        """Entropic OT warm start (blocking); device pointers in and out."""
        res = _l.SinkhornResult()
        _l.check(self._lib.sx_sinkhorn_dev(self.handle, int(S), int(D), _ptr(a), _ptr(b), _ptr(M), float(reg),
                                           int(max_iter), float(stop_thr), _ptr(plan), _ptr(u), _ptr(v), C.byref(res)))
        return res

    def sinkhorn_batch(self, S, D, B, a, b, M, reg, max_iter=1000, stop_thr=1e-9, plans=None, u=None, v=None):
        """B <= 16 entropic OT warm starts over one cost matrix (blocking); returns the B result records."""
        res = (_l.SinkhornResult * int(B))()
        _l.check(self._lib.sx_sinkhorn_batch_dev(self.handle, int(S), int(D), int(B), _ptr(a), _ptr(b), _ptr(M), float(reg),
                                                 int(max_iter), float(stop_thr), _ptr(plans), _ptr(u), _ptr(v), res))
        return list(res)

    def x_real(self, n, x, l, u, out, apply_floor: bool = True) -> None:
        _l.check(self._lib.sx_x_real_dev(self.handle, int(n), _ptr(x), _ptr(l), _ptr(u), int(bool(apply_floor)),
                                         _ptr(out)))

    def mask(self, src: "DeviceArray", mask: "DeviceArray", dst: "DeviceArray") -> None:
        _l.check(self._lib.sx_mask_f64_dev(self.handle, src.size, src.ptr, mask.ptr, dst.ptr))

    @staticmethod
    def read_price(result: "DeviceArray") -> Tuple[float, int, int]:
        raw = result.download()
        rec = _l.PriceResult.from_buffer_copy(raw.tobytes())
        return float(rec.min_rc), int(rec.argmin), int(rec.n_violating)


class BandLU:
    """K16f: band LU with partial pivoting of an n x n matrix given as device triplets (sx_bandlu_*)."""

    def __init__(self, ctx: "Context", n: int, kl: int, ku: int, row: "DeviceArray", col: "DeviceArray", val: "DeviceArray"):
        self.ctx, self.n, self.kl, self.ku = ctx, int(n), int(kl), int(ku)
        h = C.c_void_p()
        _l.check(ctx._lib.sx_bandlu_create_dev(ctx.handle, int(n), int(kl), int(ku), int(val.size), row.ptr, col.ptr, val.ptr,
                                               C.byref(h)))
        self.handle = h

    def factor(self, pivot_tol: float = 1e-11):
        """-> (replaced[n] int32: column became a unit vector, ipiv[n] int32: row swapped with j at step j)."""
        cnt = C.c_int64(0)
        rep, piv = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        _l.check(self.ctx._lib.sx_bandlu_factor_dev(self.handle, float(pivot_tol), C.byref(cnt), rep.ctypes.data, piv.ctypes.data))
        return rep, piv

    def factor_blocks(self, nblocks: int, stride: int, real_len: int, real_len_last: int, pivot_tol: float = 1e-11):
        """factor() for a block-diagonal matrix with identity padding between its blocks: the blocks side by side."""
        cnt = C.c_int64(0)
        rep, piv = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        _l.check(self.ctx._lib.sx_bandlu_factor_blocks_dev(self.handle, float(pivot_tol), int(nblocks), int(stride), int(real_len),
                                                           int(real_len_last), C.byref(cnt), rep.ctypes.data, piv.ctypes.data))
        return rep, piv

    def solve(self, X: "DeviceArray", nrhs: int = 1, ldx: Optional[int] = None, trans: bool = False) -> None:
        _l.check(self.ctx._lib.sx_bandlu_solve_dev(self.handle, int(bool(trans)), int(nrhs), X.ptr, int(ldx or self.n)))

    def solve_sparse(self, X: "DeviceArray", nrhs: int = 1, ldx: Optional[int] = None, tiny: float = 0.0) -> None:
        """A x = b for right-hand sides with few entries: panels holding nothing above `tiny` are skipped."""
        _l.check(self.ctx._lib.sx_bandlu_solve_sparse_dev(self.handle, int(nrhs), X.ptr, int(ldx or self.n), float(tiny)))

    def free(self) -> None:
        if self.handle is not None:
            self.ctx._lib.sx_bandlu_destroy(self.handle)
            self.handle = None


class DenseLU:
    """K16g: dense LU with partial pivoting of an n x n column-major device matrix (sx_denselu_*)."""

    def __init__(self, ctx: "Context", n: int, a: "DeviceArray", lda: Optional[int] = None):
        self.ctx, self.n = ctx, int(n)
        h = C.c_void_p()
        _l.check(ctx._lib.sx_denselu_create_dev(ctx.handle, int(n), C.byref(h)))
        self.handle = h
        _l.check(ctx._lib.sx_denselu_set_dev(h, a.ptr, int(lda or n)))

    def factor(self, pivot_tol: float = 1e-11):
        """-> (replaced[n] int32, rowperm[n] int32: original row at position i after the swaps)."""
        cnt = C.c_int64(0)
        rep, perm = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        _l.check(self.ctx._lib.sx_denselu_factor_dev(self.handle, float(pivot_tol), C.byref(cnt), rep.ctypes.data, perm.ctypes.data))
        return rep, perm

    def solve(self, X: "DeviceArray", nrhs: int = 1, ldx: Optional[int] = None, trans: bool = False) -> None:
        _l.check(self.ctx._lib.sx_denselu_solve_dev(self.handle, int(bool(trans)), int(nrhs), X.ptr, int(ldx or self.n)))

    def free(self) -> None:
        if self.handle is not None:
            self.ctx._lib.sx_denselu_destroy(self.handle)
            self.handle = None


class DeviceArray:
    """A typed, contiguous vector in HBM."""

    def __init__(self, ctx: Context, n: int, dtype: np.dtype, ptr: Optional[int] = None, owner=None):
        self.ctx = ctx
        self.size = n
        self.dtype = dtype
        self.nbytes = n * dtype.itemsize
        self._owned = ptr is None
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            _l.check(ctx._lib.sx_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)))
            self.ptr = p.value
        else:
            self.ptr = ptr

    def upload(self, host: np.ndarray) -> "DeviceArray":
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError(f"size mismatch: host {host.size} vs device {self.size}")
        if self.nbytes:
            _l.check(self.ctx._lib.sx_upload(self.ctx.handle, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def download(self, count: Optional[int] = None) -> np.ndarray:
        n = self.size if count is None else int(count)
        out = np.empty(n, dtype=self.dtype)
        if n:
            _l.check(self.ctx._lib.sx_download(self.ctx.handle, out.ctypes.data, self.ptr, n * self.dtype.itemsize))
        return out

    def free(self) -> None:
        if self._owned and self.ptr and getattr(self.ctx, "handle", None):
            self.ctx._lib.sx_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class DeviceMatrix:
    """A sparse matrix resident in HBM in both CSR and CSC (``sx_matrix``).

    The CSC copy is derived from the CSR arrays by a stable transposition so
    that per-column entries appear in row-major walk order (the accumulation
    order of the reference's ``A.transpose() @ y``)."""

    def __init__(self, ctx: Context, A: sp.spmatrix, csc: Optional[sp.csc_matrix] = None):
        self.ctx = ctx
        A = sp.csr_matrix(A)
        if A.dtype != np.float64:
            A = A.astype(np.float64)
        m, n = A.shape
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        col = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=np.float64)
        h = C.c_void_p()
        if csc is None:
            # the library derives the CSC layout on the device: the stable transposition, i.e. what
            # scipy's csr_tocsc would give, without seconds of host work on a large matrix
            _l.check(ctx._lib.sx_matrix_create(ctx.handle, m, n, A.nnz, rowptr.ctypes.data, col.ctypes.data,
                                               val.ctypes.data, None, None, None, C.byref(h)))
        else:
            colptr = np.ascontiguousarray(csc.indptr, dtype=np.int64)
            row = np.ascontiguousarray(csc.indices, dtype=np.int32)
            cval = np.ascontiguousarray(csc.data, dtype=np.float64)
            _l.check(ctx._lib.sx_matrix_create(ctx.handle, m, n, A.nnz, rowptr.ctypes.data, col.ctypes.data,
                                               val.ctypes.data, colptr.ctypes.data, row.ctypes.data, cval.ctypes.data,
                                               C.byref(h)))
        self.handle = h
        self.shape = (m, n)
        self.nnz = int(A.nnz)

    @classmethod
    def from_handle(cls, ctx: Context, handle) -> "DeviceMatrix":
        self = cls.__new__(cls)
        self.ctx = ctx
        self.handle = handle
        m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        _l.check(ctx._lib.sx_matrix_dims(handle, C.byref(m), C.byref(n), C.byref(nnz)))
        self.shape = (m.value, n.value)
        self.nnz = nnz.value
        return self

    def to_scipy(self) -> sp.csr_matrix:
        m, n = self.shape
        rowptr = np.empty(m + 1, dtype=np.int64)
        col = np.empty(self.nnz, dtype=np.int32)
        val = np.empty(self.nnz, dtype=np.float64)
        _l.check(self.ctx._lib.sx_matrix_download_csr(self.handle, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data))
        return sp.csr_matrix((val, col, rowptr), shape=(m, n))

    def slabs(self, which: int):
        """Operand slabs (csrc/sx_slabs.h) of the row walk (which = 0) or the column walk (which = 1) under the
        context's "slabs" option, built now if due: dict(R, width, nseg), or None for the plain walk."""
        info = (C.c_int64 * 3)()
        _l.check(self.ctx._lib.sx_matrix_slabs_info(self.ctx.handle, self.handle, int(which), info))
        return None if info[0] == 0 else dict(R=int(info[0]), width=int(info[1]), nseg=int(info[2]))

    def rowblock(self, download: bool = False):
        """Column-blocked row layout of the matrix (csrc/sx_rowblock.h) under the context's "rowblock"
        option, built now if due: a dict with the counts and, with ``download``, the arrays; None when the
        matrix uses the plain row walk."""
        info = (C.c_int64 * 6)()
        _l.check(self.ctx._lib.sx_matrix_rowblock_info(self.ctx.handle, self.handle, info))
        nst, ncells, nchunks, nent, windowed, stride = (int(v) for v in info)
        if nst == 0:
            return None
        out = dict(nst=nst, ncells=ncells, nchunks=nchunks, nent=nent, windowed=windowed, rs_stride=stride)
        if download:
            st = np.zeros(nst, dtype=[("row0", "<i8"), ("chunk0", "<i8"), ("nrows", "<i4"), ("nchunks", "<i4")])
            ch = np.zeros(nchunks, dtype=[("e0", "<i8"), ("ne", "<i4"), ("col0", "<i4"), ("cell", "<i4"), ("base", "<i4"),
                                          ("fresh", "<i4"), ("pad", "<i4")])
            rs = np.zeros(ncells * stride, dtype=np.uint16)
            idx = np.zeros(nent + 8, dtype=np.int32)
            val = np.zeros(nent + 8, dtype=np.float64)
            _l.check(self.ctx._lib.sx_matrix_rowblock_download(self.ctx.handle, self.handle, st.ctypes.data, ch.ctypes.data,
                                                               rs.ctypes.data, idx.ctypes.data, val.ctypes.data))
            out.update(st=st, chunks=ch, rowstart=rs.reshape(ncells, stride), idx=idx, val=val)
        return out

    def free(self) -> None:
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.sx_matrix_destroy(self.handle)
        self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class SimplexSession:
    """Keeps the device simplex's basis inverse between the solves of one column-generation sequence
    (sx_simplex_session_*)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = C.c_void_p()
        _l.check(ctx._lib.sx_simplex_session_create(ctx.handle, C.byref(h)))
        self.handle = h

    def free(self) -> None:
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx._lib.sx_simplex_session_destroy(self.handle)
        self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def pool_trim() -> None:
    """Hand the device memory the library keeps for its next allocations back to the driver (``sx_pool_trim``)."""
    _l.check(_l.load().sx_pool_trim())


def pool_stats() -> dict:
    """Bytes kept / in use by the library's device memory pool and requests served by it / by the driver."""

    v = [C.c_uint64(0) for _ in range(4)]
    _l.check(_l.load().sx_pool_stats(*[C.byref(x) for x in v]))
    return dict(zip(("cached_bytes", "live_bytes", "hits", "misses"), (int(x.value) for x in v)))


def default_context() -> Context:
    """Process-wide context on device LOCAL_RANK (or 0).  Raises when the
    library or a GPU is missing -- callers never get a CPU substitute."""
    global _default_ctx
    if _default_ctx is None:
        import os
        if _l.device_count() == 0:
            raise _l.SxLibraryError("no HIP device visible: the smart_crossover HIP path needs an MI355X")
        # one process per GPU: the rank's own device (SX_DEVICE overrides it: several ranks rehearsing on one GPU)
        _default_ctx = Context(int(os.environ.get("SX_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx
