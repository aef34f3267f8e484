"""ctypes binding of libsxhip.so (the C ABI declared in include/sxhip.h).

There is no CPU fallback: if the shared library is missing or cannot be
loaded, importing this module succeeds (so that CPU-only tooling can inspect
the package) but the first call to :func:`load` raises ``SxLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

SX_OK = 0
SX_ERR_INVALID = -1
SX_ERR_HIP = -2
SX_ERR_NOMEM = -3
SX_ERR_UNSUPPORTED = -4

CODE_LOW = 1
CODE_UP = 2

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEFAULT_LIB = os.path.join(_PKG_ROOT, "lib", "libsxhip.so")


class SxLibraryError(RuntimeError):
    """libsxhip.so is missing or unusable -- the HIP path cannot run."""


class SxError(RuntimeError):
    """A libsxhip call failed with SX_ERR_HIP."""


class PriceResult(C.Structure):
    _fields_ = [("min_rc", C.c_double), ("argmin", C.c_int64), ("n_violating", C.c_int64)]


class CgResult(C.Structure):
    _fields_ = [("proj_norm", C.c_double), ("b_norm", C.c_double), ("rel_residual", C.c_double),
                ("iters", C.c_int64), ("converged", C.c_int64)]


class SimplexResult(C.Structure):
    _fields_ = [("status", C.c_int64), ("iters", C.c_int64), ("phase1_iters", C.c_int64),
                ("warm_start_used", C.c_int64), ("obj", C.c_double), ("max_violation", C.c_double)]


class PdlpResult(C.Structure):
    _fields_ = [("status", C.c_int64), ("iters", C.c_int64), ("restarts", C.c_int64), ("primal_residual", C.c_double),
                ("dual_residual", C.c_double), ("gap", C.c_double), ("primal_obj", C.c_double), ("dual_obj", C.c_double),
                ("b_norm", C.c_double), ("c_norm", C.c_double), ("step", C.c_double), ("primal_weight", C.c_double)]


class SinkhornResult(C.Structure):
    _fields_ = [("iters", C.c_int64), ("status", C.c_int64), ("err", C.c_double)]


_vp = C.c_void_p
_i64 = C.c_int64
_dbl = C.c_double
_int = C.c_int
_u8 = C.c_uint8
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/sxhip.h one to one
PROTOTYPES = {
    "sx_abi_version": (_int, []),
    "sx_last_error": (C.c_char_p, []),
    "sx_device_count": (_int, [C.POINTER(_int)]),
    "sx_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "sx_ctx_destroy": (_int, [_vp]),
    "sx_ctx_sync": (_int, [_vp]),
    "sx_ctx_set_option": (_int, [_vp, C.c_char_p, _i64]),
    "sx_ctx_device_info": (_int, [_vp, C.c_char_p, _sz, C.POINTER(_int), C.POINTER(C.c_uint64)]),
    "sx_malloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "sx_free": (_int, [_vp, _vp]),
    "sx_upload": (_int, [_vp, _vp, _vp, _sz]),
    "sx_download": (_int, [_vp, _vp, _vp, _sz]),
    "sx_memset": (_int, [_vp, _vp, _int, _sz]),
    "sx_timer_start": (_int, [_vp]),
    "sx_timer_stop": (_int, [_vp, C.POINTER(C.c_float)]),
    "sx_marker_record": (_int, [_vp, _int]),
    "sx_marker_elapsed": (_int, [_vp, _int, _int, C.POINTER(C.c_float)]),
    "sx_ctx_sync_device": (_int, [_vp]),
    "sx_ctx_prefetch_block": (_int, [_vp, _sz]),
    "sx_pool_trim": (_int, []),
    "sx_pool_stats": (_int, [C.POINTER(C.c_uint64)] * 4),
    "sx_matrix_create_single": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "sx_matrix_create": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "sx_matrix_destroy": (_int, [_vp]),
    "sx_matrix_dims": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sx_matrix_arrays": (_int, [_vp] + [C.POINTER(_vp)] * 6),
    "sx_matrix_download_csr": (_int, [_vp, _vp, _vp, _vp]),
    "sx_matrix_rowblock_info": (_int, [_vp, _vp, _vp]),
    "sx_matrix_slabs_info": (_int, [_vp, _vp, _int, _vp]),
    "sx_matrix_rowblock_download": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_score_columns_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "sx_score_columns": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "sx_score_rows_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "sx_score_rows": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "sx_select_indices_dev": (_int, [_vp, _i64, _vp, _u8, _vp, _vp]),
    "sx_select_indices": (_int, [_vp, _i64, _vp, _u8, _vp, _vp]),
    "sx_perturb_cost_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp]),
    "sx_perturb_cost": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp]),
    "sx_price_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp, _vp]),
    "sx_price": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _vp, C.POINTER(PriceResult)]),
    "sx_compact_columns_dev": (_int, [_vp, _vp, _vp, C.POINTER(_vp), _vp, C.POINTER(_i64)]),
    "sx_gather_columns_dev": (_int, [_vp, _vp, _vp, _i64, C.POINTER(_vp)]),
    "sx_fixed_rhs_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_gather_f64_dev": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "sx_flow_indicator_mcf_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_flow_indicator_ot_dev": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "sx_spanning_tree_ot_dev": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "sx_argsort_desc_dev": (_int, [_vp, _i64, _vp, _vp]),
    "sx_price_ot_dev": (_int, [_vp, _i64, _i64, _vp, _vp, _dbl, _vp, _vp]),
    "sx_projector_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp, C.POINTER(CgResult)]),
    "sx_projector_std_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _int, _vp, _vp, C.POINTER(CgResult)]),
    "sx_x_real_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "sx_mask_f64_dev": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "sx_projector_norm_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _int, C.POINTER(CgResult)]),
    "sx_projector_free_dev": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(CgResult)]),
    "sx_cg_shard_open": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, C.POINTER(_vp), C.POINTER(_vp)]),
    "sx_cg_shard_start": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_int)]),
    "sx_cg_shard_local": (_int, [_vp]),
    "sx_cg_shard_update": (_int, [_vp, _int]),
    "sx_cg_shard_poll": (_int, [_vp, C.POINTER(_int), C.POINTER(_i64)]),
    "sx_cg_shard_finish": (_int, [_vp, _vp, _vp, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(CgResult)]),
    "sx_cg_shard_close": (_int, [_vp]),
    "sx_mcf_xhat_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sx_mcf_node_flows_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_mcf_arc_indicator_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_simplex_solve_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                    C.POINTER(SimplexResult)]),
    "sx_simplex_crossover_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                        C.POINTER(SimplexResult)]),
    "sx_pdlp_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _vp, _vp, C.POINTER(PdlpResult)]),
    "sx_crossover_band_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                     C.POINTER(SimplexResult)]),
    "sx_crossover_band_basis_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                           C.POINTER(SimplexResult)]),
    "sx_crossover_band_probe_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sx_bandlu_create_dev": (_int, [_vp, _i64, _int, _int, _i64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "sx_bandlu_factor_dev": (_int, [_vp, _dbl, C.POINTER(_i64), _vp, _vp]),
    "sx_bandlu_factor_blocks_dev": (_int, [_vp, _dbl, _int, _i64, _i64, _i64, C.POINTER(_i64), _vp, _vp]),
    "sx_bandlu_solve_dev": (_int, [_vp, _int, _i64, _vp, _i64]),
    "sx_bandlu_solve_sparse_dev": (_int, [_vp, _i64, _vp, _i64, _dbl]),
    "sx_bandlu_destroy": (_int, [_vp]),
    "sx_denselu_create_dev": (_int, [_vp, _i64, C.POINTER(_vp)]),
    "sx_denselu_set_dev": (_int, [_vp, _vp, _i64]),
    "sx_denselu_factor_dev": (_int, [_vp, _dbl, C.POINTER(_i64), _vp, _vp]),
    "sx_denselu_solve_dev": (_int, [_vp, _int, _i64, _vp, _i64]),
    "sx_denselu_destroy": (_int, [_vp]),
    "sx_netsimplex_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp,
                                 C.POINTER(SimplexResult)]),
    "sx_netdual_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp,
                              C.POINTER(SimplexResult)]),
    "sx_sinkhorn_dev": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _dbl, _i64, _dbl, _vp, _vp, _vp,
                               C.POINTER(SinkhornResult)]),
    "sx_sinkhorn_batch_dev": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _dbl, _i64, _dbl, _vp, _vp, _vp,
                                     C.POINTER(SinkhornResult)]),
    "sx_simplex_session_create": (_int, [_vp, C.POINTER(_vp)]),
    "sx_simplex_session_destroy": (_int, [_vp]),
    "sx_simplex_solve_session_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp,
                                            _vp, _vp, _vp, C.POINTER(SimplexResult)]),
    "sx_projector_norm": (_int, [_vp, _vp, _vp, _vp, _vp, _dbl, _int, C.POINTER(CgResult)]),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return os.environ.get("SXHIP_LIB", DEFAULT_LIB)


def load() -> C.CDLL:
    """Load libsxhip.so once and attach prototypes.  Raises SxLibraryError when
    it is absent -- the product path never substitutes CPU code for it."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SxLibraryError(
            f"{path} not found: build it with `make -C smart-crossover_amd` (or __graft_entry__.build()). "
            "The HIP path has no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise SxLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise SxLibraryError(f"{path} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.sx_abi_version() != 1:
        raise SxLibraryError(f"{path}: ABI version {lib.sx_abi_version()} != 1")
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map a C return code onto the exception the reference's Python API would raise."""
    if rc == SX_OK:
        return
    msg = (load().sx_last_error() or b"").decode(errors="replace")
    if rc == SX_ERR_INVALID:
        raise ValueError(msg)
    if rc == SX_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == SX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SxError(msg)


def device_count() -> int:
    n = _int(0)
    check(load().sx_device_count(C.byref(n)))
    return n.value
