"""CPU: libsxhip.so loads without a GPU and exports exactly the entry points that include/sxhip.h
declares; the ctypes prototype table mirrors the header one to one.  No compute calls."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "sxhip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int)\s*(sx_\w+)\s*\(", text, flags=re.M)
    assert len(names) == len(set(names)), "duplicate declaration in the header"
    return names


def test_header_declares_the_hot_path_entry_points():
    names = set(declared_functions())
    for must in ("sx_score_columns_dev", "sx_score_rows_dev", "sx_select_indices_dev", "sx_perturb_cost_dev",
                 "sx_price_dev", "sx_projector_dev", "sx_compact_columns_dev", "sx_fixed_rhs_dev",
                 "sx_flow_indicator_mcf_dev", "sx_flow_indicator_ot_dev", "sx_argsort_desc_dev", "sx_price_ot_dev",
                 "sx_matrix_create", "sx_ctx_create", "sx_last_error"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    from smart_crossover.hip import lib as sxl
    path = sxl.lib_path()
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    dll = ctypes.CDLL(path)
    missing = [n for n in declared_functions() if not hasattr(dll, n)]
    assert not missing, f"not exported: {missing}"


def test_ctypes_table_matches_header():
    from smart_crossover.hip import lib as sxl
    declared = set(declared_functions())
    bound = set(sxl.PROTOTYPES)
    assert bound == declared, f"only in header: {sorted(declared - bound)}; only in lib.py: {sorted(bound - declared)}"
    lib = sxl.load()
    assert lib.sx_abi_version() == 1
    # argument counts agree with the C declarations
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in sxl.PROTOTYPES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        count = 0 if params in ("", "void") else len(params.split(","))
        assert count == len(args), f"{name}: header has {count} parameters, lib.py binds {len(args)}"


def test_no_gpu_means_loud_failure_not_fallback():
    """On a box without a GPU the product path raises; it never computes on the CPU."""
    from smart_crossover.hip import lib as sxl
    if sxl.device_count() > 0:
        pytest.skip("a GPU is present")
    from smart_crossover.hip import Context
    from smart_crossover.hip.device import default_context
    with pytest.raises(sxl.SxError):
        Context(0)
    with pytest.raises(sxl.SxLibraryError):
        default_context()
    import numpy as np
    import scipy.sparse as sp
    from smart_crossover.formats import GeneralLP
    lp = GeneralLP(sp.identity(3, format="csr"), np.ones(3), np.ones(3), np.zeros(3), np.ones(3), np.full(3, "="))
    with pytest.raises((sxl.SxLibraryError, sxl.SxError)):
        lp.get_dual_slack(np.ones(3))
    from smart_crossover.lp_methods import algorithms as alg
    with pytest.raises((sxl.SxLibraryError, sxl.SxError)):
        alg.get_perturb_problem(lp, np.ones(3), np.ones(3), 1e-3, 1e-3, False)


def test_missing_library_is_reported(monkeypatch, tmp_path):
    from smart_crossover.hip import lib as sxl
    monkeypatch.setenv("SXHIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(sxl, "_lib", None)
    with pytest.raises(sxl.SxLibraryError):
        sxl.load()
