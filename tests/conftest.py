import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "smart-crossover_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def csr_from(g, prefix):
    shape = tuple(int(v) for v in g[prefix + "_shape"])
    return sp.csr_matrix((g[prefix + "_data"], g[prefix + "_indices"], g[prefix + "_indptr"]), shape=shape)


def same_csr(A, B):
    A = sp.csr_matrix(A)
    B = sp.csr_matrix(B)
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(A.data, B.data))


def bits_equal(a, b):
    """Bit-for-bit equality of two float arrays: signed zeros, infinities and every finite value
    must match exactly; NaNs must sit at the same positions, but their sign/payload bits are not
    compared (x86 and gfx950 generate different default NaNs for inf - inf)."""
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    b = np.ascontiguousarray(b, dtype=np.float64).copy()
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    a[na] = 0.0
    b[nb] = 0.0
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_lp_afiro.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_lp_small.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_mcf_small.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_ot_small.npz")
