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
    """Bit-for-bit equality of two float arrays (NaN payloads and signed zeros included)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


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
