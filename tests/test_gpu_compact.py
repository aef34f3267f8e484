"""GPU parity: K6 sub-problem compaction (sx_compact_columns / sx_fixed_rhs / sx_gather) against the
reference goldens and the oracle.  Bit-exact (index work and the two-sum right-hand side)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal, csr_from, same_csr
from oracle import lp_path as L
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def device_sub_problem(ctx, A, b, c, l, u, code):
    dA = ctx.matrix(A)
    dcode = ctx.to_device(code.astype(np.uint8))
    dsub, non_fix = ctx.compact_columns(dA, dcode)
    du, dl, db, dc = (ctx.to_device(v) for v in (u, l, b, c))
    b_sub = ctx.empty(A.shape[0], np.float64)
    ctx.fixed_rhs(dA, dcode, du, dl, db, b_sub)
    out = dict(non_fix=non_fix.download(), A=dsub.to_scipy(), b=b_sub.download(),
               c=ctx.gather(non_fix, dc).download(), l=ctx.gather(non_fix, dl).download(),
               u=ctx.gather(non_fix, du).download(), dsub=dsub)
    dA.free()
    return out


@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_sub_problem_matches_reference_golden(ctx, gname, request):
    g = request.getfixturevalue(gname)
    A = csr_from(g, "A")
    n = A.shape[1]
    code = np.zeros(n, np.uint8)
    code[g["fix_low"]] |= 1
    code[g["fix_up"]] |= 2
    got = device_sub_problem(ctx, A, g["b"], g["c_pt_opt"], g["l"], g["u"], code)
    assert np.array_equal(got["non_fix"], g["non_fix"])
    assert same_csr(got["A"], csr_from(g, "Asub"))
    assert bits_equal(got["b"], g["b_sub"])
    assert bits_equal(got["c"], g["c_sub"]) and bits_equal(got["l"], g["l_sub"]) and bits_equal(got["u"], g["u_sub"])
    got["dsub"].free()


def random_case(seed, m, n, k, frac_fix):
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=(m >= 2000), frac_upper=0.4)
    rng = np.random.default_rng(seed + 100)
    code = np.zeros(n, np.uint8)
    r = rng.random(n)
    code[r < frac_fix / 2] = 1
    has_up = np.isfinite(inst.u)
    code[(r > 1 - frac_fix / 2) & has_up] = 2
    code[rng.integers(0, n, 3)] = 3          # a few columns in both sets (possible for infeasible x)
    code[~has_up & (code >= 2)] = 1
    return inst, code


@pytest.mark.parametrize("seed,m,n,k,frac", [(1, 300, 2000, 6, 0.8), (2, 2000, 10000, 20, 0.5), (3, 50, 40, 3, 0.0),
                                             (4, 64, 500, 4, 1.0)])
def test_sub_problem_matches_oracle(ctx, seed, m, n, k, frac):
    inst, code = random_case(seed, m, n, k, frac)
    if frac == 1.0:
        code[:] = 1                            # every column fixed: empty sub-problem
    fix_low = np.flatnonzero(code & 1)
    fix_up = np.flatnonzero(code & 2)
    want = L.sub_problem(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, fix_low, fix_up, np.array([], dtype=np.int64))
    got = device_sub_problem(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, code)
    assert np.array_equal(got["non_fix"], want["non_fix"])
    assert same_csr(got["A"], want["A"])
    assert bits_equal(got["b"], want["b"])
    assert bits_equal(got["c"], want["c"]) and bits_equal(got["l"], want["l"]) and bits_equal(got["u"], want["u"])
    # the compacted matrix is a first-class resident matrix: score on it == oracle on the sub-LP
    if got["non_fix"].size:
        sub = got["dsub"]
        rng = np.random.default_rng(9)
        y = rng.standard_normal(m)
        xs = rng.random(got["non_fix"].size)
        s_d, cd = ctx.empty(xs.size, np.float64), ctx.empty(xs.size, np.uint8)
        s_p, fl = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
        ctx.score_columns(sub, ctx.to_device(y), ctx.to_device(want["c"]), ctx.to_device(xs), ctx.to_device(want["l"]),
                          ctx.to_device(want["u"]), 1e-3, s_d, cd)
        ctx.score_rows(sub, ctx.to_device(xs), ctx.to_device(want["b"]), ctx.to_device(y), 1e-3, s_p, fl)
        ref = L.scoring_pass(want["A"], want["b"], want["c"], want["l"], want["u"], xs, y)
        assert bits_equal(s_d.download(), ref["s_d"]) and np.array_equal(cd.download(), ref["code"])
        assert bits_equal(s_p.download(), ref["s_p"]) and np.array_equal(fl.download(), ref["rowflag"])
    got["dsub"].free()


def test_sub_problem_unsorted_duplicates_long_rows(ctx):
    from test_gpu_lp_parity import ragged_matrix
    A = ragged_matrix(7)
    m, n = A.shape
    rng = np.random.default_rng(3)
    code = (rng.random(n) < 0.6).astype(np.uint8) * rng.integers(1, 3, n).astype(np.uint8)
    code[333] = 0                              # keep the 9000-entry column
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    l, u = -rng.random(n), 1 + rng.random(n)
    want = L.sub_problem(A, b, c, l, u, np.full(m, "="), np.flatnonzero(code & 1), np.flatnonzero(code & 2),
                         np.array([], dtype=np.int64))
    got = device_sub_problem(ctx, A, b, c, l, u, code)
    assert np.array_equal(got["non_fix"], want["non_fix"])
    assert same_csr(got["A"], want["A"])
    assert bits_equal(got["b"], want["b"])
    got["dsub"].free()


def test_sub_problem_inf_bounds_are_not_multiplied(ctx):
    """u = inf on a column that is NOT fixed up must not leak a NaN (0*inf) into b_sub."""
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [0.0, 1.0, 3.0]]))
    b, c = np.array([1.0, 2.0]), np.zeros(3)
    l, u = np.array([0.0, -np.inf, 0.5]), np.array([np.inf, np.inf, 2.0])
    code = np.array([1, 0, 2], np.uint8)
    want = L.sub_problem(A, b, c, l, u, np.array(["=", "="]), np.array([0]), np.array([2]), np.array([], dtype=np.int64))
    got = device_sub_problem(ctx, A, b, c, l, u, code)
    assert bits_equal(got["b"], want["b"]) and np.all(np.isfinite(got["b"]))
    got["dsub"].free()
