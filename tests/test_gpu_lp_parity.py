"""GPU parity: LP-side kernels of libsxhip.so (through the C ABI) against the oracle and the
reference goldens.  Bit-exact for slacks, codes, flags and index sets."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal, csr_from
from oracle import lp_path as L
from oracle import net_path as N
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def run_scoring(ctx, A, b, c, l, u, x, y, gamma, gamma_dual):
    dA = ctx.matrix(A)
    m, n = A.shape
    d = {k: ctx.to_device(v, np.float64) for k, v in dict(b=b, c=c, l=l, u=u, x=x, y=y).items()}
    s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], gamma, s_d, code)
    ctx.score_rows(dA, d["x"], d["b"], d["y"], gamma_dual, s_p, flag)
    out = dict(s_d=s_d.download(), code=code.download(), s_p=s_p.download(), flag=flag.download(),
               fix_low=ctx.where(code, 1), fix_up=ctx.where(code, 2), fixed_rows=ctx.where(flag, 0xFF))
    dA.free()
    return out


def assert_scoring_equal(got, want):
    assert bits_equal(got["s_d"], want["s_d"])
    assert bits_equal(got["s_p"], want["s_p"])
    assert np.array_equal(got["code"], want["code"])
    assert np.array_equal(got["flag"], want["rowflag"])
    for k in ("fix_low", "fix_up", "fixed_rows"):
        assert got[k].dtype == np.int64
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_scoring_matches_reference_golden(ctx, gname, request):
    g = request.getfixturevalue(gname)
    A = csr_from(g, "A")
    gamma, gamma_dual = g["gamma"]
    got = run_scoring(ctx, A, g["b"], g["c"], g["l"], g["u"], g["x"], g["y"], gamma, gamma_dual)
    assert bits_equal(got["s_d"], g["s_d"])
    assert bits_equal(got["s_p"], g["s_p"])
    assert np.array_equal(got["fix_low"], g["fix_low"])
    assert np.array_equal(got["fix_up"], g["fix_up"])
    assert np.array_equal(got["fixed_rows"], g["fixed_rows"])


@pytest.mark.parametrize("m,n,k,seed", [(500, 3000, 7, 1), (2000, 10000, 20, 22), (257, 255, 3, 4), (64, 1, 5, 9)])
def test_scoring_matches_oracle_random(ctx, m, n, k, seed):
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=(m * n > 5_000_000) or m >= 2000, frac_upper=0.3)
    want = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    got = run_scoring(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y, L.GAMMA0, L.GAMMA0)
    assert_scoring_equal(got, want)
    assert want["fix_low"].size + want["fix_up"].size > 0 or n <= m     # n <= m: every column is "basic"


def test_scoring_config2_size(ctx):
    inst = workloads.config2()
    want = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    got = run_scoring(ctx, inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y, L.GAMMA0, L.GAMMA0)
    assert_scoring_equal(got, want)
    assert want["fix_low"].size > 50_000 and want["fixed_rows"].size > 1000


def ragged_matrix(seed=0):
    """Unsorted rows with duplicates, empty rows/columns, one column longer than two staging
    chunks (9000 entries) and one row longer than a chunk."""
    rng = np.random.default_rng(seed)
    m, n = 9500, 700
    rows = [rng.integers(0, m, 40_000), np.arange(9000), np.full(5000, 17), rng.integers(0, m, 300)]
    cols = [rng.integers(0, n, 40_000), np.full(9000, 333), rng.integers(0, n, 5000), rng.integers(0, n, 300)]
    r = np.concatenate(rows)
    c = np.concatenate(cols)
    keep = (c != 5) & (c != 699) & (r != 3) & (r != m - 1)        # empty columns 5, 699; empty rows 3, m-1
    r, c = r[keep], c[keep]
    order = np.argsort(r, kind="stable")
    r, c = r[order], c[order]
    indptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=m))]).astype(np.int64)
    v = rng.standard_normal(r.size)
    return sp.csr_matrix((v, c.astype(np.int32), indptr), shape=(m, n))     # not canonical on purpose


def test_scoring_ragged_long_segments_duplicates(ctx):
    A = ragged_matrix()
    assert not A.has_canonical_format
    m, n = A.shape
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    l = np.where(rng.random(n) < 0.2, -np.inf, 0.0)
    u = np.where(rng.random(n) < 0.5, np.inf, 2.0)
    want = L.scoring_pass(A, b, c, l, u, x, y, 0.5, 0.5)
    got = run_scoring(ctx, A, b, c, l, u, x, y, 0.5, 0.5)
    assert_scoring_equal(got, want)
    assert {0, 1, 2}.issubset(set(np.unique(want["code"]).tolist()))


def test_scoring_empty_matrix_and_special_values(ctx):
    # nnz == 0: slacks are c and b themselves
    A = sp.csr_matrix((5, 300))
    rng = np.random.default_rng(2)
    x, y, b, c = rng.standard_normal(300), rng.standard_normal(5), rng.standard_normal(5), rng.standard_normal(300)
    l, u = np.zeros(300), np.full(300, np.inf)
    got = run_scoring(ctx, A, b, c, l, u, x, y, 1e-3, 1e-3)
    assert_scoring_equal(got, L.scoring_pass(A, b, c, l, u, x, y))
    # NaN / inf propagate exactly like numpy
    A = sp.random(40, 90, density=0.2, random_state=3, format="csr")
    x, y = rng.standard_normal(90), rng.standard_normal(40)
    y[3] = np.nan
    y[7] = np.inf
    x[11] = -np.inf
    b, c = rng.standard_normal(40), rng.standard_normal(90)
    l, u = np.full(90, -np.inf), np.full(90, np.inf)
    with np.errstate(all="ignore"):
        want = L.scoring_pass(A, b, c, l, u, x, y)
    got = run_scoring(ctx, A, b, c, l, u, x, y, 1e-3, 1e-3)
    assert_scoring_equal(got, want)


def test_select_indices_edge_cases(ctx):
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 4095, 4096, 4097, 1_000_003):
        flags = (rng.random(n) < 0.3).astype(np.uint8) * rng.integers(1, 4, n).astype(np.uint8)
        d = ctx.to_device(flags)
        for mask in (1, 2, 0xFF):
            assert np.array_equal(ctx.where(d, mask), np.flatnonzero(flags & mask))
    d = ctx.to_device(np.zeros(5000, np.uint8))
    assert ctx.where(d).size == 0
    d = ctx.to_device(np.ones(5000, np.uint8))
    assert np.array_equal(ctx.where(d), np.arange(5000))


@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_perturb_cost_matches_golden_and_oracle(ctx, gname, request):
    g = request.getfixturevalue(gname)
    n = g["c"].size
    xi = L.xi_vector(n)
    d = {k: ctx.to_device(g[k]) for k in ("x", "l", "u", "c")}
    dxi = ctx.to_device(xi)
    out = ctx.empty(n, np.float64)
    ctx.perturb_cost(n, d["x"], d["l"], d["u"], d["c"], dxi, 0.0, True, out)
    assert bits_equal(out.download(), g["c_pt_feas"])
    sf = float(g["sf"])
    ctx.perturb_cost(n, d["x"], d["l"], d["u"], d["c"], dxi, sf, False, out)
    # same inputs (xi, sf) -> identical arithmetic, IEEE division included
    assert bits_equal(out.download(), L.perturb_cost(g["c"], g["x"], g["l"], g["u"], xi, sf, False))
    np.testing.assert_allclose(out.download(), g["c_pt_opt"], rtol=1e-12, atol=0)


def test_perturb_cost_free_columns_and_caps(ctx):
    l = np.array([0.0, -np.inf, 0.0, -np.inf, 1.0, 0.0])
    u = np.array([np.inf, np.inf, 4.0, 3.0, 2.0, np.inf])
    x = np.array([1e-9, -5.0, 3.5, 1.0, 1.5, 1e-30])
    c = np.arange(6, dtype=float)
    xi = L.xi_vector(6)
    out = ctx.empty(6, np.float64)
    for sf in (1e-3, 1e9):          # the second one hits the 1e6 cap
        ctx.perturb_cost(6, ctx.to_device(x), ctx.to_device(l), ctx.to_device(u), ctx.to_device(c), ctx.to_device(xi),
                         sf, False, out)
        assert bits_equal(out.download(), L.perturb_cost(c, x, l, u, xi, sf, False))


def test_price_matches_oracle(ctx, g3):
    A1 = csr_from(g3, "A1")
    dA = ctx.matrix(A1)
    n = A1.shape[1]
    vb = g3["vb0"].astype(np.int8)
    rc = ctx.empty(n, np.float64)
    res = ctx.price(dA, ctx.to_device(g3["y"]), ctx.to_device(g3["c1"]), ctx.to_device(vb), 1e-6, rc)
    got = rc.download()
    assert bits_equal(got, g3["rc"])                       # reference golden
    mn, am, bad = ctx.read_price(res)
    assert mn == got.min() and am == int(np.flatnonzero(got == got.min())[0])
    assert bad == int(np.count_nonzero(~(got >= -1e-6)))
    assert (bad == 0) == bool(np.all(g3["rc"] >= -N.RC_TOL))
    # no vbasis, no rc output, optimal-looking dual
    res = ctx.price(dA, ctx.to_device(np.zeros(A1.shape[0])), ctx.to_device(np.abs(g3["c1"])), None, 1e-6, None)
    mn, am, bad = ctx.read_price(res)
    assert bad == 0 and mn == np.abs(g3["c1"]).min()


def test_price_large_with_nan(ctx):
    inst = workloads.sparse_lp(3000, 200_000, 5, seed=8, stratified=True)
    dA = ctx.matrix(inst.A)
    rng = np.random.default_rng(0)
    vb = rng.integers(-2, 1, inst.c.size).astype(np.int8)
    c = inst.c.copy()
    c[12345] = np.nan
    rc = ctx.empty(c.size, np.float64)
    res = ctx.price(dA, ctx.to_device(inst.y), ctx.to_device(c), ctx.to_device(vb), 1e-6, rc)
    want = N.mcf_reduced_cost(inst.A, c, inst.y, vb.astype(int))
    assert bits_equal(rc.download(), want)
    mn, am, bad = ctx.read_price(res)
    assert mn == np.nanmin(want) and am == int(np.flatnonzero(want == np.nanmin(want))[0])
    assert bad == int(np.count_nonzero(~(want >= -1e-6)))


def test_host_pointer_entry_points(ctx):
    """The non-_dev ABI functions (numpy pointers in, numpy pointers out)."""
    from smart_crossover.hip import lib as sxl
    lib = sxl.load()
    inst = workloads.sparse_lp(300, 1000, 4, seed=3, stratified=False)
    want = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
    dA = ctx.matrix(inst.A)
    m, n = inst.A.shape
    p = lambda a: a.ctypes.data  # noqa: E731
    s_d, code = np.empty(n), np.empty(n, np.uint8)
    sxl.check(lib.sx_score_columns(ctx.handle, dA.handle, p(inst.y), p(inst.c), p(inst.x), p(inst.l), p(inst.u), 1e-3,
                                   p(s_d), p(code)))
    s_p, flag = np.empty(m), np.empty(m, np.uint8)
    sxl.check(lib.sx_score_rows(ctx.handle, dA.handle, p(inst.x), p(inst.b), p(inst.y), 1e-3, p(s_p), p(flag)))
    assert bits_equal(s_d, want["s_d"]) and np.array_equal(code, want["code"])
    assert bits_equal(s_p, want["s_p"]) and np.array_equal(flag, want["rowflag"])
    idx, cnt = np.empty(n, np.int64), np.zeros(1, np.int64)
    sxl.check(lib.sx_select_indices(ctx.handle, n, p(code), 1, p(idx), p(cnt)))
    assert np.array_equal(idx[:cnt[0]], want["fix_low"])
    xi = L.xi_vector(n)
    out = np.empty(n)
    sxl.check(lib.sx_perturb_cost(ctx.handle, n, p(inst.x), p(inst.l), p(inst.u), p(inst.c), p(xi), 0.37, 0, p(out)))
    assert bits_equal(out, L.perturb_cost(inst.c, inst.x, inst.l, inst.u, xi, 0.37, False))
    res = sxl.PriceResult()
    rc = np.empty(n)
    sxl.check(lib.sx_price(ctx.handle, dA.handle, p(inst.y), p(inst.c), None, 1e-6, p(rc), C.byref(res)))
    assert bits_equal(rc, inst.c - inst.A.T @ inst.y) and res.min_rc == rc.min()
    # error mapping: NULL matrix -> SX_ERR_INVALID -> ValueError
    with pytest.raises(ValueError):
        sxl.check(lib.sx_score_rows(ctx.handle, None, p(inst.x), p(inst.b), p(inst.y), 1e-3, p(s_p), p(flag)))


def test_matrix_index_arrays_are_validated(ctx):
    """Offsets and inner indices are checked (on the device copies) before any kernel uses them as
    addresses: the error names the first offending position."""
    from smart_crossover.hip import lib as sxl
    lib = sxl.load()
    A = ragged_matrix(5)
    m, n = A.shape
    rowptr, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data

    def create(rp, ci, single=False):
        h = C.c_void_p()
        if single:
            return lib.sx_matrix_create_single(ctx.handle, m, n, A.nnz, 0, rp.ctypes.data, ci.ctypes.data, val.ctypes.data,
                                               C.byref(h))
        return lib.sx_matrix_create(ctx.handle, m, n, A.nnz, rp.ctypes.data, ci.ctypes.data, val.ctypes.data,
                                    None, None, None, C.byref(h))

    for single in (False, True):
        bad_col = col.copy()
        bad_col[A.nnz // 2] = n                                   # one past the last column
        bad_col[A.nnz - 1] = -1
        with pytest.raises(ValueError, match=rf"\[{A.nnz // 2}\] = {n} out of range"):
            sxl.check(create(rowptr, bad_col, single))
        bad_ptr = rowptr.copy()
        k = int(np.flatnonzero(np.diff(rowptr) > 0)[3])            # a non-empty row in the middle
        bad_ptr[k + 1] = bad_ptr[k] - 1 if bad_ptr[k] > 0 else bad_ptr[k + 2] + 1
        with pytest.raises(ValueError, match="not non-decreasing"):
            sxl.check(create(bad_ptr, col, single))
        short = rowptr.copy()
        short[-1] -= 1
        with pytest.raises(ValueError, match="but nnz"):
            sxl.check(create(short, col, single))
    # and a valid one still goes through afterwards
    h = C.c_void_p()
    sxl.check(lib.sx_matrix_create(ctx.handle, m, n, A.nnz, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data,
                                   None, None, None, C.byref(h)))
    sxl.check(lib.sx_matrix_destroy(h))


def test_matrix_roundtrip_and_library_csc(ctx):
    """Library-side CSR->CSC (when the caller passes no CSC) equals the walk order."""
    from smart_crossover.hip import lib as sxl
    lib = sxl.load()
    A = ragged_matrix(3)
    m, n = A.shape
    rowptr, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data
    h = C.c_void_p()
    sxl.check(lib.sx_matrix_create(ctx.handle, m, n, A.nnz, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data,
                                   None, None, None, C.byref(h)))
    from smart_crossover.hip import DeviceMatrix
    dA = DeviceMatrix.from_handle(ctx, h)
    back = dA.to_scipy()
    assert np.array_equal(back.indptr, A.indptr) and np.array_equal(back.indices, A.indices) and bits_equal(back.data, A.data)
    rng = np.random.default_rng(0)
    y, c = rng.standard_normal(m), rng.standard_normal(n)
    s_d = ctx.empty(n, np.float64)
    ctx.score_columns(dA, ctx.to_device(y), ctx.to_device(c), None, None, None, 0.0, s_d, None)
    assert bits_equal(s_d.download(), L.dual_slack(A, c, y))
    dA.free()


@pytest.mark.parametrize("structure,window", [("staircase", 4096), ("staircase", 300), ("uniform", 4096)])
def test_window_option_never_changes_results(ctx, structure, window):
    """K1 and K10 behind the LDS operand window (off / 1 / 2 / 4 / 8 tiles per load / auto) against the
    oracle: the window only changes where an operand is read from, never a bit of the result.  The
    staircase shard has 1/4 of its entries outside any 4096-row window (fallback path); the uniform
    one has nearly all of them outside."""
    sh = workloads.lp_shard(0, 1, m=40_000, n_block=300_000, k=8, structure=structure, window=window)
    A = sh.col_block
    dC = ctx.column_shard(A)
    n = A.shape[1]
    d = {k: ctx.to_device(getattr(sh, k)) for k in ("y", "x", "c", "l", "u")}
    want_sd = sh.c - A.T @ sh.y                 # CSR matvec of the transposed view: stored order per column
    want_code = L.column_codes(sh.x, sh.l, sh.u, want_sd, L.GAMMA0)
    rng = np.random.default_rng(5)
    vb = rng.integers(-2, 1, n).astype(np.int8)
    want_rc = N.mcf_reduced_cost(A, sh.c, sh.y, vb.astype(int))
    dvb = ctx.to_device(vb)
    try:
        for opt in (0, 1, 2, 4, 8, -1):
            ctx.set_option("window", opt)
            s_d, code, rc = ctx.empty(n, np.float64), ctx.empty(n, np.uint8), ctx.empty(n, np.float64)
            ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], L.GAMMA0, s_d, code)
            res = ctx.price(dC, d["y"], d["c"], dvb, 1e-6, rc)
            assert bits_equal(s_d.download(), want_sd), opt
            assert np.array_equal(code.download(), want_code), opt
            assert bits_equal(rc.download(), want_rc), opt
            mn, am, bad = ctx.read_price(res)
            assert mn == want_rc.min() and am == int(np.flatnonzero(want_rc == want_rc.min())[0]), opt
            assert bad == int(np.count_nonzero(~(want_rc >= -1e-6))), opt
    finally:
        ctx.set_option("window", -1)
        dC.free()
