"""GPU: operand slabs of the walks without locality (csrc/sx_slabs.h).  K1 (reference formats.py:70-72 +
lp_methods/algorithms.py:104-105), K2 (formats.py:74-76 + algorithms.py:106) and K10 (network_methods/net_manager.py:
302-303, 318) walked slab after slab must return the plain walk's bits -- the sums are scipy's sequential, separately
rounded sums whatever the layout -- and the CPU oracle's."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import bits_equal
from oracle import lp_path as L
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.set_option("slabs", -1)
    c.close()


def walks(ctx, A, x, y, b, c, l, u, slabs):
    """K1, K2, K10 outputs of one matrix under the given 'slabs' option (layouts that would take over are off)."""
    m, n = A.shape
    ctx.set_option("slabs", slabs)
    ctx.set_option("rowblock", 0)
    ctx.set_option("window", 0)
    dA = ctx.matrix(sp.csr_matrix(A))
    d = {k: ctx.to_device(v) for k, v in dict(x=x, y=y, b=b, c=c, l=l, u=u).items()}
    s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    rc = ctx.empty(n, np.float64)
    vb = ctx.to_device(np.where(np.arange(n) % 5 == 0, -2, -1).astype(np.int8))
    ctx.score_columns(dA, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
    ctx.score_rows(dA, d["x"], d["b"], d["y"], 1e-3, s_p, flag)
    pres = ctx.price(dA, d["y"], d["c"], vb, 1e-6, rc)
    out = (s_d.download(), code.download(), s_p.download(), flag.download(), rc.download(), pres.download())
    dA.free()
    ctx.set_option("rowblock", -1)
    ctx.set_option("window", -1)
    ctx.set_option("slabs", -1)
    return out


def same(a, b):
    return all(bits_equal(p, q) if p.dtype == np.float64 else np.array_equal(p, q) for p, q in zip(a, b))


def random_lp(m, n, k, seed, empty_cols=0, long_rows=0):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, m, size=n * k)
    cols = np.repeat(np.arange(n), k)
    if empty_cols:
        keep = ~np.isin(cols, rng.choice(n, empty_cols, replace=False))
        rows, cols = rows[keep], cols[keep]
    if long_rows:
        extra_c = rng.choice(n, size=long_rows * (n // 3), replace=True)
        extra_r = np.repeat(rng.choice(m, long_rows, replace=False), n // 3)
        rows, cols = np.r_[rows, extra_r], np.r_[cols, extra_c]
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))   # duplicates summed, canonical
    x, y = rng.random(n), rng.standard_normal(m)
    b, c = rng.standard_normal(m), rng.standard_normal(n)
    l, u = np.zeros(n), np.where(rng.random(n) < 0.3, 2.0, np.inf)
    return A, x, y, b, c, l, u


@pytest.mark.parametrize("slabs", [2, 3, 7, 64])
def test_slabbed_walks_return_the_plain_walks_bits(ctx, slabs):
    args = random_lp(3_000, 20_000, 6, seed=slabs, empty_cols=500, long_rows=3)
    plain = walks(ctx, *args, slabs=0)
    slabbed = walks(ctx, *args, slabs=slabs)
    assert same(plain, slabbed)


def test_slabbed_walks_against_the_cpu_oracle(ctx):
    A, x, y, b, c, l, u = random_lp(2_000, 9_000, 5, seed=11, empty_cols=100)
    got = walks(ctx, A, x, y, b, c, l, u, slabs=5)
    res = L.scoring_pass(A, b, c, l, u, x, y)
    assert bits_equal(got[0], res["s_d"]) and bits_equal(got[2], res["s_p"])


def test_segments_in_descending_order_keep_the_plain_walk(ctx):
    """Stored order is the order of the adds: a matrix whose segments are not index-ascending is refused by the
    builder, so the option changes nothing."""
    rng = np.random.default_rng(3)
    m, n, k = 500, 4_000, 4
    indptr = np.arange(0, n * k + 1, k, dtype=np.int64)
    rows = np.sort(rng.integers(0, m, size=(n, k)), axis=1)[:, ::-1].ravel().astype(np.int32)   # descending
    csc = sp.csc_matrix((rng.standard_normal(n * k), rows, indptr), shape=(m, n))
    csc.has_sorted_indices = True     # keep scipy from sorting
    y, c = rng.standard_normal(m), rng.standard_normal(n)
    outs = []
    for slabs in (0, 4):
        ctx.set_option("slabs", slabs)
        ctx.set_option("window", 0)
        dC = ctx.column_shard(csc)
        s_d = ctx.empty(n, np.float64)
        ctx.score_columns(dC, ctx.to_device(y), ctx.to_device(c), None, None, None, 1e-3, s_d, None)
        outs.append(s_d.download())
        dC.free()
    ctx.set_option("slabs", -1)
    ctx.set_option("window", -1)
    assert bits_equal(outs[0], outs[1])


def test_automatic_rule_at_config5_rows_uniform(ctx):
    """1e6 rows, uniformly random: y (8 MB) does not fit an XCD's L2 -> the column walk is slabbed by itself and
    returns the plain walk's bits."""
    sh = workloads.lp_shard(0, 1, m=1_000_000, n_block=1_000_000, k=8, structure="uniform")
    n = sh.col_block.shape[1]
    d = {k: ctx.to_device(getattr(sh, k)) for k in ("y", "c", "x", "l", "u")}
    outs = []
    for slabs in (0, -1):
        ctx.set_option("slabs", slabs)
        dC = ctx.column_shard(sh.col_block)
        s_d, code = ctx.empty(n, np.float64), ctx.empty(n, np.uint8)
        ctx.score_columns(dC, d["y"], d["c"], d["x"], d["l"], d["u"], 1e-3, s_d, code)
        outs.append((s_d.download(), code.download()))
        dC.free()
    ctx.set_option("slabs", -1)
    assert bits_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("head", [True, False])
def test_row_walk_with_linking_rows_at_the_head_or_spread(ctx, head):
    """The XCD-contiguous tile map of the plain row walk is dropped when the long rows sit together (sx_build_tiles'
    imbalance figure: they would all queue on one XCD) -- a scheduling decision that must not show in the sums."""
    rng = np.random.default_rng(7)
    m, n, k = 40_000, 60_000, 6
    rows = rng.integers(0, m, size=n * k)
    cols = np.repeat(np.arange(n), k)
    long_rows = np.arange(200) if head else rng.choice(m, 200, replace=False)
    lr = np.repeat(long_rows, 3_000)
    lc = rng.integers(0, n, size=lr.size)
    A = sp.csr_matrix((rng.standard_normal(rows.size + lr.size), (np.r_[rows, lr], np.r_[cols, lc])), shape=(m, n))
    x, y, b = rng.random(n), rng.standard_normal(m), rng.standard_normal(m)
    ctx.set_option("rowblock", 0)
    ctx.set_option("slabs", 0)
    dA = ctx.row_shard(A)
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.score_rows(dA, ctx.to_device(x), ctx.to_device(b), ctx.to_device(y), 1e-3, s_p, flag)
    got = s_p.download()
    dA.free()
    ctx.set_option("rowblock", -1)
    ctx.set_option("slabs", -1)
    assert bits_equal(got, L.primal_slack(A, b, x))
