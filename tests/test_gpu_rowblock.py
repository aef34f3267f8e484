"""GPU: the column-blocked row layout (csrc/sx_rowblock.h) -- the device builder against its numpy statement
(tools/rb_layout.py), and the row walk over it against the plain walk and the CPU oracle, bit for bit
(K2: reference formats.py:74-76 + lp_methods/algorithms.py:106; the sums must be the sequential, separately
rounded sums of scipy's csr_matvec whatever the layout)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, bits_equal
from oracle import lp_path as L
import workloads

sys.path.insert(0, os.path.join(ROOT, "tools"))
import rb_layout as RB  # noqa: E402

pytestmark = pytest.mark.gpu

PARAMS = dict(R=512, cwin=4096, chunk=2048, dense_min=512, budget=96 * 512, merge_max=32768)


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def k2(ctx, dA, x, b, y, gamma_dual=1e-3):
    m = dA.shape[0]
    s_p, flag = ctx.empty(m, np.float64), ctx.empty(m, np.uint8)
    ctx.score_rows(dA, ctx.to_device(x), ctx.to_device(b), ctx.to_device(y), gamma_dual, s_p, flag)
    return s_p.download(), flag.download()


def both_walks(ctx, A, x, b, y):
    """(plain, blocked, layout info): the same matrix scored with the layout off and forced."""
    ctx.set_option("rowblock", 0)
    dA = ctx.matrix(A)
    plain = k2(ctx, dA, x, b, y)
    dA.free()
    ctx.set_option("rowblock", 1)
    dB = ctx.matrix(A)
    blocked = k2(ctx, dB, x, b, y)
    info = dB.rowblock()
    dB.free()
    ctx.set_option("rowblock", -1)
    return plain, blocked, info


def staircase(m, nb, window=4096, seed=5):
    sh = workloads.lp_shard(0, 1, m=m, n_block=nb, k=8, seed=seed, structure="staircase", window=window)
    return sh.row_block, sh.x, sh.b, sh.y[:m]


def test_device_layout_equals_numpy_builder(ctx):
    A, x, b, y = staircase(64_000, 640_000)
    ref = RB.build(A, **PARAMS)
    ctx.set_option("rowblock", 1)
    dA = ctx.matrix(A)
    got = dA.rowblock(download=True)
    ctx.set_option("rowblock", -1)
    assert got is not None
    st = ref["stats"]
    assert (got["nst"], got["ncells"], got["nchunks"], got["nent"], got["windowed"]) == \
        (st["nst"], st["ncells"], st["nchunks"], st["entries_padded"], st["windowed_entries"])
    assert got["rs_stride"] == ref["rs_stride"]
    for f in ("row0", "chunk0", "nrows", "nchunks"):
        assert np.array_equal(got["st"][f], ref["st"][f]), f
    for f in ("e0", "ne", "col0", "cell", "base", "fresh"):
        assert np.array_equal(got["chunks"][f], ref["chunks"][f]), f
    assert np.array_equal(got["rowstart"], ref["rowstart"])
    assert np.array_equal(got["idx"], ref["idx"]) and bits_equal(got["val"], ref["val"])
    # long rows exist in this instance and were sliced by position: every chunk of such a super-tile holds
    # a piece of (nearly) every row
    long_st = got["st"][got["st"]["nrows"] <= RB.LONG_ROWS]
    assert long_st.size > 0
    dA.free()


@pytest.mark.parametrize("m,nb,window", [(64_000, 640_000, 4096), (20_000, 200_000, 512), (3_000, 30_000, 64)])
def test_blocked_walk_is_bit_identical_on_staircase_lps(ctx, m, nb, window):
    A, x, b, y = staircase(m, nb, window)
    plain, blocked, info = both_walks(ctx, A, x, b, y)
    assert info is not None and info["windowed"] > 0
    assert bits_equal(plain[0], blocked[0]) and np.array_equal(plain[1], blocked[1])
    want = L.primal_slack(A, b, x)
    assert bits_equal(blocked[0], want)
    assert np.array_equal(blocked[1], L.row_flags(want, y, 1e-3))


def test_blocked_walk_on_unstructured_and_ragged_rows(ctx):
    rng = np.random.default_rng(7)
    m, n = 5_000, 60_000
    # rows of very different lengths incl. empty ones and a few long ones (some longer than a chunk)
    lens = rng.integers(0, 40, size=m)
    lens[rng.choice(m, 40, replace=False)] = rng.integers(600, 9000, size=40)
    lens[:3] = 0
    lens[-2:] = 0
    rows = np.repeat(np.arange(m), lens)
    cols = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for k in lens]) if lens.sum() else np.zeros(0, int)
    vals = rng.uniform(-1, 1, size=rows.size)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    A.sort_indices()
    x, b, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    x[rng.choice(n, 50)] = np.inf          # non-finite operands travel through the same sums
    plain, blocked, info = both_walks(ctx, A, x, b, y)
    assert info is not None
    assert bits_equal(plain[0], blocked[0]) and np.array_equal(plain[1], blocked[1])
    assert bits_equal(blocked[0], L.primal_slack(A, b, x))


def test_duplicate_columns_keep_their_stored_order(ctx):
    rng = np.random.default_rng(8)
    m, n, k = 2_000, 9_000, 12
    cols = np.sort(rng.integers(0, n, size=(m, k)), axis=1)      # duplicates inside rows are likely
    A = sp.csr_matrix((rng.uniform(-1, 1, m * k), cols.ravel(), np.arange(m + 1) * k), shape=(m, n))
    x, b, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    plain, blocked, info = both_walks(ctx, A, x, b, y)
    assert info is not None
    assert bits_equal(plain[0], blocked[0]) and np.array_equal(plain[1], blocked[1])
    assert bits_equal(blocked[0], b - L.seq_segment_sums(A.indptr, A.indices, A.data, x))


def test_descending_rows_keep_the_plain_walk(ctx):
    rng = np.random.default_rng(9)
    m, n, k = 1_500, 8_000, 10
    cols = np.sort(rng.integers(0, n, size=(m, k)), axis=1)
    cols[7] = cols[7][::-1]                                       # one row stored backwards
    A = sp.csr_matrix((rng.uniform(-1, 1, m * k), cols.ravel(), np.arange(m + 1) * k), shape=(m, n))
    x, b, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(m)
    plain, blocked, info = both_walks(ctx, A, x, b, y)
    assert info is None                                           # refused: the order of that row would change
    assert bits_equal(plain[0], blocked[0])
    assert bits_equal(blocked[0], b - L.seq_segment_sums(A.indptr, A.indices, A.data, x))


def test_auto_rule_leaves_small_and_unstructured_matrices_alone(ctx):
    A, x, b, y = staircase(3_000, 30_000, 64)        # 240k entries: below the automatic threshold
    dA = ctx.matrix(A)
    assert dA.rowblock() is None
    dA.free()


def test_projector_cg_over_the_layout_matches_the_plain_walk(ctx):
    inst = workloads.sparse_lp(3_000, 12_000, 8, seed=31, stratified=False, frac_upper=0.3)
    A = inst.A
    m, n = A.shape
    rng = np.random.default_rng(3)
    xa, xs, c = rng.uniform(0.1, 1, n), rng.uniform(0.1, 1, m), rng.standard_normal(n)
    out = {}
    for opt in (0, 1):
        ctx.set_option("rowblock", opt)
        dA = ctx.matrix(A)
        pc, pr = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
        res = ctx.projector_norm(dA, ctx.to_device(xa), ctx.to_device(xs), ctx.to_device(c), 1e-10, 2000, pc, pr)
        out[opt] = (res.proj_norm, int(res.iters), int(res.converged), pc.download(), pr.download(),
                    dA.rowblock() is not None)
        dA.free()
    ctx.set_option("rowblock", -1)
    assert out[1][5] and not out[0][5]
    assert out[0][2] == 1 and out[1][2] == 1
    assert out[1][0] == pytest.approx(out[0][0], rel=1e-9)
    np.testing.assert_allclose(out[1][3], out[0][3], rtol=1e-6, atol=1e-9 * np.abs(out[0][3]).max())
    np.testing.assert_allclose(out[1][4], out[0][4], rtol=1e-6, atol=1e-9 * max(np.abs(out[0][4]).max(), 1e-300))


def test_long_row_super_tiles_dealt_over_the_xcds_change_nothing_but_the_order_of_work(ctx):
    """Option rb_long_xcd (sx_rowblock.hip rb_build_order): a netlib-style LP with all its linking rows at the head puts every
    long-row super-tile on XCD 0 under the XCD-contiguous map; dealt over the eight XCDs (1: always, -1: the automatic rule,
    which fires here) the walk visits the same super-tiles in another order -- s_p and the flags are the plain walk's, bit
    for bit -- and the CG's row pass over the same map ends on the same projector norm."""
    inst = workloads.netlib_lp(70_000, 700_000, seed=2)
    A, x, b, y = inst.A, inst.x, inst.b, inst.y
    m, n = A.shape
    ctx.set_option("rowblock", 0)
    dA = ctx.matrix(A)
    want = k2(ctx, dA, x, b, y)
    dA.free()
    ctx.set_option("rowblock", 1)
    try:
        for mode in (0, 1, -1):
            ctx.set_option("rb_long_xcd", mode)
            dB = ctx.matrix(A)
            got = k2(ctx, dB, x, b, y)
            assert dB.rowblock() is not None and dB.rowblock()["nst"] >= 64
            assert np.array_equal(got[0].view(np.uint64), want[0].view(np.uint64)) and np.array_equal(got[1], want[1]), mode
            dB.free()
        rng = np.random.default_rng(5)
        xa, xs, c = rng.uniform(0.1, 1, n), rng.uniform(0.1, 1, m), rng.standard_normal(n)
        norms = {}
        for mode in (0, 1):
            ctx.set_option("rb_long_xcd", mode)
            dB = ctx.matrix(A)
            pc, pr = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
            res = ctx.projector_norm(dB, ctx.to_device(xa), ctx.to_device(xs), ctx.to_device(c), 1e-10, 300, pc, pr)
            norms[mode] = (res.proj_norm, int(res.iters))
            dB.free()
        assert norms[0][1] == norms[1][1] and norms[1][0] == pytest.approx(norms[0][0], rel=1e-12)
    finally:
        ctx.set_option("rb_long_xcd", -1)
        ctx.set_option("rowblock", -1)
