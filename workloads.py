"""Seeded synthetic instances of the BASELINE.json configurations.

Shared by tests/, tests/golden/make_golden.py and bench.py.  Pure numpy/scipy;
no dependency on the product package or on the oracle.

  c1  afiro-size LP            m=27,   n=51
  c2  random sparse LP         m=2e4,  n=1e5, 20 nnz/col
  c3  OT on the 28x28 grid     S=D=784
  c4  GOTO-like MCF            V=2^17, E=2^20
  c5  netlib-style LP          m=1e6,  n=1e7, 8 nnz/col
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import scipy.sparse as sp


@dataclass
class LPInstance:
    A: sp.csr_matrix          # m x n, CSR, sorted indices, no duplicates
    b: np.ndarray
    c: np.ndarray
    l: np.ndarray
    u: np.ndarray
    sense: np.ndarray         # '=' / '<'
    x: np.ndarray             # synthetic interior primal point
    y: np.ndarray             # synthetic interior dual point
    name: str = "synthetic_lp"


def _stratified_rows(rng, m: int, n: int, k: int) -> np.ndarray:
    """k distinct ascending rows per column: one uniform pick in each of k
    equal strata of [0, m).  Shape (n, k), int32."""
    width = m // k
    base = (np.arange(k, dtype=np.int64) * width)[None, :]
    pick = rng.integers(0, width, size=(n, k), dtype=np.int64)
    return (base + pick).astype(np.int32)


def _sampled_rows(rng, m: int, n: int, k: int) -> np.ndarray:
    """k distinct ascending rows per column drawn without replacement
    (argpartition of random keys); only for small m*n."""
    keys = rng.random((n, m))
    rows = np.argpartition(keys, k - 1, axis=1)[:, :k]
    return np.sort(rows, axis=1).astype(np.int32)


def sparse_lp(m: int, n: int, nnz_per_col: int, seed: int, frac_lt: float = 0.5,
              frac_upper: float = 0.25, frac_free: float = 0.0, stratified: Optional[bool] = None,
              name: str = "synthetic_lp") -> LPInstance:
    """Random sparse LP with a synthetic interior point (x, y) shaped like a
    late barrier iterate: m random columns are
    "basic" (x in U(0.1, 1), dual slack ~1e-10), the others sit 1e-9 away from
    a bound with a dual slack |N(0,1)|; y ~ N(0,1); a ``frac_lt`` share of the
    rows is '<', and of those the ones with y < 0 are tight.
    """
    rng = np.random.default_rng(seed)
    k = nnz_per_col
    if stratified is None:
        stratified = m * n > 5_000_000
    rows = _stratified_rows(rng, m, n, k) if stratified else _sampled_rows(rng, m, n, k)
    vals = rng.uniform(-1.0, 1.0, size=(n, k))
    vals[np.abs(vals) < 1e-3] = 0.5
    colptr = np.arange(n + 1, dtype=np.int64) * k
    C = sp.csc_matrix((vals.ravel(), rows.ravel(), colptr), shape=(m, n))
    A = C.tocsr()
    A.sort_indices()

    y = rng.standard_normal(m)
    sense = np.where(rng.random(m) < frac_lt, "<", "=")
    # for '<' rows the dual must be <= 0 in "min c^T x, Ax <= b"; keep the sign
    # random so that both tight (y<0) and slack (y~0) rows appear
    lt = sense == "<"
    y[lt] = np.where(rng.random(int(lt.sum())) < 0.5, -np.abs(y[lt]), 1e-11 * np.abs(y[lt]))

    l = np.zeros(n)
    u = np.full(n, np.inf)
    has_up = rng.random(n) < frac_upper
    u[has_up] = rng.uniform(1.0, 10.0, int(has_up.sum()))
    if frac_free > 0:
        free = rng.random(n) < frac_free
        l[free] = -np.inf
        u[free] = np.inf

    basic = np.zeros(n, dtype=bool)
    basic[rng.choice(n, size=min(m, n), replace=False)] = True
    x = np.full(n, 1e-9)
    x[basic] = rng.uniform(0.1, 1.0, int(basic.sum()))
    at_up = (~basic) & has_up & (rng.random(n) < 0.5)
    x[at_up] = u[at_up] - 1e-9

    s_d = np.abs(rng.standard_normal(n))
    s_d[basic] = 1e-10 * rng.random(int(basic.sum()))
    s_d[at_up] = -s_d[at_up]
    c = A.T @ y + s_d

    slack = np.zeros(m)
    loose = lt & (y > -1e-9)
    slack[loose] = rng.uniform(0.1, 1.0, int(loose.sum()))
    tight = lt & ~loose
    slack[tight] = 1e-10 * rng.random(int(tight.sum()))
    b = A @ x + slack
    return LPInstance(A=A, b=b, c=c, l=l, u=u, sense=sense, x=x, y=y, name=name)


def config1(seed: int = 2024) -> LPInstance:
    """afiro-size: m=27, n=51, 2 nnz/col (~102 nnz), entries from {+-1, 2, 0.5}."""
    inst = sparse_lp(27, 51, 2, seed, frac_lt=0.4, frac_upper=0.3, stratified=False, name="afiro_size")
    rng = np.random.default_rng(seed + 1)
    A = inst.A.copy()
    A.data = rng.choice(np.array([1.0, -1.0, 2.0, 0.5]), size=A.nnz)
    # rebuild c and b so that (x, y) keeps its interior-point shape
    s_d = inst.c - inst.A.T @ inst.y
    slack = inst.b - inst.A @ inst.x
    return LPInstance(A=A, b=A @ inst.x + slack, c=A.T @ inst.y + s_d, l=inst.l, u=inst.u, sense=inst.sense,
                      x=inst.x, y=inst.y, name="afiro_size")


def config2(seed: int = 2) -> LPInstance:
    return sparse_lp(20_000, 100_000, 20, seed, stratified=True, name="c2_2e4x1e5")


def config5(seed: int = 5, n: int = 10_000_000, m: int = 1_000_000) -> LPInstance:
    return sparse_lp(m, n, 8, seed, stratified=True, name="c5_1e6x1e7")


# --------------------------------------------------------------------------
@dataclass
class MCFInstance:
    A: sp.csr_matrix          # V x E incidence: +1 at the tail, -1 at the head (scripts/min2mcf.py convention)
    b: np.ndarray
    c: np.ndarray
    u: np.ndarray
    x: np.ndarray             # inexact interior flow
    tail: np.ndarray
    head: np.ndarray
    name: str = "synthetic_mcf"


def mcf(V: int, E: int, seed: int = 3, frac_interior: float = 0.15) -> MCFInstance:
    """GOTO-like min-cost-flow instance."""
    rng = np.random.default_rng(seed)
    tail = rng.integers(0, V, size=E, dtype=np.int64)
    head = (tail + rng.integers(1, V, size=E, dtype=np.int64)) % V
    u = rng.integers(1, 1000, size=E).astype(np.float64)
    c = rng.integers(1, 10000, size=E).astype(np.float64)
    x = np.empty(E)
    interior = rng.random(E) < frac_interior
    x[interior] = rng.uniform(0.01, 0.99, int(interior.sum())) * u[interior]
    near_up = (~interior) & (rng.random(E) < 0.3)
    near_lo = (~interior) & ~near_up
    x[near_lo] = 1e-7 * u[near_lo] * rng.random(int(near_lo.sum()))
    x[near_up] = u[near_up] * (1 - 1e-7 * rng.random(int(near_up.sum())))
    arc = np.arange(E, dtype=np.int64)
    A = sp.csr_matrix((np.concatenate([np.ones(E), -np.ones(E)]),
                       (np.concatenate([tail, head]), np.concatenate([arc, arc]))), shape=(V, E))
    A.sort_indices()
    b = A @ x
    b[-1] -= b.sum()          # exact balance up to one rounding
    return MCFInstance(A=A, b=b, c=c, u=u, x=x, tail=tail, head=head, name=f"mcf_{V}x{E}")


def config4(seed: int = 3) -> MCFInstance:
    return mcf(2 ** 17, 2 ** 20, seed)


# --------------------------------------------------------------------------
@dataclass
class OTInstance:
    s: np.ndarray
    d: np.ndarray
    M: np.ndarray
    x: np.ndarray             # flattened S*D inexact plan (Sinkhorn-like scaling sweeps)
    name: str = "synthetic_ot"


def grid_cost(side: int) -> np.ndarray:
    """Manhattan distance between the cells of a side x side grid (the cost
    scripts/mnist2ot.py builds for k = 1)."""
    r, cc = np.divmod(np.arange(side * side), side)
    return (np.abs(r[:, None] - r[None, :]) + np.abs(cc[:, None] - cc[None, :])).astype(np.float64)


def ot(S: int, D: int, seed: int = 7, sweeps: int = 50, M: Optional[np.ndarray] = None) -> OTInstance:
    rng = np.random.default_rng(seed)
    if M is None:
        M = rng.integers(1, 50, size=(S, D)).astype(np.float64)
    s = rng.random(S) + 0.05
    d = rng.random(D) + 0.05
    s /= s.sum()
    d /= d.sum()
    d[-1] += s.sum() - d.sum()
    K = np.exp(-M / 3.0)
    a = np.ones(S)
    bb = np.ones(D)
    for _ in range(sweeps):
        a = s / (K @ bb)
        bb = d / (K.T @ a)
    X = a[:, None] * K * bb[None, :]
    return OTInstance(s=s, d=d, M=M, x=X.ravel(), name=f"ot_{S}x{D}")


def config3(seed: int = 7) -> OTInstance:
    return ot(784, 784, seed, M=grid_cost(28))


# --------------------------------------------------------------------------
# config 5, sharded: one column block and one row block per rank (weak scaling)
# --------------------------------------------------------------------------
@dataclass
class LPShard:
    """What one rank holds of the global m x (world*n_block) LP."""
    rank: int
    world: int
    m: int
    n_block: int
    col_block: sp.csc_matrix      # m x n_block, columns [rank*n_block, (rank+1)*n_block), walk order
    row_block: sp.csr_matrix      # (m/world) x (world*n_block), rows [rank*m/world, (rank+1)*m/world)
    y: np.ndarray                 # m            (replicated)
    x: np.ndarray                 # world*n_block (replicated: it is the input of the crossover)
    c: np.ndarray                 # n_block      (this rank's columns)
    l: np.ndarray
    u: np.ndarray
    b: np.ndarray                 # m/world      (this rank's rows)


def _stratum(seed: int, q: int, t: int, width: int, n_block: int):
    rng = np.random.default_rng([seed, q, t])
    pick = rng.integers(0, width, size=n_block, dtype=np.int32)
    val = rng.uniform(-1.0, 1.0, size=n_block)
    val[np.abs(val) < 1e-3] = 0.5
    return pick, val


def _block_vectors(seed: int, q: int, n_block: int):
    rng = np.random.default_rng([seed, q, 1000])
    has_up = rng.random(n_block) < 0.25
    u = np.full(n_block, np.inf)
    u[has_up] = rng.uniform(1.0, 10.0, int(has_up.sum()))
    l = np.zeros(n_block)
    basic = rng.random(n_block) < 0.1
    x = np.full(n_block, 1e-9)
    x[basic] = rng.uniform(0.1, 1.0, int(basic.sum()))
    at_up = (~basic) & has_up & (rng.random(n_block) < 0.5)
    x[at_up] = u[at_up] - 1e-9
    s_d = np.abs(rng.standard_normal(n_block))
    s_d[basic] = 1e-10 * rng.random(int(basic.sum()))
    s_d[at_up] = -s_d[at_up]
    return x, l, u, s_d


def _uniform_blocks(rank, world, m, n_block, k, seed):
    """Column block + row block when every column has one entry in each of k equal row strata
    (rows uniformly random inside a stratum): no locality at all, the worst case for the gathers."""
    width = m // k
    n_total = world * n_block
    idx = np.empty((n_block, k), dtype=np.int32)
    val = np.empty((n_block, k), dtype=np.float64)
    for t in range(k):
        pick, v = _stratum(seed, rank, t, width, n_block)
        idx[:, t] = pick + t * width
        val[:, t] = v
    colptr = np.arange(n_block + 1, dtype=np.int64) * k
    col_block = sp.csc_matrix((val.ravel(), idx.ravel(), colptr), shape=(m, n_block))
    per = k // world
    t0 = rank * per
    m_loc = per * width
    rows, cols, vals = [], [], []
    for t in range(t0, t0 + per):
        for q in range(world):
            if q == rank:
                pick, v = idx[:, t] - t * width, val[:, t]
            else:
                pick, v = _stratum(seed, q, t, width, n_block)
            rows.append(pick + np.int32((t - t0) * width))
            cols.append(np.arange(q * n_block, (q + 1) * n_block, dtype=np.int32))
            vals.append(v)
    row_block = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                              shape=(m_loc, n_total))
    row_block.sort_indices()
    return col_block, row_block


STAIR_REGIONS = 8     # row regions of the default global LP; every rank owns 8/world of them


def _stair_geometry(m: int, window: int, regions: int = STAIR_REGIONS):
    mr = m // regions                         # rows per region
    nl = max(1, mr // 100)                    # linking rows at the head of each region (1 %)
    ms = mr - nl                              # stage rows of the region
    W = max(6, min(window, ms // 4))
    return mr, nl, ms, W


def _stair_linking(seed: int, q: int, n_block: int, nl: int, regions: int = STAIR_REGIONS):
    """Slot 7 of block q: one linking entry per column (region, row inside the region's linking rows, value)."""
    rng = np.random.default_rng([seed, q, 7])
    region = rng.integers(0, regions, size=n_block, dtype=np.int32)
    off = rng.integers(0, nl, size=n_block, dtype=np.int32)
    val = rng.uniform(-1.0, 1.0, size=n_block)
    val[np.abs(val) < 1e-3] = 0.5
    return region, off, val


def _staircase_blocks(rank, world, m, n_block, k, seed, window, regions=STAIR_REGIONS):
    """Netlib-style structure (staircase / block-angular with linking rows), 8 entries per column:
    six in a window of W stage rows at the column's home position, one in the following window
    (the coupling to the next stage) and one in a linking row (1 % of the rows, shared by all
    stages).  Home positions grow with the column index, so neighbouring columns touch
    neighbouring rows, as in multi-period netlib models."""
    if k != 8:
        raise ValueError("the staircase generator places exactly 8 entries per column")
    G = regions // world                      # regions owned by this rank
    if n_block % G:
        raise ValueError("n_block must be divisible by the number of regions per rank")
    mr, nl, ms, W = _stair_geometry(m, window, regions)
    ws = W // 6
    per_region = n_block // G
    j = np.arange(n_block, dtype=np.int64)
    g_loc = (j // per_region).astype(np.int32)                  # region (local) of the column's stage rows
    home = ((j % per_region) * (ms - 2 * W) // per_region).astype(np.int32)
    g_glob = g_loc + rank * G
    base = g_glob * mr + nl + home                               # first row of the column's window
    idx = np.empty((n_block, 8), dtype=np.int32)
    val = np.empty((n_block, 8), dtype=np.float64)
    for t in range(7):
        rng = np.random.default_rng([seed, rank, t])
        if t < 6:
            off = t * ws + rng.integers(0, ws, size=n_block, dtype=np.int32)
        else:
            off = W + rng.integers(0, W, size=n_block, dtype=np.int32)
        v = rng.uniform(-1.0, 1.0, size=n_block)
        v[np.abs(v) < 1e-3] = 0.5
        idx[:, t] = base + off
        val[:, t] = v
    region, off7, v7 = _stair_linking(seed, rank, n_block, nl, regions)
    idx[:, 7] = region * mr + off7
    val[:, 7] = v7
    # ascending rows inside every column: the linking entry goes first when its region is not
    # after the column's own region, last otherwise
    first = region <= g_glob
    order = np.where(first[:, None], np.array([7, 0, 1, 2, 3, 4, 5, 6]), np.arange(8))
    idx_sorted = np.take_along_axis(idx, order, axis=1)
    val_sorted = np.take_along_axis(val, order, axis=1)
    colptr = np.arange(n_block + 1, dtype=np.int64) * 8
    col_block = sp.csc_matrix((val_sorted.ravel(), idx_sorted.ravel(), colptr), shape=(m, n_block))
    # ---- row block: stage entries of my own columns + linking entries of every block that land in my regions
    r0 = rank * G * mr
    m_loc = G * mr
    n_total = world * n_block
    rows = [(idx[:, :7] - r0).ravel()]
    cols = [np.repeat(np.arange(rank * n_block, (rank + 1) * n_block, dtype=np.int32), 7)]
    vals = [val[:, :7].ravel()]
    for q in range(world):
        rg, of, vv = (region, off7, v7) if q == rank else _stair_linking(seed, q, n_block, nl, regions)
        mine = (rg >= rank * G) & (rg < (rank + 1) * G)
        rows.append((rg[mine] * mr + of[mine] - r0).astype(np.int32))
        cols.append((np.flatnonzero(mine) + q * n_block).astype(np.int32))
        vals.append(vv[mine])
    row_block = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                              shape=(m_loc, n_total))
    row_block.sort_indices()
    return col_block, row_block


def lp_shard(rank: int, world: int, m: int = 1_000_000, n_block: int = 10_000_000, k: int = 8,
             seed: int = 5, structure: str = "staircase", window: int = 4096,
             regions: int = STAIR_REGIONS) -> LPShard:
    """Rank-local part of one global m x (world*n_block) LP (weak scaling): column block ``rank``
    in CSC and row block ``rank`` (m/world rows) in CSR, generated without generating the other
    ranks' blocks.  ``structure`` is "staircase" (netlib-style, default) or "uniform" (no
    locality).  world must divide ``regions`` (staircase) / k (uniform) and k must divide m.

    ``regions`` is the number of row regions of the global staircase LP.  With the default 8 the
    row count m is shared by all ranks, so rows get denser as ranks (columns) are added; passing
    ``regions = 8 * world`` together with ``m = world * m_1`` grows rows and columns together, which
    keeps every rank's blocks shaped like the single-rank problem (row length, window width)."""
    if structure == "uniform":
        if k % world or m % k:
            raise ValueError("world must divide k and k must divide m")
        col_block, row_block = _uniform_blocks(rank, world, m, n_block, k, seed)
    elif structure == "staircase":
        if regions % world or m % regions:
            raise ValueError("world must divide the number of regions and that number must divide m")
        col_block, row_block = _staircase_blocks(rank, world, m, n_block, k, seed, window, regions)
    else:
        raise ValueError("structure must be 'staircase' or 'uniform'")
    m_loc = row_block.shape[0]
    # ---- vectors
    rng = np.random.default_rng([seed, 7777])
    y = rng.standard_normal(m)
    lt = rng.random(m) < 0.5
    y[lt] = np.where(rng.random(int(lt.sum())) < 0.5, -np.abs(y[lt]), 1e-11 * np.abs(y[lt]))
    slack = np.zeros(m)
    loose = lt & (y > -1e-9)
    slack[loose] = rng.uniform(0.1, 1.0, int(loose.sum()))
    tight = lt & ~loose
    slack[tight] = 1e-10 * rng.random(int(tight.sum()))
    xs = []
    mine = None
    for q in range(world):
        xq, lq, uq, sdq = _block_vectors(seed, q, n_block)
        xs.append(xq)
        if q == rank:
            mine = (lq, uq, sdq)
    x = np.concatenate(xs)
    l, u, s_d = mine
    c = col_block.T @ y + s_d
    r0 = rank * m_loc
    b = row_block @ x + slack[r0:r0 + m_loc]
    return LPShard(rank=rank, world=world, m=m, n_block=n_block, col_block=col_block, row_block=row_block, y=y, x=x,
                   c=c, l=l, u=u, b=b)


# --------------------------------------------------------------------------
# N1: the configuration BASELINE.json's metric is quoted on ("1e6-variable netlib-style LP")
# --------------------------------------------------------------------------
def netlib_lp(m: int = 100_000, n: int = 1_000_000, seed: int = 6, window: int = 48,
              frac_lt: float = 0.5, frac_upper: float = 0.25, name: Optional[str] = None) -> LPInstance:
    """A *consistent* netlib-style (staircase + linking rows) LP with a synthetic late-barrier iterate.

    Structure: 1 % of the rows are linking rows (at the head of the row range), the others stage rows in time
    order.  Every column is an *activity with a balance row of its own*: one entry of magnitude 0.8..1.2 there,
    seven of magnitude <= 0.15 elsewhere -- five in the stage rows within ``window``/2 of its own row, one in the
    following stage (``window``..2 ``window`` rows ahead) and one in a linking row; 1 % of the columns are linking
    activities (own row a linking row, the small entries in a window of stage rows).  Columns are in the order
    of their own rows (n/m columns per row), so neighbouring columns touch neighbouring rows, as in multi-period
    production / inventory models.  Vectors as in ``sparse_lp`` -- unlike ``lp_shard`` (a kernel workload whose
    senses are drawn independently of its slacks), (x, y) here IS a strictly complementary primal-dual pair of
    the LP to 1e-9: b = A x + slack with slack > 0 only on '<' rows whose dual is ~0, c = A^T y + s_d with
    s_d ~ 1e-10 on the m "basic" columns, ONE PER ROW (a random one of the columns that own the row).

    Why own rows: a band matrix with independent U(-1, 1) entries -- the first version of this generator -- has
    bases whose condition number grows exponentially with rows / bandwidth (products of random matrices; 1e29
    at 1e4 rows measured), and so do bases in which many columns share their large entry's row: no solver's
    territory and no netlib model's.  With one basic activity per balance row every basis the crossover can meet
    is column-dominant after a permutation."""
    rng = np.random.default_rng(seed)
    nl = max(1, m // 100)                       # linking rows
    ms = m - nl
    W = max(12, min(window, ms // 8))
    link_col = (np.arange(n) % 100) == 0        # linking activities, spread over the column range
    n_link = int(link_col.sum())
    n_stage = n - n_link
    own = np.empty(n, dtype=np.int64)
    own[~link_col] = nl + (np.arange(n_stage, dtype=np.int64) * ms) // n_stage
    own[link_col] = (np.arange(n_link, dtype=np.int64) * nl) // n_link
    # where the small entries go: a centre in the stage rows (the own row, or for linking activities the
    # position of the column in the column range)
    centre = own.copy()
    centre[link_col] = nl + (np.flatnonzero(link_col).astype(np.int64) * ms) // n
    h = W // 2
    idx = np.empty((n, 8), dtype=np.int64)
    val = rng.uniform(-0.15, 0.15, size=(n, 8))
    val[np.abs(val) < 1e-3] = 0.05
    idx[:, 0] = own
    val[:, 0] = rng.uniform(0.8, 1.2, size=n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    q = max(1, h // 3)
    for t, (lo, hi) in enumerate([(-h, -h + q), (-h + q, -1), (1, q + 1), (q + 1, 2 * q + 1), (2 * q + 1, h + 1)]):
        hi = max(hi, lo + 1)
        idx[:, 1 + t] = centre + rng.integers(lo, hi, size=n)
    idx[:, 6] = centre + W + rng.integers(0, W, size=n)
    idx[:, 7] = rng.integers(0, nl, size=n)
    stage_part = idx[:, 1:7]
    np.clip(stage_part, nl, m - 1, out=stage_part)
    colidx = np.repeat(np.arange(n, dtype=np.int64), 8)
    A = sp.coo_matrix((val.ravel(), (idx.ravel(), colidx)), shape=(m, n)).tocsr()   # (duplicates are summed)
    A.sort_indices()

    y = rng.standard_normal(m)
    sense = np.where(rng.random(m) < frac_lt, "<", "=")
    lt = sense == "<"
    y[lt] = np.where(rng.random(int(lt.sum())) < 0.5, -np.abs(y[lt]), 1e-11 * np.abs(y[lt]))
    l = np.zeros(n)
    u = np.full(n, np.inf)
    has_up = rng.random(n) < frac_upper
    u[has_up] = rng.uniform(1.0, 10.0, int(has_up.sum()))
    # one basic column per row: a random one of the row's own columns
    order = np.argsort(own, kind="stable")
    first = np.searchsorted(own[order], np.arange(m), side="left")
    count = np.searchsorted(own[order], np.arange(m), side="right") - first
    if np.any(count == 0):
        raise ValueError("netlib_lp needs at least one column per row (n >= 100 m / 99)")
    basic = np.zeros(n, dtype=bool)
    basic[order[first + (rng.random(m) * count).astype(np.int64)]] = True
    x = np.full(n, 1e-9)
    x[basic] = rng.uniform(0.1, 1.0, int(basic.sum()))
    at_up = (~basic) & has_up & (rng.random(n) < 0.5)
    x[at_up] = u[at_up] - 1e-9
    s_d = np.abs(rng.standard_normal(n))
    s_d[basic] = 1e-10 * rng.random(int(basic.sum()))
    s_d[at_up] = -s_d[at_up]
    c = A.T @ y + s_d
    slack = np.zeros(m)
    loose = lt & (y > -1e-9)
    slack[loose] = rng.uniform(0.1, 1.0, int(loose.sum()))
    tight = lt & ~loose
    slack[tight] = 1e-10 * rng.random(int(tight.sum()))
    b = A @ x + slack
    return LPInstance(A=A, b=b, c=c, l=l, u=u, sense=sense, x=x, y=y,
                      name=name or f"netlib_{m}x{n}")
