"""Host (numpy) builder of the column-blocked row layout of csrc/sx_rowblock.h -- the reference builder the
device builder is tested against, and the one tools/rb_bench.py uses for layout experiments.

    build(A_csr, R, cwin, chunk, dense_min, budget) -> dict of arrays (see sx_rowblock.h)
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

NO_WINDOW = -(1 << 30)
CELL_MAX = 65535

CHUNK_DT = np.dtype([("e0", "<i8"), ("ne", "<i4"), ("col0", "<i4"), ("cell", "<i4"), ("base", "<i4"),
                     ("fresh", "<i4"), ("pad", "<i4")])
ST_DT = np.dtype([("row0", "<i8"), ("chunk0", "<i8"), ("nrows", "<i4"), ("nchunks", "<i4")])


LONG_ROW = 512       # a row with more entries than this is "long"
LONG_BUDGET = 65536  # entries per super-tile of long rows
LONG_ROWS = 64       # rows per super-tile of long rows


def supertile_cuts(indptr: np.ndarray, R: int, budget: int) -> np.ndarray:
    """Row cuts.  Runs of ordinary rows: <= R rows and <= budget entries per super-tile.  Long rows (more
    than LONG_ROW entries, e.g. the linking rows of a block-angular LP) are summed by one lane each, entry
    after entry, chunk after chunk -- their parallelism has to come from the number of workgroups, so they
    go into super-tiles of <= LONG_ROWS rows / LONG_BUDGET entries (a single longer row stands alone), which
    build() cuts into cells by position inside the row instead of by column, so that every staged chunk
    holds a slice of every row of the super-tile and all its lanes add at the same time."""
    m = indptr.size - 1
    length = np.diff(indptr)
    is_long = length > LONG_ROW
    # boundaries between runs of long / ordinary rows
    change = np.flatnonzero(np.diff(is_long.astype(np.int8))) + 1
    run_starts = np.concatenate([[0], change, [m]])
    cuts = [0]
    for a, b in zip(run_starts[:-1], run_starts[1:]):
        if a == b:
            continue
        lim_rows, lim_entries = (min(R, LONG_ROWS), LONG_BUDGET) if is_long[a] else (R, budget)
        row = int(a)
        while row < b:
            r_end = min(row + lim_rows, int(b))
            r_b = int(np.searchsorted(indptr, indptr[row] + lim_entries, side="right")) - 1
            nxt = max(row + 1, min(r_end, r_b))
            cuts.append(nxt)
            row = nxt
    return np.asarray(cuts, dtype=np.int64)


def build(A: sp.csr_matrix, R: int = 1024, cwin: int = 4096, chunk: int = 4096, dense_min: int = 512,
          budget: int = 98304, merge_max: int = 32768):
    A = sp.csr_matrix(A)
    m, n = A.shape
    indptr = A.indptr.astype(np.int64)
    col = A.indices.astype(np.int64)
    nnz = col.size
    rows_sorted = True
    if nnz:
        d = np.diff(col)
        inner = np.ones(nnz - 1, dtype=bool)
        inner[indptr[1:-1][(indptr[1:-1] > 0) & (indptr[1:-1] < nnz)] - 1] = False
        rows_sorted = bool(np.all(d[inner] >= 0))
    if not rows_sorted:
        raise ValueError("row-block layout needs non-descending columns inside every row")
    cuts = supertile_cuts(indptr, R, budget)
    nst = cuts.size - 1
    st_nrows = np.diff(cuts)
    st_e = indptr[cuts]
    st_of_entry = np.repeat(np.arange(nst, dtype=np.int64), np.diff(st_e))
    row_of_entry = np.repeat(np.arange(m, dtype=np.int64), np.diff(indptr))
    lrow = row_of_entry - cuts[st_of_entry]
    blk = col // cwin
    # super-tiles of long rows: "block" = slice of the row by position, chunk // nrows entries of each row
    long_st = (np.diff(indptr)[cuts[:-1]] > LONG_ROW) if nst else np.zeros(0, dtype=bool)
    if long_st.any():
        slice_len = np.maximum(1, chunk // np.maximum(st_nrows, 1))
        e_long = long_st[st_of_entry]
        pos = np.arange(nnz, dtype=np.int64) - indptr[row_of_entry]
        blk = np.where(e_long, pos // slice_len[st_of_entry], blk)
    nblk = int(blk.max()) + 1 if nnz else 1
    key = st_of_entry * nblk + blk
    # occupied (super-tile, block) bins in ascending order
    ukey, cnt = np.unique(key, return_counts=True)
    u_st = ukey // nblk
    u_blk = ukey % nblk
    bin_long = long_st[u_st] if ukey.size else np.zeros(0, dtype=bool)
    dense = (cnt >= dense_min) & ~bin_long          # gets an LDS window
    alone = dense | bin_long                        # never merged with its neighbours
    # cells: a dense bin is a cell of its own; consecutive sparse bins of one super-tile merge while the
    # total stays <= merge_max
    new_cell = np.ones(ukey.size, dtype=bool)
    run = 0
    # vectorised greedy is awkward; the number of bins is small (~ nst * span / cwin)
    prev_st, prev_alone = -1, True
    for i in range(ukey.size):
        if u_st[i] == prev_st and not alone[i] and not prev_alone and run + cnt[i] <= merge_max:
            new_cell[i] = False
            run += cnt[i]
        else:
            run = cnt[i]
        prev_st, prev_alone = u_st[i], alone[i]
    cell_of_bin = np.cumsum(new_cell) - 1
    ncells = int(cell_of_bin[-1]) + 1 if ukey.size else 0
    first_bin = np.flatnonzero(new_cell)
    cell_st = u_st[first_bin]
    cell_col0 = np.where(dense[first_bin], u_blk[first_bin] * cwin, NO_WINDOW).astype(np.int64)
    cell_ne = np.add.reduceat(cnt, first_bin) if ukey.size else np.zeros(0, dtype=np.int64)
    if cell_ne.size and cell_ne.max() > CELL_MAX:
        raise ValueError(f"a cell holds {cell_ne.max()} entries (> {CELL_MAX}): lower R or the budget")
    # per-entry cell, then the stable sort by (cell, local row): CSR order is (row, col), so inside a
    # (cell, row) group columns stay ascending
    cell_of_entry = cell_of_bin[np.searchsorted(ukey, key)]
    order = np.argsort(cell_of_entry * (R + 1) + lrow, kind="stable")
    # padded cell offsets (multiples of 4)
    cell_pad = (cell_ne + 3) & ~3
    cell_e0 = np.zeros(ncells + 1, dtype=np.int64)
    np.cumsum(cell_pad, out=cell_e0[1:])
    total = int(cell_e0[-1])
    sorted_cell = cell_of_entry[order]
    cell_first_sorted = np.zeros(ncells + 1, dtype=np.int64)
    np.cumsum(cell_ne, out=cell_first_sorted[1:])
    pos = cell_e0[sorted_cell] + (np.arange(nnz, dtype=np.int64) - cell_first_sorted[sorted_cell])
    idx = np.zeros(total + 8, dtype=np.int32)
    val = np.zeros(total + 8, dtype=np.float64)
    idx[pos] = A.indices[order]
    val[pos] = A.data[order]
    # the gap behind every cell: (col0, 0.0) pairs (column 0 for a direct cell), so a staged pad reads inside
    # the window like every real entry of the cell
    gap = cell_pad - cell_ne
    if gap.any():
        gcell = np.repeat(np.arange(ncells, dtype=np.int64), gap)
        gfirst = np.zeros(ncells + 1, dtype=np.int64)
        np.cumsum(gap, out=gfirst[1:])
        gpos = cell_e0[gcell] + cell_ne[gcell] + (np.arange(gcell.size, dtype=np.int64) - gfirst[gcell])
        idx[gpos] = np.where(cell_col0[gcell] != NO_WINDOW, cell_col0[gcell], 0).astype(np.int32)
    # row starts per cell
    stride = R + 4
    hist = np.bincount(sorted_cell * stride + lrow[order] + 1, minlength=ncells * stride).reshape(ncells, stride)
    rowstart = np.cumsum(hist, axis=1)
    # rows beyond the super-tile's count keep the cell total (cumsum does that already)
    rowstart = rowstart.astype(np.uint16)
    st, ch = _chunks(cuts, st_nrows, cell_st, cell_ne, cell_e0, cell_col0, chunk)
    nchunks = ch.size
    stats = {"nst": nst, "ncells": ncells, "nchunks": nchunks, "entries_padded": total,
             "windowed_entries": int(cell_ne[cell_col0 != NO_WINDOW].sum()), "nnz": int(nnz),
             "windowed_cells": int((cell_col0 != NO_WINDOW).sum()),
             "rowstart_bytes": int(rowstart.nbytes)}
    return {"st": st, "chunks": ch, "rowstart": rowstart, "rs_stride": stride, "idx": idx, "val": val, "stats": stats,
            "R": R, "cwin": cwin, "chunk": chunk,
            "_cells": (cuts, st_nrows, cell_st, cell_ne, cell_e0, cell_col0)}


def _chunks(cuts, st_nrows, cell_st, cell_ne, cell_e0, cell_col0, chunk):
    nst, ncells = cuts.size - 1, cell_ne.size
    nch_cell = (cell_ne + chunk - 1) // chunk
    nchunks = int(nch_cell.sum())
    ch = np.zeros(nchunks, dtype=CHUNK_DT)
    ch_cell = np.repeat(np.arange(ncells, dtype=np.int64), nch_cell)
    ch_first = np.zeros(ncells + 1, dtype=np.int64)
    np.cumsum(nch_cell, out=ch_first[1:])
    k_in_cell = np.arange(nchunks, dtype=np.int64) - ch_first[ch_cell]
    ch["base"] = k_in_cell * chunk
    ch["e0"] = cell_e0[ch_cell] + ch["base"]
    ch["ne"] = np.minimum(chunk, cell_ne[ch_cell] - ch["base"])
    ch["col0"] = cell_col0[ch_cell]
    ch["cell"] = ch_cell
    ch["fresh"] = (k_in_cell == 0)
    st = np.zeros(nst, dtype=ST_DT)
    st["row0"] = cuts[:-1]
    st["nrows"] = st_nrows
    cells_per_st = np.bincount(cell_st, minlength=nst) if ncells else np.zeros(nst, dtype=np.int64)
    st_cell0 = np.zeros(nst + 1, dtype=np.int64)
    np.cumsum(cells_per_st, out=st_cell0[1:])
    st["chunk0"] = ch_first[st_cell0[:-1]]
    st["nchunks"] = ch_first[st_cell0[1:]] - ch_first[st_cell0[:-1]]
    return st, ch


def rechunk(L, chunk: int):
    """The same cells cut into chunks of another size (cheap)."""
    st, ch = _chunks(*L["_cells"], chunk)
    out = dict(L)
    out.update(st=st, chunks=ch, chunk=chunk)
    return out


def reference_rows(L, x: np.ndarray, m: int) -> np.ndarray:
    """Row sums computed from the layout on the host, sequentially per row (slow; tests only)."""
    out = np.zeros(m)
    ch, st, rs = L["chunks"], L["st"], L["rowstart"]
    for s in st:
        for k in range(s["chunk0"], s["chunk0"] + s["nchunks"]):
            c = ch[k]
            for r in range(s["nrows"]):
                k0 = max(int(rs[c["cell"], r]), int(c["base"]))
                k1 = min(int(rs[c["cell"], r + 1]), int(c["base"] + c["ne"]))
                for e in range(k0, k1):
                    p = int(c["e0"]) - int(c["base"]) + e
                    out[s["row0"] + r] = out[s["row0"] + r] + L["val"][p] * x[L["idx"][p]]
    return out
