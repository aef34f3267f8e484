// Builder of the column-blocked row layout (sx_rowblock.h) on the device.  One-off per matrix: the cuts of
// the super-tiles are made on the host from the row pointers (a few thousand binary searches), everything
// that touches entries runs on the GPU -- bin histogram, cell assignment, a stable radix sort of the
// entries by (cell, local row), scatter, row starts, chunk table.  tools/rb_layout.py states the same
// construction in numpy; tests/test_gpu_rowblock.py compares the two array by array.
#include "sx_internal.h"
#include "sx_rowblock.h"
#include "sx_segwalk.h"

#include <algorithm>
#include <vector>

namespace {

inline unsigned grid1d(int64_t n, int64_t cap = 1 << 20) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

// ---------------------------------------------------------------- entries -> (bin, local row)
// One workgroup per row tile of the CSR tile table: the tile's row pointers go to LDS, every entry finds its
// row there by bisection.  bin = super-tile * nblk + block, where block = column / RB_CWIN in a super-tile
// of ordinary rows and (position inside the row) / slice in one of long rows.
__global__ __launch_bounds__(SX_WG) void k_rb_bins(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                   const int64_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ col,
                                                   const int64_t *__restrict__ cuts, int64_t nst,
                                                   const int32_t *__restrict__ slice, int64_t nblk,
                                                   int32_t *__restrict__ cnt, int32_t *__restrict__ ebin,
                                                   uint16_t *__restrict__ elrow, int *__restrict__ descending) {
    __shared__ int64_t ptr[SX_WG + 1];
    __shared__ int32_t row_st[SX_WG];
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int64_t r0 = tiles[t], r1 = tiles[t + 1];
    const int nr = static_cast<int>(r1 - r0);
    for (int k = threadIdx.x; k <= nr; k += SX_WG) ptr[k] = rowptr[r0 + k];
    if (static_cast<int>(threadIdx.x) < nr) { // super-tile of my row: last cut <= row
        const int64_t row = r0 + threadIdx.x;
        int64_t lo = 0, hi = nst; // cuts[lo] <= row < cuts[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (cuts[mid] <= row) lo = mid;
            else hi = mid;
        }
        row_st[threadIdx.x] = static_cast<int32_t>(lo);
    }
    __syncthreads();
    const int64_t p_lo = ptr[0], p_hi = ptr[nr];
    for (int64_t e = p_lo + threadIdx.x; e < p_hi; e += SX_WG) {
        int lo = 0, hi = nr; // ptr[lo] <= e < ptr[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ptr[mid] <= e) lo = mid;
            else hi = mid;
        }
        const int64_t st = row_st[lo];
        const int32_t c = col[e];
        if (e > ptr[lo] && col[e - 1] > c) *descending = 1; // benign race: every writer stores 1
        const int32_t sl = slice[st];
        const int64_t blk = sl ? (e - ptr[lo]) / sl : c / RB_CWIN;
        const int64_t bin = st * nblk + blk;
        atomicAdd(&cnt[bin], 1);
        ebin[e] = static_cast<int32_t>(bin);
        elrow[e] = static_cast<uint16_t>(r0 + lo - cuts[st]);
    }
}

// ---------------------------------------------------------------- bins -> cells (one lane per super-tile)
// pass 0: loc[bin] = cell number inside the super-tile, ncell[st] = cells of the super-tile
// pass 1: cellid[bin] = global cell number; cell_ne / cell_col0 / cell_pad / cell_nch per cell
__global__ __launch_bounds__(SX_WG) void k_rb_cells(int pass, int dense_min, int64_t nst, int64_t nblk,
                                                    const int32_t *__restrict__ cnt,
                                                    const int32_t *__restrict__ slice,
                                                    const int64_t *__restrict__ cell_base,
                                                    int32_t *__restrict__ loc, int64_t *__restrict__ ncell,
                                                    int64_t *__restrict__ cell_ne, int32_t *__restrict__ cell_col0,
                                                    int64_t *__restrict__ cell_pad, int64_t *__restrict__ cell_nch,
                                                    unsigned long long *__restrict__ totals /* [0] windowed, [1] oversize */) {
    const int64_t st = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (st >= nst) return;
    const bool is_long = slice[st] != 0;
    const int64_t base = pass ? cell_base[st] : 0;
    int64_t local = 0, run = 0;
    bool prev_alone = true, started = false;
    int64_t cur_ne = 0;
    int32_t cur_col0 = RB_NO_WINDOW;
    unsigned long long windowed = 0, oversize = 0;
    auto close_cell = [&]() {
        if (!pass || !started) return;
        const int64_t cell = base + local - 1;
        cell_ne[cell] = cur_ne;
        cell_col0[cell] = cur_col0;
        cell_pad[cell] = (cur_ne + 3) & ~static_cast<int64_t>(3);
        cell_nch[cell] = (cur_ne + RB_CHUNK - 1) / RB_CHUNK;
        if (cur_col0 != RB_NO_WINDOW) windowed += static_cast<unsigned long long>(cur_ne);
        if (cur_ne > RB_CELL_MAX) oversize += 1;
    };
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t c = cnt[st * nblk + b];
        if (c == 0) continue;
        const bool dense = c >= dense_min && !is_long;
        const bool alone = dense || is_long;
        if (started && !alone && !prev_alone && run + c <= RB_MERGE_MAX) {
            run += c;
            cur_ne += c;
        } else {
            close_cell();
            ++local;
            run = c;
            cur_ne = c;
            cur_col0 = dense ? static_cast<int32_t>(b * RB_CWIN) : RB_NO_WINDOW;
        }
        prev_alone = alone;
        started = true;
        loc[st * nblk + b] = static_cast<int32_t>(base + local - 1);
    }
    close_cell();
    if (!pass) ncell[st] = local;
    else {
        if (windowed) atomicAdd(&totals[0], windowed);
        if (oversize) atomicAdd(&totals[1], oversize);
    }
}

// sort key of entry e: (cell << 11) | local row; payload e
__global__ __launch_bounds__(SX_WG) void k_rb_keys(int64_t nnz, const int32_t *__restrict__ ebin,
                                                   const uint16_t *__restrict__ elrow,
                                                   const int32_t *__restrict__ cellid, uint64_t *__restrict__ key,
                                                   int32_t *__restrict__ payload) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < nnz;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        key[e] = (static_cast<uint64_t>(static_cast<uint32_t>(cellid[ebin[e]])) << 11) | elrow[e];
        payload[e] = static_cast<int32_t>(e);
    }
}

// sorted position p -> slot in the layout; counts of (cell, row) for the row starts
__global__ __launch_bounds__(SX_WG) void k_rb_scatter(int64_t nnz, const uint64_t *__restrict__ key,
                                                      const int32_t *__restrict__ payload,
                                                      const int64_t *__restrict__ cell_e0,
                                                      const int64_t *__restrict__ cell_first,
                                                      const int32_t *__restrict__ col, const double *__restrict__ val,
                                                      int32_t *__restrict__ idx_out, double *__restrict__ val_out,
                                                      uint32_t *__restrict__ rowhist) {
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; p < nnz;
         p += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const uint64_t k = key[p];
        const int64_t cell = static_cast<int64_t>(k >> 11);
        const int lrow = static_cast<int>(k & 2047);
        const int64_t pos = cell_e0[cell] + (p - cell_first[cell]);
        const int32_t src = payload[p];
        idx_out[pos] = col[src];
        val_out[pos] = val[src];
        atomicAdd(&rowhist[cell * RB_RS_STRIDE + lrow + 1], 1u);
    }
}

// per cell (one wave each): gap entries, row starts = running sum of the (cell, row) counts
__global__ __launch_bounds__(SX_WG) void k_rb_rowstart(int64_t ncells, const uint32_t *__restrict__ rowhist,
                                                       const int64_t *__restrict__ cell_e0,
                                                       const int64_t *__restrict__ cell_ne,
                                                       const int64_t *__restrict__ cell_pad,
                                                       const int32_t *__restrict__ cell_col0,
                                                       uint16_t *__restrict__ rowstart, int32_t *__restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x) >> 6;
    if (cell >= ncells) return;
    unsigned carry = 0;
    for (int k0 = 0; k0 < RB_RS_STRIDE; k0 += 64) {
        const int k = k0 + lane;
        unsigned v = (k < RB_RS_STRIDE) ? rowhist[cell * RB_RS_STRIDE + k] : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        v += carry;
        if (k < RB_RS_STRIDE) rowstart[cell * RB_RS_STRIDE + k] = static_cast<uint16_t>(v);
        carry = __shfl(v, 63, 64);
    }
    const int64_t gap = cell_pad[cell] - cell_ne[cell];
    if (lane < gap) {
        const int32_t c0 = cell_col0[cell];
        idx_out[cell_e0[cell] + cell_ne[cell] + lane] = (c0 != RB_NO_WINDOW) ? c0 : 0; // val stays 0.0
    }
}

__global__ __launch_bounds__(SX_WG) void k_rb_chunks(int64_t ncells, const int64_t *__restrict__ cell_e0,
                                                     const int64_t *__restrict__ cell_ne,
                                                     const int32_t *__restrict__ cell_col0,
                                                     const int64_t *__restrict__ ch_first,
                                                     sx_rb_chunk *__restrict__ chunks) {
    const int64_t cell = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (cell >= ncells) return;
    const int64_t ne = cell_ne[cell], e0 = cell_e0[cell];
    int64_t k = ch_first[cell];
    for (int64_t base = 0; base < ne; base += RB_CHUNK, ++k) {
        sx_rb_chunk c;
        c.e0 = e0 + base;
        c.ne = static_cast<int32_t>(ne - base < RB_CHUNK ? ne - base : RB_CHUNK);
        c.col0 = cell_col0[cell];
        c.cell = static_cast<int32_t>(cell);
        c.base = static_cast<int32_t>(base);
        c.fresh = base == 0 ? 1 : 0;
        c.staged = 0;
        chunks[k] = c;
    }
}

__global__ __launch_bounds__(SX_WG) void k_rb_supertiles(int64_t nst, const int64_t *__restrict__ cuts,
                                                         const int64_t *__restrict__ cell_base,
                                                         const int64_t *__restrict__ ch_first,
                                                         sx_rb_supertile *__restrict__ st) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (s >= nst) return;
    sx_rb_supertile r;
    r.row0 = cuts[s];
    r.nrows = static_cast<int32_t>(cuts[s + 1] - cuts[s]);
    r.chunk0 = ch_first[cell_base[s]];
    r.nchunks = static_cast<int32_t>(ch_first[cell_base[s + 1]] - r.chunk0);
    st[s] = r;
}

// ---- long rows: mark their chunks, list their entries (column, slot), later sorted by column
__global__ __launch_bounds__(SX_WG) void k_rb_mark_long(int64_t nst, const sx_rb_supertile *__restrict__ st,
                                                        const int32_t *__restrict__ slice, sx_rb_chunk *__restrict__ chunks,
                                                        unsigned long long *__restrict__ count) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (s >= nst || slice[s] == 0) return;
    unsigned long long tot = 0;
    for (int64_t k = st[s].chunk0; k < st[s].chunk0 + st[s].nchunks; ++k) {
        chunks[k].staged = 1;
        tot += static_cast<unsigned long long>(chunks[k].ne);
    }
    atomicAdd(count, tot);
}
// one workgroup per chunk; the order of the list does not matter (it is sorted by column afterwards, and equal
// columns scatter to different slots)
__global__ __launch_bounds__(SX_WG) void k_rb_long_fill(int64_t nchunks, const sx_rb_chunk *__restrict__ chunks,
                                                        const int32_t *__restrict__ idx, unsigned long long *__restrict__ cursor,
                                                        uint64_t *__restrict__ key, int32_t *__restrict__ pay) {
    __shared__ unsigned long long base;
    const int64_t k = blockIdx.x;
    if (k >= nchunks) return;
    const sx_rb_chunk c = chunks[k];
    if (!c.staged) return;
    if (threadIdx.x == 0) base = atomicAdd(cursor, static_cast<unsigned long long>(c.ne));
    __syncthreads();
    for (int t = threadIdx.x; t < c.ne; t += SX_WG) {
        key[base + t] = static_cast<uint64_t>(static_cast<uint32_t>(idx[c.e0 + t]));
        pay[base + t] = static_cast<int32_t>(c.e0 + t);
    }
}
__global__ __launch_bounds__(SX_WG) void k_rb_long_lists(int64_t nl, const uint64_t *__restrict__ key, const int32_t *__restrict__ pay,
                                                         const double *__restrict__ val, int32_t *__restrict__ lcol,
                                                         int32_t *__restrict__ le, double *__restrict__ lval) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (i >= nl) return;
    lcol[i] = static_cast<int32_t>(key[i]);
    le[i] = pay[i];
    lval[i] = val[pay[i]];
}

// temporaries of one build: freed on every exit path
struct Temps {
    std::vector<void *> p;
    ~Temps() {
        for (void *q : p)
            if (q) (void)sx_dfree(q);
    }
    template <class T>
    int get(size_t count, T **out, bool zero, hipStream_t s) {
        void *d = nullptr;
        const size_t bytes = sizeof(T) * (count ? count : 1);
        SX_HIP(sx_dmalloc(&d, bytes));
        p.push_back(d);
        if (zero) SX_HIP(hipMemsetAsync(d, 0, bytes, s));
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

// row cuts of the super-tiles (tools/rb_layout.py::supertile_cuts)
void make_cuts(const std::vector<int64_t> &rowptr, std::vector<int64_t> &cuts, int64_t long_rows) {
    const int64_t m = static_cast<int64_t>(rowptr.size()) - 1;
    cuts.assign(1, 0);
    auto is_long = [&](int64_t r) { return rowptr[r + 1] - rowptr[r] > RB_LONG_ROW; };
    int64_t a = 0;
    while (a < m) {
        const bool lg = is_long(a);
        int64_t b = a + 1;
        while (b < m && is_long(b) == lg) ++b;
        const int64_t lim_rows = lg ? std::min<int64_t>(RB_R, long_rows) : RB_R;
        const int64_t lim_entries = lg ? RB_LONG_BUDGET : RB_BUDGET;
        int64_t row = a;
        while (row < b) {
            const int64_t r_end = std::min(row + lim_rows, b);
            const int64_t r_b = (std::upper_bound(rowptr.begin(), rowptr.end(), rowptr[row] + lim_entries) - rowptr.begin()) - 1;
            const int64_t nxt = std::max(row + 1, std::min(r_end, r_b));
            cuts.push_back(nxt);
            row = nxt;
        }
        a = b;
    }
}

int build(sx_ctx *ctx, const sx_matrix *A, bool force, sx_rowblock **out) {
    *out = nullptr;
    const int64_t m = A->m, n = A->n, nnz = A->nnz;
    if (m == 0 || nnz == 0 || nnz >= INT32_MAX) return SX_OK;
    hipStream_t s = ctx->stream;
    std::vector<int64_t> rowptr(static_cast<size_t>(m) + 1);
    SX_HIP(hipMemcpyAsync(rowptr.data(), A->csr_ptr, sizeof(int64_t) * rowptr.size(), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> cuts;
    make_cuts(rowptr, cuts, ctx->opt_rb_long_rows);
    const int64_t nst = static_cast<int64_t>(cuts.size()) - 1;
    std::vector<int32_t> slice(static_cast<size_t>(nst), 0);
    int64_t nblk = (n + RB_CWIN - 1) / RB_CWIN;
    for (int64_t k = 0; k < nst; ++k) {
        const int64_t r0 = cuts[static_cast<size_t>(k)], r1 = cuts[static_cast<size_t>(k) + 1];
        if (rowptr[r0 + 1] - rowptr[r0] > RB_LONG_ROW) {
            const int64_t sl = std::max<int64_t>(1, RB_CHUNK / (r1 - r0));
            slice[static_cast<size_t>(k)] = static_cast<int32_t>(sl);
            int64_t longest = 0;
            for (int64_t r = r0; r < r1; ++r) longest = std::max(longest, rowptr[r + 1] - rowptr[r]);
            nblk = std::max(nblk, (longest + sl - 1) / sl);
        }
    }
    if (nst * nblk >= (static_cast<int64_t>(1) << 28)) return SX_OK; // bin table too large: plain walk

    Temps tmp;
    int64_t *d_cuts, *d_ncell, *d_cell_base;
    int32_t *d_slice, *d_cnt, *d_loc, *d_ebin;
    uint16_t *d_elrow;
    int *d_desc;
    unsigned long long *d_totals;
    SX_TRY(tmp.get(static_cast<size_t>(nst) + 1, &d_cuts, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(nst), &d_slice, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(nst * nblk), &d_cnt, true, s));
    SX_TRY(tmp.get(static_cast<size_t>(nst * nblk), &d_loc, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(nnz), &d_ebin, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(nnz), &d_elrow, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(nst) + 1, &d_ncell, true, s));
    SX_TRY(tmp.get(static_cast<size_t>(nst) + 1, &d_cell_base, false, s));
    SX_TRY(tmp.get(1, &d_desc, true, s));
    SX_TRY(tmp.get(2, &d_totals, true, s));
    SX_HIP(hipMemcpyAsync(d_cuts, cuts.data(), sizeof(int64_t) * cuts.size(), hipMemcpyHostToDevice, s));
    SX_HIP(hipMemcpyAsync(d_slice, slice.data(), sizeof(int32_t) * slice.size(), hipMemcpyHostToDevice, s));

    hipLaunchKernelGGL(k_rb_bins, dim3(static_cast<unsigned>(A->n_csr_tiles)), dim3(SX_WG), 0, s, A->csr_tiles,
                       A->n_csr_tiles, A->csr_ptr, A->csr_idx, d_cuts, nst, d_slice, nblk, d_cnt, d_ebin, d_elrow, d_desc);
    hipLaunchKernelGGL(k_rb_cells, dim3(grid1d(nst)), dim3(SX_WG), 0, s, 0, ctx->opt_rb_dense_min, nst, nblk, d_cnt, d_slice,
                       static_cast<const int64_t *>(nullptr), d_loc, d_ncell, static_cast<int64_t *>(nullptr),
                       static_cast<int32_t *>(nullptr), static_cast<int64_t *>(nullptr), static_cast<int64_t *>(nullptr),
                       d_totals);
    SX_TRY(sx_scan_exclusive(ctx, d_ncell, nst, d_cell_base));
    int64_t ncells = 0;
    int descending = 0;
    SX_HIP(hipMemcpyAsync(&ncells, d_cell_base + nst, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(&descending, d_desc, sizeof(int), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (descending || ncells == 0 || ncells >= (static_cast<int64_t>(1) << 26)) return SX_OK;

    int64_t *d_cell_ne, *d_cell_pad, *d_cell_nch, *d_cell_e0, *d_cell_first, *d_ch_first;
    int32_t *d_cell_col0;
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_cell_ne, true, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_cell_pad, true, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_cell_nch, true, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_cell_e0, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_cell_first, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells) + 1, &d_ch_first, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(ncells), &d_cell_col0, false, s));
    hipLaunchKernelGGL(k_rb_cells, dim3(grid1d(nst)), dim3(SX_WG), 0, s, 1, ctx->opt_rb_dense_min, nst, nblk, d_cnt, d_slice, d_cell_base, d_loc,
                       d_ncell, d_cell_ne, d_cell_col0, d_cell_pad, d_cell_nch, d_totals);
    SX_TRY(sx_scan_exclusive(ctx, d_cell_pad, ncells, d_cell_e0));
    SX_TRY(sx_scan_exclusive(ctx, d_cell_ne, ncells, d_cell_first));
    SX_TRY(sx_scan_exclusive(ctx, d_cell_nch, ncells, d_ch_first));
    unsigned long long totals[2] = {0, 0};
    int64_t nent = 0, nchunks = 0;
    SX_HIP(hipMemcpyAsync(totals, d_totals, sizeof(totals), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(&nent, d_cell_e0 + ncells, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(&nchunks, d_ch_first + ncells, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (totals[1] != 0) return SX_OK;                                       // a cell beyond 65535 entries
    if (!force && 2 * static_cast<int64_t>(totals[0]) < nnz) return SX_OK; // windows would not pay
    if (nchunks >= INT32_MAX) return SX_OK;

    // ---- sort the entries by (cell, local row); stable, so stored order survives inside a row
    uint64_t *key[2];
    int32_t *pay[2];
    int64_t *hist, *offs;
    const int64_t nblocks = sx_sort_blocks(nnz);
    for (int k = 0; k < 2; ++k) {
        SX_TRY(tmp.get(static_cast<size_t>(nnz), &key[k], false, s));
        SX_TRY(tmp.get(static_cast<size_t>(nnz), &pay[k], false, s));
    }
    SX_TRY(tmp.get(static_cast<size_t>(256 * nblocks + 1), &hist, false, s));
    SX_TRY(tmp.get(static_cast<size_t>(256 * nblocks + 1), &offs, false, s));
    hipLaunchKernelGGL(k_rb_keys, dim3(grid1d(nnz, 8192)), dim3(SX_WG), 0, s, nnz, d_ebin, d_elrow, d_loc, key[0], pay[0]);
    int bits = 11;
    while ((static_cast<int64_t>(1) << (bits - 11)) < ncells) ++bits;
    int cur = 0;
    SX_TRY(sx_sort_pairs(ctx, nnz, key, pay, hist, offs, (bits + 7) / 8, &cur));

    // ---- the layout itself
    sx_rowblock *rb = new (std::nothrow) sx_rowblock();
    if (!rb) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    struct Guard {
        sx_rowblock *rb;
        ~Guard() {
            if (rb) sx_rowblock_free(rb);
        }
    } guard{rb};
    rb->nst = nst;
    rb->ncells = ncells;
    rb->nchunks = nchunks;
    rb->nent = nent;
    rb->nnz = nnz;
    rb->windowed = static_cast<int64_t>(totals[0]);
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->idx), sizeof(int32_t) * static_cast<size_t>(nent + 8)));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->val), sizeof(double) * static_cast<size_t>(nent + 8)));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->rowstart), sizeof(uint16_t) * static_cast<size_t>(ncells) * RB_RS_STRIDE));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->chunks), sizeof(sx_rb_chunk) * static_cast<size_t>(nchunks)));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->st), sizeof(sx_rb_supertile) * static_cast<size_t>(nst)));
    SX_HIP(hipMemsetAsync(rb->idx, 0, sizeof(int32_t) * static_cast<size_t>(nent + 8), s));
    SX_HIP(hipMemsetAsync(rb->val, 0, sizeof(double) * static_cast<size_t>(nent + 8), s));
    uint32_t *rowhist;
    SX_TRY(tmp.get(static_cast<size_t>(ncells) * RB_RS_STRIDE, &rowhist, true, s));
    hipLaunchKernelGGL(k_rb_scatter, dim3(grid1d(nnz, 8192)), dim3(SX_WG), 0, s, nnz, key[cur], pay[cur], d_cell_e0,
                       d_cell_first, A->csr_idx, A->csr_val, rb->idx, rb->val, rowhist);
    hipLaunchKernelGGL(k_rb_rowstart, dim3(grid1d(ncells * 64)), dim3(SX_WG), 0, s, ncells, rowhist, d_cell_e0, d_cell_ne,
                       d_cell_pad, d_cell_col0, rb->rowstart, rb->idx);
    hipLaunchKernelGGL(k_rb_chunks, dim3(grid1d(ncells)), dim3(SX_WG), 0, s, ncells, d_cell_e0, d_cell_ne, d_cell_col0,
                       d_ch_first, rb->chunks);
    hipLaunchKernelGGL(k_rb_supertiles, dim3(grid1d(nst)), dim3(SX_WG), 0, s, nst, d_cuts, d_cell_base, d_ch_first, rb->st);
    SX_HIP(hipGetLastError());
    // ---- long rows: their entries once more, sorted by column, for the product pre-pass (sx_rowblock.h)
    if (ctx->opt_rb_stage_long) {
        unsigned long long *d_cnt2;
        SX_TRY(tmp.get(2, &d_cnt2, true, s));
        hipLaunchKernelGGL(k_rb_mark_long, dim3(grid1d(nst)), dim3(SX_WG), 0, s, nst, rb->st, d_slice, rb->chunks, d_cnt2);
        unsigned long long nl = 0;
        SX_HIP(hipMemcpyAsync(&nl, d_cnt2, sizeof(nl), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        if (nl > 0 && nl < INT32_MAX && nent + 8 < INT32_MAX) {
            uint64_t *lkey[2];
            int32_t *lpay[2];
            int64_t *lhist, *loffs;
            const int64_t lblocks = sx_sort_blocks(static_cast<int64_t>(nl));
            for (int k = 0; k < 2; ++k) {
                SX_TRY(tmp.get(static_cast<size_t>(nl), &lkey[k], false, s));
                SX_TRY(tmp.get(static_cast<size_t>(nl), &lpay[k], false, s));
            }
            SX_TRY(tmp.get(static_cast<size_t>(256 * lblocks + 1), &lhist, false, s));
            SX_TRY(tmp.get(static_cast<size_t>(256 * lblocks + 1), &loffs, false, s));
            hipLaunchKernelGGL(k_rb_long_fill, dim3(static_cast<unsigned>(nchunks)), dim3(SX_WG), 0, s, nchunks, rb->chunks, rb->idx,
                               d_cnt2 + 1, lkey[0], lpay[0]);
            int cbits = 1;
            while ((static_cast<int64_t>(1) << cbits) < n) ++cbits;
            int lcur = 0;
            SX_TRY(sx_sort_pairs(ctx, static_cast<int64_t>(nl), lkey, lpay, lhist, loffs, (cbits + 7) / 8, &lcur));
            rb->nl = static_cast<int64_t>(nl);
            SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->lcol), sizeof(int32_t) * static_cast<size_t>(nl)));
            SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->le), sizeof(int32_t) * static_cast<size_t>(nl)));
            SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->lval), sizeof(double) * static_cast<size_t>(nl)));
            SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->lprod), sizeof(double) * static_cast<size_t>(nent + 8)));
            SX_HIP(hipMemsetAsync(rb->lprod, 0, sizeof(double) * static_cast<size_t>(nent + 8), s));
            hipLaunchKernelGGL(k_rb_long_lists, dim3(grid1d(static_cast<int64_t>(nl))), dim3(SX_WG), 0, s, static_cast<int64_t>(nl),
                               lkey[lcur], lpay[lcur], rb->val, rb->lcol, rb->le, rb->lval);
            SX_HIP(hipGetLastError());
        } else if (nl > 0) { // too large for 32-bit slots: the chunks go back to the gather
            SX_HIP(hipMemsetAsync(d_cnt2, 0, 2 * sizeof(unsigned long long), s));
            rb->nl = -1;
        }
    }
    SX_HIP(hipStreamSynchronize(s)); // the temporaries go away when this function returns
    guard.rb = nullptr;
    *out = rb;
    return SX_OK;
}

} // namespace

void sx_rowblock_free(sx_rowblock *rb) {
    if (!rb) return;
    void *ptrs[10] = {rb->st, rb->chunks, rb->rowstart, rb->idx, rb->val, rb->lcol, rb->le, rb->lval, rb->lprod, rb->order};
    for (void *p : ptrs)
        if (p) (void)sx_dfree(p);
    delete rb;
}

int sx_rowblock_get(sx_ctx *ctx, const sx_matrix *A, const sx_rowblock **out) {
    *out = nullptr;
    const int opt = ctx->opt_rowblock;
    if (opt == 0 || A->csr_ptr == nullptr || A->csr_tiles == nullptr) return SX_OK;
    const bool force = opt > 0;
    if (!force && (A->nnz < RB_AUTO_NNZ || A->m < RB_AUTO_ROWS)) return SX_OK;
    // the verdict of a forced build covers the automatic one, not the other way round
    if (!A->rb && A->rb_tried < (force ? 2 : 1)) {
        A->rb_tried = force ? 2 : 1;
        sx_rowblock *rb = nullptr;
        SX_TRY(build(ctx, A, force, &rb));
        A->rb = rb;
    }
    *out = A->rb;
    return SX_OK;
}

// ------------------------------------------------------------------ introspection (tests, tools)
SX_API int sx_matrix_rowblock_info(sx_ctx *ctx, const sx_matrix *A, int64_t *info /* [6] */) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr && info != nullptr, "NULL argument");
    const sx_rowblock *rb = nullptr;
    SX_TRY(sx_rowblock_get(ctx, A, &rb));
    for (int k = 0; k < 6; ++k) info[k] = 0;
    if (rb) {
        info[0] = rb->nst;
        info[1] = rb->ncells;
        info[2] = rb->nchunks;
        info[3] = rb->nent;
        info[4] = rb->windowed;
        info[5] = RB_RS_STRIDE;
    }
    return SX_OK;
}

SX_API int sx_matrix_rowblock_download(sx_ctx *ctx, const sx_matrix *A, void *st, void *chunks, uint16_t *rowstart,
                                       int32_t *idx, double *val) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    const sx_rowblock *rb = nullptr;
    SX_TRY(sx_rowblock_get(ctx, A, &rb));
    SX_REQUIRE(rb != nullptr, "the matrix has no row-block layout");
    hipStream_t s = ctx->stream;
    if (st) SX_HIP(hipMemcpyAsync(st, rb->st, sizeof(sx_rb_supertile) * static_cast<size_t>(rb->nst), hipMemcpyDeviceToHost, s));
    if (chunks)
        SX_HIP(hipMemcpyAsync(chunks, rb->chunks, sizeof(sx_rb_chunk) * static_cast<size_t>(rb->nchunks), hipMemcpyDeviceToHost, s));
    if (rowstart)
        SX_HIP(hipMemcpyAsync(rowstart, rb->rowstart, sizeof(uint16_t) * static_cast<size_t>(rb->ncells) * RB_RS_STRIDE,
                              hipMemcpyDeviceToHost, s));
    if (idx) SX_HIP(hipMemcpyAsync(idx, rb->idx, sizeof(int32_t) * static_cast<size_t>(rb->nent + 8), hipMemcpyDeviceToHost, s));
    if (val) SX_HIP(hipMemcpyAsync(val, rb->val, sizeof(double) * static_cast<size_t>(rb->nent + 8), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    return SX_OK;
}
