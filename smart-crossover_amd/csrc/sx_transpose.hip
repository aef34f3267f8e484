// CSR -> CSC on the device: the *stable* transposition (inside every column the entries keep their
// row-major walk order), i.e. exactly what scipy's csr_tocsc produces and what the column kernels'
// rounding order is defined on (formats.py:72: A.transpose() @ y).  Used by sx_matrix_create when the
// caller passes no CSC arrays: a host counting sort of an 8e7-entry matrix takes seconds, this takes
// milliseconds.
//
//   counts   per-column entry counts (atomics; the counts, not the order, come from them) -> scan -> colptr
//   order    stable ascending sort of the entry positions by column = the library's ranking kernel
//            (descending key, descending position among ties) on the keys -col, read as is
//   gather   row index (binary search of the position in rowptr) and value of every entry in that order
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

__global__ __launch_bounds__(SX_WG) void k_tr_count(int64_t nnz, const int32_t *__restrict__ col,
                                                    unsigned long long *__restrict__ cnt,
                                                    double *__restrict__ key) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < nnz;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int32_t c = col[e];
        atomicAdd(&cnt[c], 1ull);
        key[e] = -static_cast<double>(c); // descending -col == ascending col; int32 is exact in a double
    }
}

__global__ __launch_bounds__(SX_WG) void k_tr_gather(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr,
                                                     const double *__restrict__ val,
                                                     const int64_t *__restrict__ order,
                                                     int32_t *__restrict__ row_out, double *__restrict__ val_out) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < nnz;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        // the ranking breaks ties by *descending* position, a stable ascending order needs ascending
        // positions: entries of one column are therefore taken from the back of their run -- see below
        const int64_t src = order[k];
        int64_t lo = 0, hi = m; // last row whose first entry is <= src
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (rowptr[mid] <= src) lo = mid;
            else hi = mid;
        }
        row_out[k] = static_cast<int32_t>(lo);
        val_out[k] = val[src];
    }
}

// positions of one column arrive in descending order from the ranking kernel: reverse every run
__global__ __launch_bounds__(SX_WG) void k_tr_fix_order(int64_t n, const int64_t *__restrict__ colptr,
                                                        const int64_t *__restrict__ ranked,
                                                        int64_t *__restrict__ order) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * SX_WG) >> 6;
    for (int64_t j = wave; j < n; j += nwaves) {
        const int64_t lo = colptr[j], hi = colptr[j + 1];
        for (int64_t k = lo + lane; k < hi; k += 64) order[k] = ranked[lo + (hi - 1 - k)];
    }
}

inline unsigned grid1d(int64_t n, int64_t cap = 8192) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

template <class T>
int alloc_padded(int64_t count, T **out, hipStream_t s) {
    void *p = nullptr;
    const size_t bytes = sizeof(T) * static_cast<size_t>(count + SX_PAD);
    SX_HIP(sx_dmalloc(&p, bytes));
    if (hipMemsetAsync(static_cast<char *>(p) + sizeof(T) * static_cast<size_t>(count), 0, sizeof(T) * SX_PAD, s) !=
        hipSuccess) {
        (void)sx_dfree(p);
        sx_set_error("hipMemsetAsync failed");
        return SX_ERR_HIP;
    }
    *out = static_cast<T *>(p);
    return SX_OK;
}

} // namespace

// rowptr / col / val: device CSR arrays of an m x n matrix with nnz entries.  On success the three
// outputs are device arrays (padded like every sparse array of the library) owned by the caller.
int sx_transpose_dev(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int32_t *col,
                     const double *val, int64_t **colptr_out, int32_t **row_out, double **val_out) {
    hipStream_t s = ctx->stream;
    *colptr_out = nullptr;
    *row_out = nullptr;
    *val_out = nullptr;
    int64_t *colptr = nullptr;
    int32_t *rows = nullptr;
    double *vals = nullptr;
    unsigned long long *cnt = nullptr;
    double *key = nullptr;
    int64_t *ranked = nullptr, *order = nullptr;
    int rc = SX_OK;
    do {
        if ((rc = alloc_padded(n + 1, &colptr, s)) != SX_OK) break;
        if ((rc = alloc_padded(nnz, &rows, s)) != SX_OK) break;
        if ((rc = alloc_padded(nnz, &vals, s)) != SX_OK) break;
        if (sx_dmalloc(reinterpret_cast<void **>(&cnt), sizeof(unsigned long long) * static_cast<size_t>(n + 1)) != hipSuccess ||
            sx_dmalloc(reinterpret_cast<void **>(&key), sizeof(double) * static_cast<size_t>(nnz + 1)) != hipSuccess ||
            sx_dmalloc(reinterpret_cast<void **>(&ranked), sizeof(int64_t) * static_cast<size_t>(nnz + 1)) != hipSuccess ||
            sx_dmalloc(reinterpret_cast<void **>(&order), sizeof(int64_t) * static_cast<size_t>(nnz + 1)) != hipSuccess) {
            sx_set_error("out of device memory while transposing (%lld entries)", (long long)nnz);
            rc = SX_ERR_NOMEM;
            break;
        }
        if (hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * static_cast<size_t>(n + 1), s) != hipSuccess) {
            rc = SX_ERR_HIP;
            break;
        }
        if (nnz > 0) hipLaunchKernelGGL(k_tr_count, dim3(grid1d(nnz)), dim3(SX_WG), 0, s, nnz, col, cnt, key);
        if ((rc = sx_scan_exclusive(ctx, reinterpret_cast<const int64_t *>(cnt), n, colptr)) != SX_OK) break;
        if (nnz > 0) {
            if ((rc = sx_argsort_desc_dev(ctx, nnz, key, ranked)) != SX_OK) break;
            hipLaunchKernelGGL(k_tr_fix_order, dim3(grid1d(n * 64)), dim3(SX_WG), 0, s, n, colptr, ranked, order);
            hipLaunchKernelGGL(k_tr_gather, dim3(grid1d(nnz)), dim3(SX_WG), 0, s, m, nnz, rowptr, val, order, rows, vals);
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            sx_set_error("device transposition failed");
            rc = SX_ERR_HIP;
        }
    } while (0);
    if (cnt) (void)sx_dfree(cnt);
    if (key) (void)sx_dfree(key);
    if (ranked) (void)sx_dfree(ranked);
    if (order) (void)sx_dfree(order);
    if (rc != SX_OK) {
        if (colptr) (void)sx_dfree(colptr);
        if (rows) (void)sx_dfree(rows);
        if (vals) (void)sx_dfree(vals);
        return rc;
    }
    *colptr_out = colptr;
    *row_out = rows;
    *val_out = vals;
    return SX_OK;
}
