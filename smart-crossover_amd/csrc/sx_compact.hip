// K6 / K12: restricted sub-problem on the device (reference: LPManager.fix_variables /
// update_subproblem, lp_methods/lp_manager.py:40-66; MCFManagerStd.update_subproblem,
// network_methods/net_manager.py:202-209):
//   non_fix = columns whose code is 0, ascending             (np.setdiff1d)
//   A_sub   = A[:, non_fix]  in CSR and CSC, entry order preserved
//   b_sub   = b - A[:, fix_up] @ u[fix_up] - A[:, fix_low] @ l[fix_low]   (bit-exact, see below)
//   c/l/u   = gathers
// b_sub keeps the reference's rounding: each of the two products-sums is a per-row sequential sum
// over the selected entries in stored order.  The kernel walks *all* entries of a row and stages
// +0.0 for unselected ones: a running sum that starts at +0.0 is never -0.0, so adding +0.0 is an
// exact no-op and the result equals the sum over the selected entries alone.
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

// ------------------------------------------------------------------ exclusive scan (int64)
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_TILE = SX_WG * SCAN_PER_THREAD;

__device__ __forceinline__ long long block_exclusive(long long mine, long long &total) {
    // exclusive prefix of `mine` over the workgroup, thread order; total = sum
    long long incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __shared__ long long wsum[SX_WG / 64];
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return woff + incl - mine;
}

__global__ __launch_bounds__(SX_WG) void k_scan_block_sums(const int64_t *__restrict__ in, int64_t n,
                                                           int64_t *__restrict__ block_sum) {
    const int64_t first = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    long long s = 0;
    for (int t = 0; t < SCAN_PER_THREAD; ++t)
        if (first + t < n) s += in[first + t];
    long long total;
    (void)block_exclusive(s, total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(SX_WG) void k_scan_sums(int64_t *__restrict__ block_sum, int64_t nblocks,
                                                     int64_t *__restrict__ total_out) {
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += SX_WG) {
        const int64_t b = b0 + threadIdx.x;
        const long long mine = (b < nblocks) ? block_sum[b] : 0;
        long long total;
        const long long ex = block_exclusive(mine, total);
        if (b < nblocks) block_sum[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// out[i] = exclusive prefix; out has n+1 entries (out[n] = total)
__global__ __launch_bounds__(SX_WG) void k_scan_write(const int64_t *__restrict__ in, int64_t n,
                                                      const int64_t *__restrict__ block_off,
                                                      int64_t *__restrict__ out) {
    const int64_t first = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    long long v[SCAN_PER_THREAD];
    long long s = 0;
    for (int t = 0; t < SCAN_PER_THREAD; ++t) {
        v[t] = (first + t < n) ? in[first + t] : 0;
        s += v[t];
    }
    long long total;
    long long run = block_off[blockIdx.x] + block_exclusive(s, total);
    for (int t = 0; t < SCAN_PER_THREAD; ++t) {
        if (first + t < n) out[first + t] = run;
        run += v[t];
    }
    if (first <= n - 1 && n - 1 < first + SCAN_PER_THREAD) out[n] = run; // thread owning the last element
}

} // namespace

// exclusive scan of in[0..n) into out[0..n] (out[n] = total); uses ctx->ws
int sx_scan_exclusive(sx_ctx *ctx, const int64_t *in, int64_t n, int64_t *out) {
    if (n == 0) {
        SX_HIP(hipMemsetAsync(out, 0, sizeof(int64_t), ctx->stream));
        return SX_OK;
    }
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    SX_TRY(sx_reserve(ctx, sizeof(int64_t) * static_cast<size_t>(nb)));
    int64_t *bs = static_cast<int64_t *>(ctx->ws);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream, in, n, bs);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SX_WG), 0, ctx->stream, bs, nb,
                       static_cast<int64_t *>(nullptr));
    hipLaunchKernelGGL(k_scan_write, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream, in, n, bs, out);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

namespace {

// ------------------------------------------------------------------ compaction kernels
__global__ __launch_bounds__(SX_WG) void k_keep_flags(int64_t n, const uint8_t *__restrict__ code,
                                                      uint8_t *__restrict__ keep) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG)
        keep[j] = code[j] ? 0 : 1;
}

__global__ __launch_bounds__(SX_WG) void k_fill_i32(int64_t n, int32_t v, int32_t *__restrict__ out) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG)
        out[j] = v;
}

// colmap[non_fix[k]] = k ; len[k] = length of column non_fix[k]
__global__ __launch_bounds__(SX_WG) void k_colmap(int64_t nsub, const int64_t *__restrict__ non_fix,
                                                  const int64_t *__restrict__ colptr,
                                                  int32_t *__restrict__ colmap, int64_t *__restrict__ len) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < nsub;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t j = non_fix[k];
        colmap[j] = static_cast<int32_t>(k);
        len[k] = colptr[j + 1] - colptr[j];
    }
}

// CSC copy: kept column k <- column non_fix[k]; one lane per column (columns are short)
__global__ __launch_bounds__(SX_WG) void k_copy_columns(int64_t nsub, const int64_t *__restrict__ non_fix,
                                                        const int64_t *__restrict__ colptr,
                                                        const int32_t *__restrict__ rowidx,
                                                        const double *__restrict__ val,
                                                        const int64_t *__restrict__ colptr_sub,
                                                        int32_t *__restrict__ rowidx_sub,
                                                        double *__restrict__ val_sub) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < nsub;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t j = non_fix[k];
        int64_t src = colptr[j], end = colptr[j + 1], dst = colptr_sub[k];
        for (; src < end; ++src, ++dst) {
            rowidx_sub[dst] = rowidx[src];
            val_sub[dst] = val[src];
        }
    }
}

// CSR: kept-entry count per row (a wave per row so that long rows are read coalesced)
__global__ __launch_bounds__(SX_WG) void k_row_keep_count(int64_t m, const int64_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ colidx,
                                                          const int32_t *__restrict__ colmap,
                                                          int64_t *__restrict__ len) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * SX_WG) >> 6;
    for (int64_t i = wave; i < m; i += nwaves) {
        long long cnt = 0;
        for (int64_t e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) cnt += (colmap[colidx[e]] >= 0) ? 1 : 0;
        cnt = sx_wave_sum(cnt);
        if (lane == 0) len[i] = cnt;
    }
}

// CSR write: a wave per row, order-preserving (ballot ranks inside each 64-entry step)
__global__ __launch_bounds__(SX_WG) void k_row_write(int64_t m, const int64_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ colidx,
                                                     const double *__restrict__ val,
                                                     const int32_t *__restrict__ colmap,
                                                     const int64_t *__restrict__ rowptr_sub,
                                                     int32_t *__restrict__ colidx_sub,
                                                     double *__restrict__ val_sub) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * SX_WG) >> 6;
    for (int64_t i = wave; i < m; i += nwaves) {
        int64_t dst = rowptr_sub[i];
        const int64_t end = rowptr[i + 1];
        for (int64_t e0 = rowptr[i]; e0 < end; e0 += 64) {
            const int64_t e = e0 + lane;
            int32_t nc = -1;
            double v = 0.0;
            if (e < end) {
                nc = colmap[colidx[e]];
                v = val[e];
            }
            const unsigned long long mask = __ballot(nc >= 0);
            if (nc >= 0) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                colidx_sub[dst + rank] = nc;
                val_sub[dst + rank] = v;
            }
            dst += __popcll(mask);
        }
    }
}

// ------------------------------------------------------------------ fixed-column right-hand side
struct StageFixed {
    const uint8_t *__restrict__ code;
    const double *__restrict__ u;
    const double *__restrict__ l;
    __device__ __forceinline__ void operator()(double v, int32_t j, double (&o)[2]) const {
        const uint8_t cd = code[j];
        o[0] = (cd & SX_CODE_UP) ? v * u[j] : 0.0;
        o[1] = (cd & SX_CODE_LOW) ? v * l[j] : 0.0;
    }
};

__global__ __launch_bounds__(SX_WG) void k_fixed_rhs(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                     int swizzle, const int64_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ colidx,
                                                     const double *__restrict__ val,
                                                     const uint8_t *__restrict__ code,
                                                     const double *__restrict__ u,
                                                     const double *__restrict__ l,
                                                     const double *__restrict__ b,
                                                     double *__restrict__ b_sub) {
    __shared__ sx_walk_lds<2, 2048> lds;
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    double acc[2];
    int64_t i;
    bool valid;
    sx_segwalk<2, 2048>(tiles, tile, rowptr, colidx, val, StageFixed{code, u, l}, lds, i, valid, acc);
    if (valid) b_sub[i] = (b[i] - acc[0]) - acc[1];
}

__global__ __launch_bounds__(SX_WG) void k_gather_f64(int64_t n, const int64_t *__restrict__ idx,
                                                      const double *__restrict__ src,
                                                      double *__restrict__ dst) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < n;
         k += static_cast<int64_t>(gridDim.x) * SX_WG)
        dst[k] = src[idx[k]];
}

inline unsigned grid1d(int64_t n, int64_t cap = 8192) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

template <class T>
int dev_alloc_padded(sx_ctx *ctx, int64_t count, T **out) {
    *out = nullptr;
    T *d = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d), sizeof(T) * static_cast<size_t>(count + SX_PAD)));
    *out = d;
    SX_HIP(hipMemsetAsync(d + count, 0, sizeof(T) * SX_PAD, ctx->stream));
    return SX_OK;
}

struct scratch {
    std::vector<void *> p;
    ~scratch() {
        for (void *q : p)
            if (q) (void)sx_dfree(q);
    }
    template <class T>
    int get(size_t count, T **out) {
        void *d = nullptr;
        SX_HIP(sx_dmalloc(&d, sizeof(T) * (count ? count : 1)));
        p.push_back(d);
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

} // namespace

SX_API int sx_compact_columns_dev(sx_ctx *ctx, const sx_matrix *A, const uint8_t *code,
                                  sx_matrix **A_sub_out, int64_t *non_fix, int64_t *n_sub_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && code && A_sub_out && non_fix && n_sub_out, "NULL argument");
    *A_sub_out = nullptr;
    *n_sub_out = 0;
    const int64_t n = A->n;
    hipStream_t s = ctx->stream;
    scratch tmp;
    uint8_t *keep;
    int64_t *count_dev;
    SX_TRY(tmp.get(static_cast<size_t>(n) + 16, &keep));
    SX_TRY(tmp.get(1, &count_dev));
    int64_t nsub = 0;
    if (n > 0) {
        hipLaunchKernelGGL(k_keep_flags, dim3(grid1d(n)), dim3(SX_WG), 0, s, n, code, keep);
        SX_TRY(sx_select_indices_dev(ctx, n, keep, 1, non_fix, count_dev));
        SX_HIP(hipMemcpyAsync(&nsub, count_dev, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
    }
    SX_TRY(sx_gather_columns_dev(ctx, A, non_fix, nsub, A_sub_out));
    *n_sub_out = nsub;
    return SX_OK;
}

SX_API int sx_gather_columns_dev(sx_ctx *ctx, const sx_matrix *A, const int64_t *non_fix, int64_t nsub,
                                 sx_matrix **A_sub_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && A_sub_out, "NULL argument");
    SX_REQUIRE(nsub >= 0 && (nsub == 0 || non_fix), "bad column list");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "column gathering needs both layouts of A");
    *A_sub_out = nullptr;
    const int64_t m = A->m, n = A->n;
    hipStream_t s = ctx->stream;
    scratch tmp;
    int64_t *len_c, *len_r;
    int32_t *colmap;
    SX_TRY(tmp.get(static_cast<size_t>(n), &colmap));
    SX_TRY(tmp.get(static_cast<size_t>(m), &len_r));
    SX_TRY(tmp.get(static_cast<size_t>(nsub), &len_c));

    sx_matrix *S = new (std::nothrow) sx_matrix();
    if (!S) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    S->ctx = ctx;
    S->m = m;
    S->n = nsub;
    int rc = SX_OK;
    do {
        if ((rc = dev_alloc_padded(ctx, nsub + 1, &S->csc_ptr)) != SX_OK) break;
        if ((rc = dev_alloc_padded(ctx, m + 1, &S->csr_ptr)) != SX_OK) break;
        if (n > 0) hipLaunchKernelGGL(k_fill_i32, dim3(grid1d(n)), dim3(SX_WG), 0, s, n, -1, colmap);
        if (nsub > 0)
            hipLaunchKernelGGL(k_colmap, dim3(grid1d(nsub)), dim3(SX_WG), 0, s, nsub, non_fix, A->csc_ptr, colmap, len_c);
        if ((rc = sx_scan_exclusive(ctx, len_c, nsub, S->csc_ptr)) != SX_OK) break;
        if (m > 0)
            hipLaunchKernelGGL(k_row_keep_count, dim3(grid1d(m * 64)), dim3(SX_WG), 0, s, m, A->csr_ptr, A->csr_idx,
                               colmap, len_r);
        if ((rc = sx_scan_exclusive(ctx, len_r, m, S->csr_ptr)) != SX_OK) break;
        int64_t nnz_c = 0, nnz_r = 0;
        if (hipMemcpyAsync(&nnz_c, S->csc_ptr + nsub, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&nnz_r, S->csr_ptr + m, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            sx_set_error("nnz download failed in compaction");
            rc = SX_ERR_HIP;
            break;
        }
        if (nnz_c != nnz_r) {
            sx_set_error("internal error: CSC (%lld) and CSR (%lld) sub-matrix sizes differ", (long long)nnz_c,
                         (long long)nnz_r);
            rc = SX_ERR_HIP;
            break;
        }
        S->nnz = nnz_c;
        if ((rc = dev_alloc_padded(ctx, nnz_c, &S->csc_idx)) != SX_OK) break;
        if ((rc = dev_alloc_padded(ctx, nnz_c, &S->csc_val)) != SX_OK) break;
        if ((rc = dev_alloc_padded(ctx, nnz_c, &S->csr_idx)) != SX_OK) break;
        if ((rc = dev_alloc_padded(ctx, nnz_c, &S->csr_val)) != SX_OK) break;
        if (nsub > 0)
            hipLaunchKernelGGL(k_copy_columns, dim3(grid1d(nsub)), dim3(SX_WG), 0, s, nsub, non_fix, A->csc_ptr,
                               A->csc_idx, A->csc_val, S->csc_ptr, S->csc_idx, S->csc_val);
        if (m > 0)
            hipLaunchKernelGGL(k_row_write, dim3(grid1d(m * 64)), dim3(SX_WG), 0, s, m, A->csr_ptr, A->csr_idx,
                               A->csr_val, colmap, S->csr_ptr, S->csr_idx, S->csr_val);
        if (hipGetLastError() != hipSuccess) {
            sx_set_error("compaction kernel launch failed");
            rc = SX_ERR_HIP;
            break;
        }
        if ((rc = sx_build_tiles(ctx, S->csr_ptr, m, &S->csr_tiles, &S->n_csr_tiles, &S->csr_imbalance)) != SX_OK) break;
        if ((rc = sx_build_tiles(ctx, S->csc_ptr, nsub, &S->csc_tiles, &S->n_csc_tiles, &S->csc_imbalance)) != SX_OK) break;
        if (hipStreamSynchronize(s) != hipSuccess) {
            sx_set_error("stream sync failed after compaction");
            rc = SX_ERR_HIP;
        }
    } while (0);
    if (rc != SX_OK) {
        sx_matrix_destroy(S);
        return rc;
    }
    *A_sub_out = S;
    return SX_OK;
}

SX_API int sx_fixed_rhs_dev(sx_ctx *ctx, const sx_matrix *A, const uint8_t *code, const double *u,
                            const double *l, const double *b, double *b_sub) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && code && u && l && b && b_sub, "NULL argument");
    SX_REQUIRE(A->csr_ptr != nullptr, "matrix has no CSR layout");
    if (A->m == 0) return SX_OK;
    const int swz = ctx->opt_xcd_swizzle;
    const unsigned grid = swz ? static_cast<unsigned>(((A->n_csr_tiles + 7) >> 3) << 3)
                              : static_cast<unsigned>(A->n_csr_tiles);
    hipLaunchKernelGGL(k_fixed_rhs, dim3(grid), dim3(SX_WG), 0, ctx->stream, A->csr_tiles, A->n_csr_tiles, swz,
                       A->csr_ptr, A->csr_idx, A->csr_val, code, u, l, b, b_sub);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_gather_f64_dev(sx_ctx *ctx, int64_t n, const int64_t *idx, const double *src, double *dst) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return SX_OK;
    SX_REQUIRE(idx && src && dst, "NULL argument");
    hipLaunchKernelGGL(k_gather_f64, dim3(grid1d(n)), dim3(SX_WG), 0, ctx->stream, n, idx, src, dst);
    SX_HIP(hipGetLastError());
    return SX_OK;
}
