// Window table of the column walk (see sx_window.h): per tile, the start of the SXL_CAP-long index
// window that covers most of the tile's entries, plus the statistics the "auto" mode decides on.
// Built once per matrix, on the first K1 / K10 call that may use it.
//
// Measured on MI355X, c5 shard (profiles/r01/kbench_window_run.txt), K1 in ms, plain -> windowed run of 4:
//   staircase window   64 rows : 0.305 -> 0.329   (gathers already hit L1: window load is pure overhead)
//   staircase window 4096 rows : 0.424 -> 0.344
//   staircase window  32 Ki rows: 0.477 -> 0.463
// Hence auto mode: windowed only when at least half of the sampled indices fall inside the window and
// the window is actually wide (median extent >= 1024 rows).  Two further designs were built, measured
// slower and removed: persistent 1024-lane workgroups sharing a sliding 64 KiB ring, and a register
// prefetch of tile t+1's entries while tile t is summed (no gain: the walk is not latency-bound).
#include "sx_internal.h"
#include "sx_window.h"

#include <algorithm>

namespace {

constexpr int SXL_THREADS = 1024; // sample size for the densest-window search

// sample up to 1024 evenly spaced entries of the tile, sort them in LDS, then one binary search per
// sample value counts how many samples fall in [value, value + CAP)
__global__ __launch_bounds__(SXL_THREADS) void k_win_lo(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                        const int64_t *__restrict__ ptr,
                                                        const int32_t *__restrict__ idx, int64_t bound,
                                                        int32_t *__restrict__ win_lo, int32_t *__restrict__ stat) {
    __shared__ int32_t s[SXL_THREADS];
    __shared__ int best_cnt[SXL_THREADS / 64];
    __shared__ int best_pos[SXL_THREADS / 64];
    const int64_t t = blockIdx.x;
    const int64_t p_lo = ptr[tiles[t]], p_hi = ptr[tiles[t + 1]];
    const int64_t cnt = p_hi - p_lo;
    const int tid = threadIdx.x;
    int32_t v = INT32_MAX; // padding sorts last and is never counted
    if (cnt >= SXL_THREADS) v = idx[p_lo + (cnt * tid) / SXL_THREADS];
    else if (tid < cnt) v = idx[p_lo + tid];
    s[tid] = v;
    __syncthreads();
    for (int k = 2; k <= SXL_THREADS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) { // bitonic sort, ascending
            const int other = tid ^ j;
            if (other > tid) {
                const int32_t a = s[tid], b = s[other];
                const bool up = (tid & k) == 0;
                if ((a > b) == up) {
                    s[tid] = b;
                    s[other] = a;
                }
            }
            __syncthreads();
        }
    int count = 0;
    if (s[tid] != INT32_MAX) {
        const int64_t lim = static_cast<int64_t>(s[tid]) + SXL_CAP;
        int lo = tid, hi = SXL_THREADS;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (static_cast<int64_t>(s[mid]) < lim) lo = mid + 1;
            else hi = mid;
        }
        count = lo - tid;
    }
    int pos = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int c2 = __shfl_down(count, o, 64), p2 = __shfl_down(pos, o, 64);
        if (c2 > count || (c2 == count && p2 < pos)) {
            count = c2;
            pos = p2;
        }
    }
    if ((tid & 63) == 0) {
        best_cnt[tid >> 6] = count;
        best_pos[tid >> 6] = pos;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < SXL_THREADS / 64; ++w)
            if (best_cnt[w] > count || (best_cnt[w] == count && best_pos[w] < pos)) {
                count = best_cnt[w];
                pos = best_pos[w];
            }
        int64_t lo = (count > 0) ? s[pos] : 0;
        const int32_t extent = (count > 0) ? s[pos + count - 1] - s[pos] + 1 : 0;
        if (lo > bound - SXL_CAP) lo = bound - SXL_CAP;
        if (lo < 0) lo = 0;
        win_lo[t] = static_cast<int32_t>(lo);
        stat[3 * t + 0] = count;                                                   // samples covered
        stat[3 * t + 1] = static_cast<int32_t>(cnt < SXL_THREADS ? cnt : SXL_THREADS); // samples taken
        stat[3 * t + 2] = extent;                                                  // rows they span
    }
}

} // namespace

// *win_lo_out stays NULL when the operand is shorter than a window; *useful_out = 1 when the auto rule
// (file header) says the windowed walk should pay off on this matrix
int sx_window_setup(sx_ctx *ctx, const int64_t *tiles, int64_t ntiles, const int64_t *ptr, const int32_t *idx,
                    int64_t bound, int32_t **win_lo_out, int *useful_out, int *local_out) {
    *win_lo_out = nullptr;
    *useful_out = 0;
    if (ntiles == 0 || bound < SXL_CAP) return SX_OK;
    int32_t *wl = nullptr, *stat = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&wl), sizeof(int32_t) * ntiles));
    if (sx_dmalloc(reinterpret_cast<void **>(&stat), sizeof(int32_t) * 3 * ntiles) != hipSuccess) {
        (void)sx_dfree(wl);
        sx_set_error("out of device memory for the window statistics");
        return SX_ERR_NOMEM;
    }
    hipLaunchKernelGGL(k_win_lo, dim3(static_cast<unsigned>(ntiles)), dim3(SXL_THREADS), 0, ctx->stream, tiles, ntiles,
                       ptr, idx, bound, wl, stat);
    std::vector<int32_t> h(static_cast<size_t>(3 * ntiles));
    const bool ok = hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(h.data(), stat, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, ctx->stream) ==
                        hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess;
    (void)sx_dfree(stat);
    if (!ok) {
        (void)sx_dfree(wl);
        sx_set_error("window table kernel failed");
        return SX_ERR_HIP;
    }
    int64_t covered = 0, taken = 0;
    std::vector<int32_t> extent(static_cast<size_t>(ntiles));
    for (int64_t t = 0; t < ntiles; ++t) {
        covered += h[3 * t];
        taken += h[3 * t + 1];
        extent[t] = h[3 * t + 2];
    }
    std::nth_element(extent.begin(), extent.begin() + ntiles / 2, extent.end());
    *useful_out = (taken > 0 && 2 * covered >= taken && extent[ntiles / 2] >= 1024) ? 1 : 0;
    if (local_out) *local_out = (taken > 0 && 2 * covered >= taken) ? 1 : 0; // the gathers of a tile cluster: no case for operand slabs
    *win_lo_out = wl;
    return SX_OK;
}

// tiles per window load the column walk of A should use under the context's "window" option
// (-1 auto, 0 off, 1/2/4/8 forced); 0 = plain walk.  Builds the table on first use.
int sx_window_run_csc(sx_ctx *ctx, const sx_matrix *A, int *run_out) {
    *run_out = 0;
    if (ctx->opt_window == 0 || !A->csc_ptr) return SX_OK;
    if (!A->csc_win_tried) {
        A->csc_win_tried = 1;
        SX_TRY(sx_window_setup(ctx, A->csc_tiles, A->n_csc_tiles, A->csc_ptr, A->csc_idx, A->m, &A->csc_win_lo,
                               &A->csc_win_useful, &A->csc_win_local));
    }
    if (!A->csc_win_lo) return SX_OK;
    if (ctx->opt_window < 0) *run_out = A->csc_win_useful ? 4 : 0;
    else *run_out = ctx->opt_window >= 8 ? 8 : ctx->opt_window >= 4 ? 4 : ctx->opt_window >= 2 ? 2 : 1;
    return SX_OK;
}
