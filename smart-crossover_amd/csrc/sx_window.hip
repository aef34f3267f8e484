// Optional LDS operand window for the column walk (K1), off by default ("window" option = 1 enables).
//
// With the plain walk (sx_segwalk.h) HBM traffic is already minimal, but every y[row] gather is its own
// vector-L1 lookup and ~45 % of them miss the 32 KiB L1 at the default workload (profiles/r01/pmc_*.txt):
// K1 runs at 0.6-0.7 of the HBM peak when the gathers hit (narrow staircase) and at 0.45 when they do
// not.  This variant serves most gathers from LDS: each 256-lane workgroup loads, with coalesced loads
// whose latency overlaps the entry stream, the 32 KiB window of the operand (4096 doubles) that covers
// most row indices of its tile; indices outside the window fall back to global memory, so results are
// bit-identical to the plain walk.  48 KiB of LDS -> 3 workgroups per CU.
//
// Measured (profiles/r01/kbench_window_variants.txt): +10 % at the default staircase window (W = 4096
// rows), -19 % when the gathers already hit L1 (W = 64), neutral without locality.  A second design --
// persistent 1024-lane workgroups sharing a sliding 64 KiB ring with register prefetch -- was built and
// measured slower everywhere (16 waves in lockstep lose the overlap that five independent workgroups
// per CU provide) and was removed.  Hence: opt-in, not default.
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

constexpr int SXL_CAP = 4096;      // window length (doubles)
constexpr int SXL_THREADS = 1024;  // setup kernel: sample size for the densest-window search

// per tile: start of the SXL_CAP-long index window that covers most of the tile's entries,
// found on an evenly spaced sample of up to 1024 entries (sorted in LDS, then one binary search each)
__global__ __launch_bounds__(SXL_THREADS) void k_win_lo(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                        const int64_t *__restrict__ ptr,
                                                        const int32_t *__restrict__ idx, int64_t bound,
                                                        int32_t *__restrict__ win_lo) {
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
    if (s[tid] != INT32_MAX) { // sample values in [s[tid], s[tid] + CAP)
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
        if (lo > bound - SXL_CAP) lo = bound - SXL_CAP;
        if (lo < 0) lo = 0;
        win_lo[t] = static_cast<int32_t>(lo);
    }
}

struct StageWin {
    const double *__restrict__ vec;
    const double *win;
    int64_t wlo;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const {
        const uint64_t d = static_cast<uint64_t>(static_cast<int64_t>(i) - wlo);
        const double yv = (d < static_cast<uint64_t>(SXL_CAP)) ? win[d] : vec[i];
        o[0] = v * yv;
    }
};

__global__ __launch_bounds__(SX_WG) void k_score_columns_lw(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle, const int32_t *__restrict__ win_lo,
    const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, const double *__restrict__ val,
    int64_t m, const double *__restrict__ y, const double *__restrict__ c, const double *__restrict__ x,
    const double *__restrict__ l, const double *__restrict__ u, double gamma, double *__restrict__ s_d,
    uint8_t *__restrict__ code) {
    __shared__ sx_walk_lds<1, 2048> lds;
    __shared__ double win[SXL_CAP];
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    const int64_t wlo = win_lo[tile];
#pragma unroll
    for (int r = 0; r < SXL_CAP / SX_WG; ++r) { // 16 independent coalesced 8-byte loads per lane
        const int k = r * SX_WG + threadIdx.x;
        int64_t row = wlo + k;
        if (row > m - 1) row = m - 1;
        win[k] = y[row];
    }
    __syncthreads();
    double acc[1];
    int64_t j;
    bool valid;
    double cj = 0.0, xj = 0.0, lj = 0.0, uj = 0.0;
    auto pre = [&](int64_t seg, bool ok) {
        if (ok) {
            cj = c[seg];
            if (code) {
                xj = x[seg];
                lj = l[seg];
                uj = u[seg];
            }
        }
    };
    sx_segwalk<1, 2048, false>(tiles, tile, colptr, rowidx, val, StageWin{y, win, wlo}, lds, j, valid, acc, pre);
    if (!valid) return;
    const double sd = cj - acc[0];
    if (s_d) s_d[j] = sd;
    if (code) {
        const bool low = (xj - lj) < (gamma * sd);
        const bool up = (uj - xj) < (gamma * (-sd));
        code[j] = static_cast<uint8_t>((low ? SX_CODE_LOW : 0u) | (up ? SX_CODE_UP : 0u));
    }
}

} // namespace

// window table of one pointer array; *win_lo_out stays NULL when the operand is shorter than a window
int sx_window_setup(sx_ctx *ctx, const int64_t *tiles, int64_t ntiles, const int64_t *ptr, const int32_t *idx,
                    int64_t bound, int32_t **win_lo_out) {
    *win_lo_out = nullptr;
    if (ntiles == 0 || bound < SXL_CAP) return SX_OK;
    int32_t *wl = nullptr;
    SX_HIP(hipMalloc(reinterpret_cast<void **>(&wl), sizeof(int32_t) * ntiles));
    hipLaunchKernelGGL(k_win_lo, dim3(static_cast<unsigned>(ntiles)), dim3(SXL_THREADS), 0, ctx->stream, tiles, ntiles,
                       ptr, idx, bound, wl);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(wl);
        sx_set_error("window table kernel failed");
        return SX_ERR_HIP;
    }
    *win_lo_out = wl;
    return SX_OK;
}

int sx_window_score_columns(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c, const double *x,
                            const double *l, const double *u, double gamma, double *s_d, uint8_t *code) {
    const int swz = ctx->opt_xcd_swizzle;
    const unsigned grid = swz ? static_cast<unsigned>(((A->n_csc_tiles + 7) >> 3) << 3)
                              : static_cast<unsigned>(A->n_csc_tiles);
    hipLaunchKernelGGL(k_score_columns_lw, dim3(grid), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, A->n_csc_tiles, swz,
                       A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, A->m, y, c, x, l, u, gamma, s_d, code);
    SX_HIP(hipGetLastError());
    return SX_OK;
}
