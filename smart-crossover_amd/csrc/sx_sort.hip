// K9: ranking of the flow indicators -- np.argsort(ind)[::-1] (network_methods/net_manager.py:184,379).
// The reference's sort is numpy's default *unstable* sort, whose order inside runs of equal keys is
// not reproducible; this library fixes the rule: descending key, and inside a run of equal keys
// descending index, i.e. a stable ascending argsort read backwards (np.argsort(kind="stable")[::-1]).
// NaN keys rank as the largest values, as in numpy.
//
// Implementation: least-significant-digit radix sort on the order-preserving 64-bit image of the
// double, 8 passes of 8 bits, (key, int32 index) pairs, three kernels per pass:
//   histogram   per-tile digit counts, digit-major so that one exclusive scan yields global offsets
//   scan        sx_scan_exclusive (sx_compact.hip)
//   scatter     stable: ranks inside a tile come from wave ballots (match-any over the 8 digit bits)
//               and per-wave digit counters in LDS, visited in element order
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = SX_WG * SORT_ITEMS; // 2048 keys per workgroup

__device__ __forceinline__ uint64_t key_image(double v) {
    if (v != v) return ~0ull; // NaN -> largest, all NaNs tie
    if (v == 0.0) v = 0.0;    // -0.0 and +0.0 tie, as in numpy
    uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(SX_WG) void k_sort_init(int64_t n, const double *__restrict__ key,
                                                     uint64_t *__restrict__ img, int32_t *__restrict__ idx) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        img[i] = key_image(key[i]);
        idx[i] = static_cast<int32_t>(i);
    }
}

// hist[d * nblocks + block] = number of keys of this tile whose digit is d
__global__ __launch_bounds__(SX_WG) void k_sort_hist(int64_t n, const uint64_t *__restrict__ img, int shift,
                                                     int64_t nblocks, int64_t *__restrict__ hist) {
    __shared__ unsigned cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = static_cast<int64_t>(blockIdx.x) * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int64_t i = base + r * SX_WG + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(img[i] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    hist[static_cast<int64_t>(threadIdx.x) * nblocks + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(SX_WG) void k_sort_scatter(int64_t n, const uint64_t *__restrict__ img_in,
                                                        const int32_t *__restrict__ idx_in, int shift,
                                                        int64_t nblocks, const int64_t *__restrict__ offs,
                                                        uint64_t *__restrict__ img_out,
                                                        int32_t *__restrict__ idx_out) {
    __shared__ unsigned run[256];                 // keys of each digit seen so far in this tile
    __shared__ unsigned wcnt[SX_WG / 64][256];    // per wave, this round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    run[threadIdx.x] = 0;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * SORT_TILE;
    for (int r = 0; r < SORT_ITEMS; ++r) {
#pragma unroll
        for (int w = 0; w < SX_WG / 64; ++w) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
        const int64_t i = base + r * SX_WG + threadIdx.x;
        const bool live = i < n;
        uint64_t k = 0;
        int32_t v = 0;
        unsigned d = 0;
        if (live) {
            k = img_in[i];
            v = idx_in[i];
            d = static_cast<unsigned>((k >> shift) & 0xFF);
        }
        // lanes of this wave holding the same digit
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot(live && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
        if (live && rank_in_wave == 0) wcnt[wave][d] = __popcll(same);
        __syncthreads();
        unsigned local = 0;
        if (live) {
            local = run[d] + rank_in_wave;
            for (int w = 0; w < wave; ++w) local += wcnt[w][d];
        }
        __syncthreads();
        {
            unsigned add = 0;
#pragma unroll
            for (int w = 0; w < SX_WG / 64; ++w) add += wcnt[w][threadIdx.x];
            run[threadIdx.x] += add;
        }
        if (live) {
            const int64_t dst = offs[static_cast<int64_t>(d) * nblocks + blockIdx.x] + local;
            img_out[dst] = k;
            idx_out[dst] = v;
        }
        __syncthreads();
    }
}

// out[i] = sorted_idx[n-1-i]  (descending key, descending index inside ties)
__global__ __launch_bounds__(SX_WG) void k_sort_reverse(int64_t n, const int32_t *__restrict__ idx,
                                                        int64_t *__restrict__ out) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * SX_WG)
        out[i] = idx[n - 1 - i];
}

inline unsigned grid1d(int64_t n, int64_t cap = 8192) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

} // namespace

// Stable LSD radix sort of n (64-bit key, int32 payload) pairs on the `nbytes` low-order bytes of the key
// (internal: the transposition and the row-block layout builder sort integer keys with it).  img[0] / idx[0]
// hold the input, img[1] / idx[1] are the second buffers, hist / offs hold 256 * nblocks + 1 counters each
// (nblocks = sx_sort_blocks(n)); *cur_out says which buffer pair holds the result.
int64_t sx_sort_blocks(int64_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }

int sx_sort_pairs(sx_ctx *ctx, int64_t n, uint64_t *const img[2], int32_t *const idx[2], int64_t *hist, int64_t *offs,
                  int nbytes, int *cur_out) {
    const int64_t nblocks = sx_sort_blocks(n);
    hipStream_t s = ctx->stream;
    int cur = 0;
    for (int pass = 0; pass < nbytes; ++pass) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(k_sort_hist, dim3(static_cast<unsigned>(nblocks)), dim3(SX_WG), 0, s, n, img[cur], shift,
                           nblocks, hist);
        SX_TRY(sx_scan_exclusive(ctx, hist, 256 * nblocks, offs));
        hipLaunchKernelGGL(k_sort_scatter, dim3(static_cast<unsigned>(nblocks)), dim3(SX_WG), 0, s, n, img[cur],
                           idx[cur], shift, nblocks, offs, img[cur ^ 1], idx[cur ^ 1]);
        cur ^= 1;
    }
    SX_HIP(hipGetLastError());
    *cur_out = cur;
    return SX_OK;
}

SX_API int sx_argsort_desc_dev(sx_ctx *ctx, int64_t n, const double *key, int64_t *idx_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    SX_REQUIRE(n < INT32_MAX, "n exceeds the int32 payload range");
    if (n == 0) return SX_OK;
    SX_REQUIRE(key && idx_out, "NULL argument");
    const int64_t nblocks = sx_sort_blocks(n);
    hipStream_t s = ctx->stream;
    // double buffers and histograms live in the context's second grow-only block (the scan helper works
    // in the first one): no allocation and no host synchronisation per call once the block is large enough
    const size_t n8 = (sizeof(uint64_t) * static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255);
    const size_t n4 = (sizeof(int32_t) * static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255);
    const size_t nh = (sizeof(int64_t) * (256 * static_cast<size_t>(nblocks) + 1) + 255) & ~static_cast<size_t>(255);
    SX_TRY(sx_reserve2(ctx, 2 * n8 + 2 * n4 + 2 * nh));
    char *base = static_cast<char *>(ctx->ws2);
    uint64_t *img[2] = {reinterpret_cast<uint64_t *>(base), reinterpret_cast<uint64_t *>(base + n8)};
    int32_t *idx[2] = {reinterpret_cast<int32_t *>(base + 2 * n8), reinterpret_cast<int32_t *>(base + 2 * n8 + n4)};
    int64_t *hist = reinterpret_cast<int64_t *>(base + 2 * n8 + 2 * n4);
    int64_t *offs = reinterpret_cast<int64_t *>(base + 2 * n8 + 2 * n4 + nh);

    hipLaunchKernelGGL(k_sort_init, dim3(grid1d(n)), dim3(SX_WG), 0, s, n, key, img[0], idx[0]);
    int cur = 0;
    SX_TRY(sx_sort_pairs(ctx, n, img, idx, hist, offs, 8, &cur));
    hipLaunchKernelGGL(k_sort_reverse, dim3(grid1d(n)), dim3(SX_WG), 0, s, n, idx[cur], idx_out);
    SX_HIP(hipGetLastError());
    return SX_OK;
}
