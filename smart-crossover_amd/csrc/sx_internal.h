// Internal definitions shared by the translation units of libsxhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "sxhip.h"

#define SX_API extern "C" __attribute__((visibility("default")))

void sx_set_error(const char *fmt, ...);

#define SX_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            sx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,          \
                         __LINE__);                                                                \
            return e_ == hipErrorOutOfMemory ? SX_ERR_NOMEM : SX_ERR_HIP;                          \
        }                                                                                          \
    } while (0)

#define SX_REQUIRE(cond, ...)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            sx_set_error(__VA_ARGS__);                                                             \
            return SX_ERR_INVALID;                                                                 \
        }                                                                                          \
    } while (0)

#define SX_TRY(call)                                                                               \
    do {                                                                                           \
        int r_ = (call);                                                                           \
        if (r_ != SX_OK) return r_;                                                                \
    } while (0)

// Every sparse array on the device is allocated with SX_PAD trailing zero entries so that the
// 4-wide vector loads of the segment-walk kernels may run past the last entry.
constexpr int64_t SX_PAD = 8;

struct sx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // growable scratch (never grown inside an _dev call that could be under graph capture:
    // sx_reserve is called by the host-pointer wrappers and by the first use of each size)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    void *ws2 = nullptr; // second block (sx_reserve2)
    size_t ws2_bytes = 0;
    void *ws3 = nullptr; // third block (sx_reserve3): the temporaries of the network solvers, carved by sx_arena
    size_t ws3_bytes = 0;
    // dense basis inverse of the last simplex session that was destroyed: the next session with the same
    // row count takes it over instead of allocating (a crossover opens one session per call)
    double *spare_binv = nullptr;
    int64_t spare_binv_m = 0;
    // tree arrays of the last dual network simplex solve (sx_netdual.hip): the next round of a column generation
    // starts from that very tree, so its preorder, sizes and potentials need not be rebuilt
    void *nd_tree = nullptr;      // int4[V] {parent, pred arc, pos, size}
    int32_t *nd_order = nullptr;  // [V] node at preorder position
    double *nd_y = nullptr;       // [V] potentials
    int64_t nd_tree_V = 0;
    int nd_tree_root = -1;
    // pinned staging of sx_download (two halves, used alternately): a device-to-host copy into never-touched pageable
    // memory -- a fresh numpy array -- faults its pages inside the driver (3 ms per 8 MB measured in a process with a
    // large heap); into pinned memory it runs at link speed and the memcpy out of it at memory speed
    void *pin = nullptr;
    size_t pin_half = 0;
    // one big device block kept between calls and allocatable AHEAD of its use on a helper thread (sx_ctx_prefetch_block):
    // hipMalloc of tens of GB takes 25-50 ms per GB (0.5-1.4 s for the 28 GB of eta file + tableau the sparse crossover needs
    // at 1e6 rows), longer than the host work it could hide behind inside the call -- the backend asks for it before the
    // first-order stage, and a second call of a process finds it still there
    void *blk = nullptr;
    size_t blk_bytes = 0;
    std::thread *blk_thread = nullptr; // allocating blk_pending (blk_pending_bytes)
    void *blk_pending = nullptr;
    size_t blk_pending_bytes = 0;
    // timers
    hipEvent_t t0[8];
    hipEvent_t t1[8];
    int t_depth = 0;
    bool t_made = false;
    int cu_count = 0;
    std::vector<hipEvent_t> markers;
    // tuning knobs (sx_ctx_set_option); defaults are the measured best on MI355X
    int opt_xcd_swizzle = 1; // XCD-contiguous block -> tile map
    int opt_nt_stream = 0;   // non-temporal loads for the streamed entry arrays
    int opt_chunk = 4096;    // staged entries per chunk (2048 or 4096)
    int opt_window = -1;     // LDS operand window of the column walk: -1 auto, 0 off, 1/2/4/8 tiles per load
    int opt_graph = 1;       // replay the CG iteration batch as a hipGraph
    int opt_spx_defer = -1;  // K16 basis inverse: -1 auto, 0 rank-one update per pivot, 1 rank-32 update per batch
    int opt_rowblock = -1;   // column-blocked row layout of the row walk: -1 auto, 0 off, 1 whenever possible
    int opt_spx_check = 2048; // K16: pivots between two checks of A x + s = b (drift of the explicit inverse)
    int opt_spx_force_reinvert = 0; // K16, tests: rebuild the inverse at every check whatever the residual says
    int opt_netsimplex = -1; // K16n for network re-solves: -1 by size, 0 never (general simplex), 1 whenever it applies
    int opt_ns_lds = 1;      // K16n: tree arrays and potentials in LDS when they fit (V <= 4608)
    int opt_ns_block = 0;    // K16n network simplex: arcs priced per lane and block (0: by size, 1..64)
    int opt_netdual = -1;    // K16d dual network simplex on the whole GPU: -1 / 1 whenever it applies, 0 never
    int opt_nd_grid = 0;     // K16d: workgroups of its cooperative grid (0: by size)
    int opt_spx_pricing = 1; // K16 entering variable: 0 Dantzig (largest reduced cost), 1 Devex reference weights
    int opt_slabs = -1;        // operand slabs of a walk without locality (sx_slabs.h): -1 auto, 0 never, R >= 2: R slabs
    int opt_run_prefetch = 0;  // windowed column walk (K1, K10): loads one step ahead (sx_runwalk.h); 0: tile by tile
    int opt_rb_long_xcd = -1;  // row-blocked walks: the long-row super-tiles dealt over the XCDs (sx_rowblock.h: order[]): -1 when the
                               // matrix puts them unevenly, 0 never, 1 always
    int opt_rb_dense_min = 512; // entries from which a column block of a super-tile gets an LDS window (read when a layout is built;
                                // sx_rowblock.h RB_DENSE_MIN)
    int opt_rb_long_rows = 64; // rows per super-tile of long rows (read when a layout is built; sx_rowblock.h RB_LONG_ROWS)
    int opt_rb_stage_long = 0; // row-blocked layout: products of the long rows by a column-ordered pre-pass (0: gather in the
                               // walk).  Measured SLOWER (K2 0.442 vs 0.381 ms at config 5: the 1e7 scattered 8-byte stores cost more
                               // than the 1e7 gathered lines they replace; profiles/r03/experiments/k2_long_row_staging.md): off
};

// Device memory of the library (sx_pool.hip): hipMalloc / hipFree with freed blocks kept for the next request of their size
// class.  sx_dfree synchronises the device as hipFree does; nullptr is fine.
hipError_t sx_pool_malloc(void **p, size_t bytes);
hipError_t sx_pool_free(void *p);
template <class T>
inline hipError_t sx_dmalloc(T **p, size_t bytes) {
    return sx_pool_malloc(reinterpret_cast<void **>(p), bytes);
}
inline hipError_t sx_dfree(void *p) { return sx_pool_free(p); }

int sx_reserve(sx_ctx *ctx, size_t bytes);  // ensure ctx->ws holds >= bytes
// the context's big block (sx_ctx.hip): take hands it to the caller when it holds >= bytes (waits for a pending
// allocation; false: none of that size), give hands one back (the larger of the two is kept)
bool sx_ctx_take_block(sx_ctx *ctx, size_t bytes, void **base, size_t *got);
void sx_ctx_give_block(sx_ctx *ctx, void *base, size_t bytes);
int sx_reserve2(sx_ctx *ctx, size_t bytes); // ensure ctx->ws2 holds >= bytes
int sx_reserve3(sx_ctx *ctx, size_t bytes); // ensure ctx->ws3 holds >= bytes

// Device temporaries of one call carved out of ctx->ws3 (reserved once for the call's upper bound): a network solve
// needs 25-45 arrays, and hipMalloc / hipFree cost 50-100 us apiece -- more than a small solve itself.  What does
// not fit falls back on hipMalloc and is freed with the arena.
struct sx_arena {
    sx_ctx *ctx;
    size_t off = 0;
    std::vector<void *> extra;
    explicit sx_arena(sx_ctx *c) : ctx(c) {}
    ~sx_arena() {
        for (void *q : extra) (void)sx_dfree(q);
    }
    template <class T>
    int get(size_t count, T **out) {
        const size_t need = (sizeof(T) * (count ? count : 1) + 255) & ~static_cast<size_t>(255);
        if (ctx->ws3 && off + need <= ctx->ws3_bytes) {
            *out = reinterpret_cast<T *>(static_cast<char *>(ctx->ws3) + off);
            off += need;
            return SX_OK;
        }
        void *d = nullptr;
        SX_HIP(sx_dmalloc(&d, need));
        extra.push_back(d);
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

struct sx_matrix {
    sx_ctx *ctx = nullptr;
    int64_t m = 0, n = 0, nnz = 0;
    int64_t *csr_ptr = nullptr;
    int32_t *csr_idx = nullptr;
    double *csr_val = nullptr;
    int64_t *csc_ptr = nullptr;
    int32_t *csc_idx = nullptr;
    double *csc_val = nullptr;
    // tile tables (sx_tiles.hip): first segment of every tile + sentinel
    int64_t *csr_tiles = nullptr;
    int64_t n_csr_tiles = 0;
    int64_t *csc_tiles = nullptr;
    int64_t n_csc_tiles = 0;
    double csr_imbalance = 1.0, csc_imbalance = 1.0; // of the XCD-contiguous tile ranges (sx_build_tiles)
    // optional per-tile operand window of the column walk (sx_window.hip), built on first use
    mutable int32_t *csc_win_lo = nullptr;
    mutable int csc_win_tried = 0;
    mutable int csc_win_useful = 0; // verdict of the auto rule (sampling; overruled by the timing of sx_lp_kernels.hip window_autotune)
    mutable int csc_win_tuned = 0;
    mutable int csc_swizzle_off = 0; // window_autotune: this matrix' column walk is faster WITHOUT the XCD-contiguous tile map
    mutable int csc_win_local = 0;  // at least half of a tile's sampled indices inside one 4096-row window: the gathers have locality
    // optional column-blocked copy of the rows (sx_rowblock.h), built on first use of a row walk
    mutable struct sx_rowblock *rb = nullptr;
    mutable int rb_tried = 0; // 1: the automatic rule has spoken, 2: so has a forced build
    // optional operand slabs of a walk without locality (sx_slabs.h): [0] row walk, [1] column walk
    mutable struct sx_slabs *slabs[2] = {nullptr, nullptr};
    mutable int slabs_tried[2] = {0, 0}; // the slab count slabs[] was built (or refused) for
};

int sx_window_setup(sx_ctx *ctx, const int64_t *tiles, int64_t ntiles, const int64_t *ptr, const int32_t *idx,
                    int64_t bound, int32_t **win_lo_out, int *useful_out, int *local_out = nullptr);
// tiles per window load for A's column walk under ctx's "window" option (0 = plain walk)
int sx_window_run_csc(sx_ctx *ctx, const sx_matrix *A, int *run_out);

// stable CSR -> CSC transposition of device arrays (sx_transpose.hip); outputs are owned by the caller
int sx_transpose_dev(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int32_t *col,
                     const double *val, int64_t **colptr_out, int32_t **row_out, double **val_out);

// *imbalance_out (optional): serial cost of the heaviest of the eight XCD-contiguous tile ranges over the mean, a
// tile costing its average segment length (the adds a lane makes one after another) + 64
int sx_build_tiles(sx_ctx *ctx, const int64_t *ptr_dev, int64_t nseg, int64_t **tiles_out,
                   int64_t *ntiles_out, double *imbalance_out = nullptr);
// the XCD-contiguous tile map (speed only) is used for a walk unless its ranges are that uneven: long segments that
// sit together -- the linking rows at the head of an LP -- would all queue on one XCD
constexpr double SX_SWIZZLE_MAX_IMBALANCE = 1.5;
// XCD-contiguous tile map of A's COLUMN walks (K1, K10, the CG's column pass): the context's option, unless the timing of
// window_autotune (sx_lp_kernels.hip) found this matrix faster without it
inline int sx_csc_swizzle(const sx_ctx *ctx, const sx_matrix *A) {
    return (ctx->opt_xcd_swizzle && A->n_csc_tiles >= 64 && !A->csc_swizzle_off) ? 1 : 0;
}
// exclusive scan of in[0..n) into out[0..n] (out[n] = total); uses ctx->ws (sx_compact.hip)
int sx_scan_exclusive(sx_ctx *ctx, const int64_t *in, int64_t n, int64_t *out);

// stable LSD radix sort of (uint64 key, int32 payload) pairs on the low `nbytes` bytes (sx_sort.hip)
int64_t sx_sort_blocks(int64_t n);
int sx_sort_pairs(sx_ctx *ctx, int64_t n, uint64_t *const img[2], int32_t *const idx[2], int64_t *hist, int64_t *offs,
                  int nbytes, int *cur_out);

// band LU (sx_bandlu.hip): does sx_bandlu_create_dev take these widths?  (the sparse crossover asks before it builds a band)
bool sx_bandlu_supports(int kl, int ku);

// dense LU (sx_denselu.hip): the device matrix of a handle (column major, *ld doubles per column), for the module that fills it
int sx_denselu_matrix(sx_denselu *h, double **a_dev, int64_t *ld);

// RAII guard: make the context's device current for the duration of a call.
struct sx_device_guard {
    int prev = -1;
    bool ok = true;
    explicit sx_device_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~sx_device_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

#define SX_ENTER(ctx)                                                                              \
    SX_REQUIRE((ctx) != nullptr, "ctx is NULL");                                                   \
    sx_device_guard guard_((ctx)->device);                                                         \
    SX_REQUIRE(guard_.ok, "cannot select device %d", (ctx)->device)

// temporary device copies of host arrays for the host-pointer wrappers
struct sx_stage {
    sx_ctx *ctx;
    std::vector<void *> bufs;
    explicit sx_stage(sx_ctx *c) : ctx(c) {}
    ~sx_stage() {
        for (void *p : bufs) (void)sx_dfree(p);
    }
    // allocate `bytes` on the device; copy from src when src != nullptr
    int in(const void *src, size_t bytes, void **out) {
        *out = nullptr;
        if (bytes == 0) bytes = 8;
        void *p = nullptr;
        SX_HIP(sx_dmalloc(&p, bytes));
        bufs.push_back(p);
        if (src) SX_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        *out = p;
        return SX_OK;
    }
    int out(void *dst_host, const void *src_dev, size_t bytes) {
        if (dst_host && bytes)
            SX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return SX_OK;
    }
};
