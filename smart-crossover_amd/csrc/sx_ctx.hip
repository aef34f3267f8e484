// Context, memory plumbing, stopwatch and matrix residency of libsxhip.so.
#include "sx_internal.h"
#include "sx_rowblock.h"
#include "sx_slabs.h"

#include <algorithm>

static thread_local char g_err[512] = "";

void sx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

SX_API int sx_abi_version(void) { return SX_ABI_VERSION; }
SX_API const char *sx_last_error(void) { return g_err; }

SX_API int sx_device_count(int *count) {
    SX_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return SX_OK;
}

SX_API int sx_ctx_create(int device, void *stream, sx_ctx **out) {
    SX_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    SX_HIP(hipGetDeviceCount(&ndev));
    SX_REQUIRE(device >= 0 && device < ndev, "device %d out of range (%d visible)", device, ndev);
    sx_device_guard guard(device);
    SX_REQUIRE(guard.ok, "cannot select device %d", device);
    sx_ctx *ctx = new (std::nothrow) sx_ctx();
    if (!ctx) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    ctx->device = device;
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
        ctx->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            sx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return SX_ERR_HIP;
        }
        ctx->owns_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    *out = ctx;
    return SX_OK;
}

namespace {
void blk_join(sx_ctx *ctx) { // a pending allocation becomes the context's block (the larger of the two stays)
    if (!ctx->blk_thread) return;
    ctx->blk_thread->join();
    delete ctx->blk_thread;
    ctx->blk_thread = nullptr;
    void *p = ctx->blk_pending;
    const size_t b = ctx->blk_pending_bytes;
    ctx->blk_pending = nullptr;
    ctx->blk_pending_bytes = 0;
    if (p) sx_ctx_give_block(ctx, p, b);
}
} // namespace

bool sx_ctx_take_block(sx_ctx *ctx, size_t bytes, void **base, size_t *got) {
    blk_join(ctx);
    if (!ctx->blk || ctx->blk_bytes < bytes) return false;
    *base = ctx->blk;
    *got = ctx->blk_bytes;
    ctx->blk = nullptr;
    ctx->blk_bytes = 0;
    return true;
}

void sx_ctx_give_block(sx_ctx *ctx, void *base, size_t bytes) {
    if (!base) return;
    if (ctx->blk && ctx->blk_bytes >= bytes) {
        (void)sx_dfree(base);
        return;
    }
    if (ctx->blk) (void)sx_dfree(ctx->blk);
    ctx->blk = base;
    ctx->blk_bytes = bytes;
}

// Start allocating a block of `bytes` on a helper thread (returns at once); the sparse crossover takes it when it starts.
// A block of that size already there, or on its way: nothing to do.  bytes = 0: free what the context holds.
SX_API int sx_ctx_prefetch_block(sx_ctx *ctx, size_t bytes) {
    SX_ENTER(ctx);
    if (bytes == 0) {
        blk_join(ctx);
        if (ctx->blk) (void)sx_dfree(ctx->blk);
        ctx->blk = nullptr;
        ctx->blk_bytes = 0;
        return SX_OK;
    }
    if (ctx->blk_thread && ctx->blk_pending_bytes >= bytes) return SX_OK;
    blk_join(ctx);
    if (ctx->blk && ctx->blk_bytes >= bytes) return SX_OK;
    size_t free_b = 0, total_b = 0;
    SX_HIP(hipMemGetInfo(&free_b, &total_b));
    if (ctx->blk) { // (replaced by the larger one: its memory counts as free)
        (void)sx_dfree(ctx->blk);
        free_b += ctx->blk_bytes;
        ctx->blk = nullptr;
        ctx->blk_bytes = 0;
    }
    if (static_cast<double>(bytes) > 0.6 * static_cast<double>(free_b)) return SX_OK; // (the call allocates what it can by itself)
    ctx->blk_pending_bytes = bytes;
    const int device = ctx->device;
    ctx->blk_thread = new (std::nothrow) std::thread([ctx, device, bytes]() {
        void *p = nullptr;
        if (hipSetDevice(device) != hipSuccess || sx_dmalloc(&p, bytes) != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
        }
        ctx->blk_pending = p;
    });
    if (!ctx->blk_thread) ctx->blk_pending_bytes = 0;
    return SX_OK;
}

SX_API int sx_ctx_destroy(sx_ctx *ctx) {
    if (!ctx) return SX_OK;
    sx_device_guard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    blk_join(ctx);
    if (ctx->blk) (void)sx_dfree(ctx->blk);
    if (ctx->ws) (void)sx_dfree(ctx->ws);
    if (ctx->ws2) (void)sx_dfree(ctx->ws2);
    if (ctx->ws3) (void)sx_dfree(ctx->ws3);
    if (ctx->spare_binv) (void)sx_dfree(ctx->spare_binv);
    if (ctx->nd_tree) (void)sx_dfree(ctx->nd_tree);
    if (ctx->nd_order) (void)sx_dfree(ctx->nd_order);
    if (ctx->nd_y) (void)sx_dfree(ctx->nd_y);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->t_made)
        for (int i = 0; i < 8; ++i) {
            (void)hipEventDestroy(ctx->t0[i]);
            (void)hipEventDestroy(ctx->t1[i]);
        }
    for (hipEvent_t e : ctx->markers)
        if (e) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SX_OK;
}

SX_API int sx_ctx_set_option(sx_ctx *ctx, const char *key, int64_t value) {
    SX_REQUIRE(ctx != nullptr && key != nullptr, "ctx or key is NULL");
    if (!strcmp(key, "xcd_swizzle")) {
        ctx->opt_xcd_swizzle = value ? 1 : 0;
    } else if (!strcmp(key, "nt_stream")) {
        if (value != 0 && value != 1 && value != 2 && value != 16 && value != 17 && value != 18) {
            sx_set_error("nt_stream must be 0, 1, 2, 16, 17 or 18");
            return SX_ERR_INVALID;
        }
        ctx->opt_nt_stream = value;
    } else if (!strcmp(key, "chunk")) {
        SX_REQUIRE(value == 2048 || value == 4096, "chunk must be 2048 or 4096");
        ctx->opt_chunk = static_cast<int>(value);
    } else if (!strcmp(key, "window")) {
        ctx->opt_window = value < 0 ? -1 : static_cast<int>(value);
    } else if (!strcmp(key, "netsimplex")) {
        SX_REQUIRE(value >= -1 && value <= 1, "netsimplex must be -1 (by size), 0 (off) or 1 (whenever it applies)");
        ctx->opt_netsimplex = static_cast<int>(value);
    } else if (!strcmp(key, "netdual")) {
        SX_REQUIRE(value >= -1 && value <= 1, "netdual must be -1 / 1 (whenever it applies) or 0 (off)");
        ctx->opt_netdual = static_cast<int>(value);
    } else if (!strcmp(key, "nd_grid")) {
        SX_REQUIRE(value >= 0 && value <= 256, "nd_grid must be 0 (by size) or 1..256 workgroups");
        ctx->opt_nd_grid = static_cast<int>(value);
    } else if (!strcmp(key, "ns_lds")) {
        ctx->opt_ns_lds = value ? 1 : 0;
    } else if (!strcmp(key, "ns_block")) {
        SX_REQUIRE(value >= 0 && value <= 64, "ns_block must be 0 (by size) or 1..64 arcs per lane");
        ctx->opt_ns_block = static_cast<int>(value);
    } else if (!strcmp(key, "graph")) {
        ctx->opt_graph = value ? 1 : 0;
    } else if (!strcmp(key, "spx_defer")) {
        ctx->opt_spx_defer = value < 0 ? -1 : (value ? 1 : 0);
    } else if (!strcmp(key, "rowblock")) {
        SX_REQUIRE(value >= -1 && value <= 1, "rowblock must be -1 (auto), 0 (off) or 1 (whenever possible)");
        ctx->opt_rowblock = static_cast<int>(value);
    } else if (!strcmp(key, "slabs")) {
        SX_REQUIRE(value >= -1 && value != 1 && value <= 256, "slabs must be -1 (auto), 0 (never) or 2..256");
        ctx->opt_slabs = static_cast<int>(value);
    } else if (!strcmp(key, "run_prefetch")) {
        ctx->opt_run_prefetch = value ? 1 : 0;
    } else if (!strcmp(key, "rb_long_xcd")) {
        SX_REQUIRE(value >= -1 && value <= 1, "rb_long_xcd must be -1 (auto), 0 (never) or 1 (always)");
        ctx->opt_rb_long_xcd = static_cast<int>(value);
    } else if (!strcmp(key, "rb_dense_min")) {
        SX_REQUIRE(value >= 1, "rb_dense_min must be positive");
        ctx->opt_rb_dense_min = static_cast<int>(value > (1 << 30) ? (1 << 30) : value); // (read when a layout is built)
    } else if (!strcmp(key, "rb_long_rows")) {
        SX_REQUIRE(value >= 1 && value <= 64, "rb_long_rows must be in [1, 64]");
        ctx->opt_rb_long_rows = static_cast<int>(value); // (read when a layout is built)
    } else if (!strcmp(key, "rb_stage_long")) {
        ctx->opt_rb_stage_long = value ? 1 : 0; // (read when a layout is built)
    } else if (!strcmp(key, "spx_check")) {
        SX_REQUIRE(value >= 64 && value % 64 == 0, "spx_check must be a positive multiple of 64 pivots");
        ctx->opt_spx_check = static_cast<int>(value);
    } else if (!strcmp(key, "spx_force_reinvert")) {
        ctx->opt_spx_force_reinvert = value ? 1 : 0;
    } else if (!strcmp(key, "spx_pricing")) {
        SX_REQUIRE(value == 0 || value == 1, "spx_pricing must be 0 (Dantzig) or 1 (Devex)");
        ctx->opt_spx_pricing = static_cast<int>(value);
    } else {
        sx_set_error("unknown option '%s'", key);
        return SX_ERR_INVALID;
    }
    return SX_OK;
}

SX_API int sx_ctx_sync(sx_ctx *ctx) {
    SX_ENTER(ctx);
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_ctx_device_info(sx_ctx *ctx, char *name, size_t name_len, int *cu_count,
                              uint64_t *hbm_bytes) {
    SX_ENTER(ctx);
    hipDeviceProp_t prop;
    SX_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) {
        snprintf(name, name_len, "%s", prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = static_cast<uint64_t>(prop.totalGlobalMem);
    return SX_OK;
}

static int reserve_block(sx_ctx *ctx, void **block, size_t *have, size_t bytes) {
    if (bytes <= *have) return SX_OK;
    // the old block may still be in use by enqueued kernels
    SX_HIP(hipStreamSynchronize(ctx->stream));
    if (*block) SX_HIP(sx_dfree(*block));
    *block = nullptr;
    *have = 0;
    size_t want = std::max(bytes, static_cast<size_t>(1) << 20);
    want = (want + 255) & ~static_cast<size_t>(255);
    SX_HIP(sx_dmalloc(block, want));
    *have = want;
    return SX_OK;
}

int sx_reserve(sx_ctx *ctx, size_t bytes) { return reserve_block(ctx, &ctx->ws, &ctx->ws_bytes, bytes); }

// second grow-only block for kernels that call helpers using the first one (the radix sort keeps its
// double buffers here while sx_scan_exclusive works in ctx->ws)
int sx_reserve2(sx_ctx *ctx, size_t bytes) { return reserve_block(ctx, &ctx->ws2, &ctx->ws2_bytes, bytes); }
int sx_reserve3(sx_ctx *ctx, size_t bytes) { return reserve_block(ctx, &ctx->ws3, &ctx->ws3_bytes, bytes); }

SX_API int sx_malloc(sx_ctx *ctx, size_t bytes, void **dev_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(dev_out != nullptr, "dev_out is NULL");
    *dev_out = nullptr;
    if (bytes == 0) bytes = 8;
    SX_HIP(sx_dmalloc(dev_out, bytes));
    return SX_OK;
}

SX_API int sx_free(sx_ctx *ctx, void *dev) {
    SX_ENTER(ctx);
    if (!dev) return SX_OK;
    SX_HIP(hipStreamSynchronize(ctx->stream));
    SX_HIP(sx_dfree(dev));
    return SX_OK;
}

SX_API int sx_upload(sx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    SX_ENTER(ctx);
    if (bytes == 0) return SX_OK;
    SX_REQUIRE(dst_dev && src_host, "NULL pointer in sx_upload");
    SX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream)); // the host buffer may be reused on return
    return SX_OK;
}

SX_API int sx_download(sx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    SX_ENTER(ctx);
    if (bytes == 0) return SX_OK;
    SX_REQUIRE(dst_host && src_dev, "NULL pointer in sx_download");
    constexpr size_t HALF = static_cast<size_t>(8) << 20; // 8 MiB per half
    if (bytes >= (static_cast<size_t>(64) << 10)) {
        if (!ctx->pin && hipHostMalloc(&ctx->pin, 2 * HALF, hipHostMallocDefault) == hipSuccess) ctx->pin_half = HALF;
        if (ctx->pin) { // piece k + 1 is on its way while piece k is copied out
            char *dst = static_cast<char *>(dst_host);
            const char *src = static_cast<const char *>(src_dev);
            char *half[2] = {static_cast<char *>(ctx->pin), static_cast<char *>(ctx->pin) + ctx->pin_half};
            size_t off = 0, n_cur = bytes < ctx->pin_half ? bytes : ctx->pin_half;
            int cur = 0;
            SX_HIP(hipMemcpyAsync(half[0], src, n_cur, hipMemcpyDeviceToHost, ctx->stream));
            while (off < bytes) {
                SX_HIP(hipStreamSynchronize(ctx->stream));
                const size_t next_off = off + n_cur;
                size_t n_next = 0;
                if (next_off < bytes) {
                    n_next = bytes - next_off < ctx->pin_half ? bytes - next_off : ctx->pin_half;
                    SX_HIP(hipMemcpyAsync(half[cur ^ 1], src + next_off, n_next, hipMemcpyDeviceToHost, ctx->stream));
                }
                memcpy(dst + off, half[cur], n_cur);
                off = next_off;
                n_cur = n_next;
                cur ^= 1;
            }
            return SX_OK;
        }
        (void)hipGetLastError(); // no pinned memory to be had: the plain copy below
    }
    SX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_memset(sx_ctx *ctx, void *dst_dev, int byte, size_t bytes) {
    SX_ENTER(ctx);
    if (bytes == 0) return SX_OK;
    SX_REQUIRE(dst_dev != nullptr, "NULL pointer in sx_memset");
    SX_HIP(hipMemsetAsync(dst_dev, byte, bytes, ctx->stream));
    return SX_OK;
}

SX_API int sx_timer_start(sx_ctx *ctx) {
    SX_ENTER(ctx);
    if (!ctx->t_made) {
        for (int i = 0; i < 8; ++i) {
            SX_HIP(hipEventCreate(&ctx->t0[i]));
            SX_HIP(hipEventCreate(&ctx->t1[i]));
        }
        ctx->t_made = true;
    }
    SX_REQUIRE(ctx->t_depth < 8, "timer nesting deeper than 8");
    SX_HIP(hipEventRecord(ctx->t0[ctx->t_depth], ctx->stream));
    ctx->t_depth++;
    return SX_OK;
}

SX_API int sx_timer_stop(sx_ctx *ctx, float *ms_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(ctx->t_depth > 0, "sx_timer_stop without sx_timer_start");
    ctx->t_depth--;
    int d = ctx->t_depth;
    SX_HIP(hipEventRecord(ctx->t1[d], ctx->stream));
    SX_HIP(hipEventSynchronize(ctx->t1[d]));
    float ms = 0.f;
    SX_HIP(hipEventElapsedTime(&ms, ctx->t0[d], ctx->t1[d]));
    if (ms_out) *ms_out = ms;
    return SX_OK;
}

SX_API int sx_marker_record(sx_ctx *ctx, int id) {
    SX_ENTER(ctx);
    SX_REQUIRE(id >= 0 && id < 4096, "marker id %d out of range [0,4096)", id);
    if (ctx->markers.size() <= static_cast<size_t>(id)) ctx->markers.resize(id + 1, nullptr);
    if (!ctx->markers[id]) SX_HIP(hipEventCreate(&ctx->markers[id]));
    SX_HIP(hipEventRecord(ctx->markers[id], ctx->stream));
    return SX_OK;
}

SX_API int sx_marker_elapsed(sx_ctx *ctx, int id_from, int id_to, float *ms_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(ms_out != nullptr, "ms_out is NULL");
    const int hi = static_cast<int>(ctx->markers.size());
    SX_REQUIRE(id_from >= 0 && id_from < hi && id_to >= 0 && id_to < hi && ctx->markers[id_from] &&
                   ctx->markers[id_to],
               "marker %d or %d was never recorded", id_from, id_to);
    SX_HIP(hipEventSynchronize(ctx->markers[id_to]));
    SX_HIP(hipEventElapsedTime(ms_out, ctx->markers[id_from], ctx->markers[id_to]));
    return SX_OK;
}

SX_API int sx_ctx_sync_device(sx_ctx *ctx) {
    SX_ENTER(ctx);
    SX_HIP(hipDeviceSynchronize());
    return SX_OK;
}

// ------------------------------------------------------------------------------------ matrix
template <class T>
static int upload_padded(sx_ctx *ctx, const T *host, int64_t count, T **dev_out) {
    *dev_out = nullptr;
    size_t bytes = static_cast<size_t>(count + SX_PAD) * sizeof(T);
    T *d = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d), bytes));
    *dev_out = d;
    SX_HIP(hipMemsetAsync(d + count, 0, SX_PAD * sizeof(T), ctx->stream));
    if (count)
        SX_HIP(hipMemcpyAsync(d, host, static_cast<size_t>(count) * sizeof(T), hipMemcpyHostToDevice,
                              ctx->stream));
    return SX_OK;
}

// Validation of a layout's offsets and inner indices.  The O(1) checks are done on the host before
// anything is uploaded; the O(n) / O(nnz) scans run on the device copies (first offending position by
// atomicMin) and are read back before any kernel uses the arrays as addresses.
constexpr unsigned long long SX_NO_BAD = ~0ull;

__global__ __launch_bounds__(256) void k_check_layout(const int64_t *__restrict__ ptr, int64_t nseg,
                                                      const int32_t *__restrict__ idx, int64_t nnz, int64_t bound,
                                                      unsigned long long *__restrict__ bad /* [0] ptr, [1] idx */) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long bp = SX_NO_BAD, bi = SX_NO_BAD;
    for (int64_t s = t0; s < nseg; s += stride)
        if (ptr[s + 1] < ptr[s] && static_cast<unsigned long long>(s) < bp) bp = static_cast<unsigned long long>(s);
    for (int64_t k = t0; k < nnz; k += stride) {
        const int32_t v = idx[k];
        if ((v < 0 || v >= bound) && static_cast<unsigned long long>(k) < bi) bi = static_cast<unsigned long long>(k);
    }
    if (bp != SX_NO_BAD) atomicMin(bad, bp);
    if (bi != SX_NO_BAD) atomicMin(bad + 1, bi);
}

static int check_ends(const int64_t *ptr, int64_t nseg, int64_t nnz, const char *what) {
    SX_REQUIRE(ptr[0] == 0, "%s[0] must be 0", what);
    SX_REQUIRE(ptr[nseg] == nnz, "%s[last] = %lld but nnz = %lld", what, (long long)ptr[nseg],
               (long long)nnz);
    return SX_OK;
}

// host_* are the caller's arrays (for the message), dev_* their uploaded copies
static int check_layout_dev(sx_ctx *ctx, const int64_t *dev_ptr, int64_t nseg, const int32_t *dev_idx,
                            const int32_t *host_idx, int64_t nnz, int64_t bound, const char *ptr_name,
                            const char *idx_name) {
    SX_TRY(sx_reserve(ctx, 2 * sizeof(unsigned long long)));
    unsigned long long *bad = static_cast<unsigned long long *>(ctx->ws);
    SX_HIP(hipMemsetAsync(bad, 0xff, 2 * sizeof(unsigned long long), ctx->stream));
    const int64_t work = nseg > nnz ? nseg : nnz;
    int64_t grid = (work + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_check_layout, dim3(static_cast<unsigned>(grid)), dim3(256), 0, ctx->stream, dev_ptr, nseg,
                       dev_idx, nnz, bound, bad);
    unsigned long long host_bad[2] = {SX_NO_BAD, SX_NO_BAD};
    SX_HIP(hipMemcpyAsync(host_bad, bad, sizeof(host_bad), hipMemcpyDeviceToHost, ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    SX_REQUIRE(host_bad[0] == SX_NO_BAD, "%s is not non-decreasing at %lld", ptr_name, (long long)host_bad[0]);
    SX_REQUIRE(host_bad[1] == SX_NO_BAD, "%s[%lld] = %d out of range [0,%lld)", idx_name, (long long)host_bad[1],
               host_idx[host_bad[1]], (long long)bound);
    return SX_OK;
}

SX_API int sx_matrix_create(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz,
                            const int64_t *csr_rowptr, const int32_t *csr_col,
                            const double *csr_val, const int64_t *csc_colptr,
                            const int32_t *csc_row, const double *csc_val, sx_matrix **out) {
    SX_ENTER(ctx);
    SX_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SX_REQUIRE(m >= 0 && n >= 0 && nnz >= 0, "negative dimension");
    SX_REQUIRE(m < INT32_MAX && n < INT32_MAX, "dimension exceeds int32 inner-index range");
    SX_REQUIRE(csr_rowptr != nullptr, "csr_rowptr is NULL");
    SX_REQUIRE(nnz == 0 || (csr_col && csr_val), "csr_col/csr_val is NULL");
    SX_TRY(check_ends(csr_rowptr, m, nnz, "csr_rowptr"));
    const bool have_csc = csc_colptr != nullptr;
    if (have_csc) {
        SX_REQUIRE(nnz == 0 || (csc_row && csc_val), "csc_row/csc_val is NULL");
        SX_TRY(check_ends(csc_colptr, n, nnz, "csc_colptr"));
    }
    sx_matrix *A = new (std::nothrow) sx_matrix();
    if (!A) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    A->ctx = ctx;
    A->m = m;
    A->n = n;
    A->nnz = nnz;
    int rc = SX_OK;
    if ((rc = upload_padded(ctx, csr_rowptr, m + 1, &A->csr_ptr)) == SX_OK &&
        (rc = upload_padded(ctx, csr_col, nnz, &A->csr_idx)) == SX_OK &&
        (rc = upload_padded(ctx, csr_val, nnz, &A->csr_val)) == SX_OK &&
        (rc = check_layout_dev(ctx, A->csr_ptr, m, A->csr_idx, csr_col, nnz, n, "csr_rowptr", "csr_col")) == SX_OK) {
        if (have_csc) {
            if ((rc = upload_padded(ctx, csc_colptr, n + 1, &A->csc_ptr)) == SX_OK &&
                (rc = upload_padded(ctx, csc_row, nnz, &A->csc_idx)) == SX_OK &&
                (rc = upload_padded(ctx, csc_val, nnz, &A->csc_val)) == SX_OK)
                rc = check_layout_dev(ctx, A->csc_ptr, n, A->csc_idx, csc_row, nnz, m, "csc_colptr", "csc_row");
        } else {
            // stable transposition on the device (sx_transpose.hip): per-column entries in row-major walk order
            rc = sx_transpose_dev(ctx, m, n, nnz, A->csr_ptr, A->csr_idx, A->csr_val, &A->csc_ptr, &A->csc_idx,
                                  &A->csc_val);
        }
        if (rc == SX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
            sx_set_error("stream sync failed after matrix upload");
            rc = SX_ERR_HIP;
        }
    }
    if (rc == SX_OK) rc = sx_build_tiles(ctx, A->csr_ptr, m, &A->csr_tiles, &A->n_csr_tiles, &A->csr_imbalance);
    if (rc == SX_OK) rc = sx_build_tiles(ctx, A->csc_ptr, n, &A->csc_tiles, &A->n_csc_tiles, &A->csc_imbalance);
    if (rc != SX_OK) {
        sx_matrix_destroy(A);
        return rc;
    }
    *out = A;
    return SX_OK;
}

SX_API int sx_matrix_create_single(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int is_csc,
                                   const int64_t *ptr, const int32_t *idx, const double *val,
                                   sx_matrix **out) {
    SX_ENTER(ctx);
    SX_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    SX_REQUIRE(m >= 0 && n >= 0 && nnz >= 0, "negative dimension");
    SX_REQUIRE(m < INT32_MAX && n < INT32_MAX, "dimension exceeds int32 inner-index range");
    SX_REQUIRE(ptr != nullptr, "ptr is NULL");
    SX_REQUIRE(nnz == 0 || (idx && val), "idx/val is NULL");
    const int64_t nseg = is_csc ? n : m, bound = is_csc ? m : n;
    SX_TRY(check_ends(ptr, nseg, nnz, is_csc ? "colptr" : "rowptr"));
    sx_matrix *A = new (std::nothrow) sx_matrix();
    if (!A) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    A->ctx = ctx;
    A->m = m;
    A->n = n;
    A->nnz = nnz;
    int rc;
    if (is_csc) {
        (rc = upload_padded(ctx, ptr, n + 1, &A->csc_ptr)) == SX_OK &&
            (rc = upload_padded(ctx, idx, nnz, &A->csc_idx)) == SX_OK &&
            (rc = upload_padded(ctx, val, nnz, &A->csc_val));
    } else {
        (rc = upload_padded(ctx, ptr, m + 1, &A->csr_ptr)) == SX_OK &&
            (rc = upload_padded(ctx, idx, nnz, &A->csr_idx)) == SX_OK &&
            (rc = upload_padded(ctx, val, nnz, &A->csr_val));
    }
    if (rc == SX_OK) // synchronises: the uploads are complete afterwards
        rc = check_layout_dev(ctx, is_csc ? A->csc_ptr : A->csr_ptr, nseg, is_csc ? A->csc_idx : A->csr_idx, idx, nnz,
                              bound, is_csc ? "colptr" : "rowptr", is_csc ? "row index" : "column index");
    if (rc == SX_OK)
        rc = is_csc ? sx_build_tiles(ctx, A->csc_ptr, n, &A->csc_tiles, &A->n_csc_tiles, &A->csc_imbalance)
                    : sx_build_tiles(ctx, A->csr_ptr, m, &A->csr_tiles, &A->n_csr_tiles, &A->csr_imbalance);
    if (rc != SX_OK) {
        sx_matrix_destroy(A);
        return rc;
    }
    *out = A;
    return SX_OK;
}

SX_API int sx_matrix_destroy(sx_matrix *A) {
    if (!A) return SX_OK;
    sx_device_guard guard(A->ctx->device);
    (void)hipStreamSynchronize(A->ctx->stream);
    void *ptrs[9] = {A->csr_ptr, A->csr_idx,   A->csr_val,   A->csc_ptr,   A->csc_idx,
                     A->csc_val, A->csr_tiles, A->csc_tiles, A->csc_win_lo};
    for (void *p : ptrs)
        if (p) (void)sx_dfree(p);
    sx_rowblock_free(A->rb);
    sx_slabs_free(A->slabs[0]);
    sx_slabs_free(A->slabs[1]);
    delete A;
    return SX_OK;
}

SX_API int sx_matrix_dims(const sx_matrix *A, int64_t *m, int64_t *n, int64_t *nnz) {
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    if (m) *m = A->m;
    if (n) *n = A->n;
    if (nnz) *nnz = A->nnz;
    return SX_OK;
}

SX_API int sx_matrix_arrays(const sx_matrix *A, const int64_t **csr_rowptr, const int32_t **csr_col,
                            const double **csr_val, const int64_t **csc_colptr,
                            const int32_t **csc_row, const double **csc_val) {
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    if (csr_rowptr) *csr_rowptr = A->csr_ptr;
    if (csr_col) *csr_col = A->csr_idx;
    if (csr_val) *csr_val = A->csr_val;
    if (csc_colptr) *csc_colptr = A->csc_ptr;
    if (csc_row) *csc_row = A->csc_idx;
    if (csc_val) *csc_val = A->csc_val;
    return SX_OK;
}

SX_API int sx_matrix_download_csr(const sx_matrix *A, int64_t *rowptr, int32_t *col, double *val) {
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(A->csr_ptr != nullptr, "matrix has no CSR layout");
    sx_ctx *ctx = A->ctx;
    SX_ENTER(ctx);
    if (rowptr)
        SX_HIP(hipMemcpyAsync(rowptr, A->csr_ptr, sizeof(int64_t) * (A->m + 1), hipMemcpyDeviceToHost,
                              ctx->stream));
    if (col && A->nnz)
        SX_HIP(hipMemcpyAsync(col, A->csr_idx, sizeof(int32_t) * A->nnz, hipMemcpyDeviceToHost,
                              ctx->stream));
    if (val && A->nnz)
        SX_HIP(hipMemcpyAsync(val, A->csr_val, sizeof(double) * A->nnz, hipMemcpyDeviceToHost,
                              ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}
