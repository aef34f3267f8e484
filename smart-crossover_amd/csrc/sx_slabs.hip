// Builder of the operand slabs (sx_slabs.h): counts per (slab, segment), one scan per slab, an order-preserving
// scatter, a tile table per slab.  One-off per matrix and walk direction.
#include "sx_internal.h"
#include "sx_segwalk.h"
#include "sx_slabs.h"

#include <new>
#include <vector>

namespace {

struct SlabDev {
    int64_t *ptr;
    int32_t *idx;
    double *val;
};

// one lane per segment: entries per slab; a segment whose indices descend raises *descending
__global__ __launch_bounds__(SX_WG) void k_slab_count(int64_t nseg, const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                      int64_t width, int R, int64_t *__restrict__ cnt, int *__restrict__ descending) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (j >= nseg) return;
    const int64_t e0 = ptr[j], e1 = ptr[j + 1];
    int s = 0;
    int64_t run = 0;
    int32_t prev = -1;
    for (int64_t e = e0; e < e1; ++e) {
        const int32_t i = idx[e];
        if (i < prev) *descending = 1; // benign race: every writer stores 1
        prev = i;
        int si = static_cast<int>(i / width);
        si = si < R ? si : R - 1; // (an index beyond the operand: the walk would fault on it as well; keep the table whole)
        if (si != s) {
            if (si > s) { // (si < s only on a descending segment: refused anyway)
                cnt[static_cast<int64_t>(s) * nseg + j] = run;
                for (int t = s + 1; t < si; ++t) cnt[static_cast<int64_t>(t) * nseg + j] = 0;
                s = si;
                run = 0;
            }
        }
        ++run;
    }
    cnt[static_cast<int64_t>(s) * nseg + j] = run;
    for (int t = s + 1; t < R; ++t) cnt[static_cast<int64_t>(t) * nseg + j] = 0;
}

// one lane per segment: its entries to their slabs, stored order kept
__global__ __launch_bounds__(SX_WG) void k_slab_scatter(int64_t nseg, const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                        const double *__restrict__ val, int64_t width, int R,
                                                        const SlabDev *__restrict__ slabs) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (j >= nseg) return;
    const int64_t e0 = ptr[j], e1 = ptr[j + 1];
    int s = -1;
    int64_t at = 0;
    SlabDev d{nullptr, nullptr, nullptr};
    for (int64_t e = e0; e < e1; ++e) {
        const int32_t i = idx[e];
        int si = static_cast<int>(i / width);
        si = si < R ? si : R - 1;
        if (si != s) {
            s = si;
            d = slabs[s];
            at = d.ptr[j];
        }
        d.idx[at] = static_cast<int32_t>(i - static_cast<int64_t>(s) * width);
        d.val[at] = val[e];
        ++at;
    }
}

inline unsigned grid1d(int64_t n) { return static_cast<unsigned>((n + SX_WG - 1) / SX_WG); }

int build(sx_ctx *ctx, const int64_t *ptr, const int32_t *idx, const double *val, int64_t nseg, int64_t bound, int64_t nnz,
          int R, sx_slabs **out) {
    *out = nullptr;
    hipStream_t st = ctx->stream;
    int64_t width = (bound + R - 1) / R;
    width = (width + 15) & ~static_cast<int64_t>(15); // slab operands start on a 128-byte line
    R = static_cast<int>((bound + width - 1) / width);
    if (R < 2) return SX_OK;
    sx_slabs *S = new (std::nothrow) sx_slabs();
    if (!S) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    struct Guard {
        sx_slabs *S;
        void *tmp[3] = {nullptr, nullptr, nullptr};
        ~Guard() {
            for (void *p : tmp)
                if (p) (void)sx_dfree(p);
            if (S) sx_slabs_free(S);
        }
    } guard{S};
    S->R = R;
    S->nseg = nseg;
    S->width = width;
    S->slab = new (std::nothrow) sx_slab[R];
    if (!S->slab) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    int64_t *cnt = nullptr;
    int *desc = nullptr;
    SlabDev *dev = nullptr;
    SX_HIP(sx_dmalloc(&guard.tmp[0], sizeof(int64_t) * static_cast<size_t>(R) * static_cast<size_t>(nseg)));
    SX_HIP(sx_dmalloc(&guard.tmp[1], sizeof(int)));
    SX_HIP(sx_dmalloc(&guard.tmp[2], sizeof(SlabDev) * static_cast<size_t>(R)));
    cnt = static_cast<int64_t *>(guard.tmp[0]);
    desc = static_cast<int *>(guard.tmp[1]);
    dev = static_cast<SlabDev *>(guard.tmp[2]);
    SX_HIP(hipMemsetAsync(desc, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_slab_count, dim3(grid1d(nseg)), dim3(SX_WG), 0, st, nseg, ptr, idx, width, R, cnt, desc);
    SX_HIP(hipGetLastError());
    int descending = 0;
    SX_HIP(hipMemcpyAsync(&descending, desc, sizeof(int), hipMemcpyDeviceToHost, st));
    SX_HIP(hipStreamSynchronize(st));
    if (descending) return SX_OK; // stored order is not index order: the plain walk keeps the sums' order
    std::vector<SlabDev> host(static_cast<size_t>(R));
    for (int s = 0; s < R; ++s) {
        sx_slab &L = S->slab[s];
        L.off = static_cast<int64_t>(s) * width;
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&L.ptr), sizeof(int64_t) * static_cast<size_t>(nseg + 9)));
        SX_HIP(hipMemsetAsync(L.ptr, 0, sizeof(int64_t) * static_cast<size_t>(nseg + 9), st));
        SX_TRY(sx_scan_exclusive(ctx, cnt + static_cast<int64_t>(s) * nseg, nseg, L.ptr));
        SX_HIP(hipMemcpyAsync(&L.nnz, L.ptr + nseg, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        SX_HIP(hipStreamSynchronize(st));
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&L.idx), sizeof(int32_t) * static_cast<size_t>(L.nnz + 8)));
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&L.val), sizeof(double) * static_cast<size_t>(L.nnz + 8)));
        SX_HIP(hipMemsetAsync(L.idx, 0, sizeof(int32_t) * static_cast<size_t>(L.nnz + 8), st));
        SX_HIP(hipMemsetAsync(L.val, 0, sizeof(double) * static_cast<size_t>(L.nnz + 8), st));
        host[static_cast<size_t>(s)] = SlabDev{L.ptr, L.idx, L.val};
    }
    SX_HIP(hipMemcpyAsync(dev, host.data(), sizeof(SlabDev) * static_cast<size_t>(R), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_slab_scatter, dim3(grid1d(nseg)), dim3(SX_WG), 0, st, nseg, ptr, idx, val, width, R, dev);
    SX_HIP(hipGetLastError());
    SX_HIP(hipStreamSynchronize(st)); // `host` leaves scope
    int64_t total = 0;
    for (int s = 0; s < R; ++s) {
        SX_TRY(sx_build_tiles(ctx, S->slab[s].ptr, nseg, &S->slab[s].tiles, &S->slab[s].ntiles));
        total += S->slab[s].nnz;
    }
    if (total != nnz) {
        sx_set_error("operand slabs hold %lld of %lld entries", (long long)total, (long long)nnz);
        return SX_ERR_HIP;
    }
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&S->carry), sizeof(double) * static_cast<size_t>(nseg + 8)));
    SX_HIP(hipStreamSynchronize(st));
    guard.S = nullptr;
    *out = S;
    return SX_OK;
}

} // namespace

void sx_slabs_free(sx_slabs *S) {
    if (!S) return;
    if (S->slab) {
        for (int s = 0; s < S->R; ++s) {
            void *p[4] = {S->slab[s].ptr, S->slab[s].idx, S->slab[s].val, S->slab[s].tiles};
            for (void *q : p)
                if (q) (void)sx_dfree(q);
        }
        delete[] S->slab;
    }
    if (S->carry) (void)sx_dfree(S->carry);
    delete S;
}

int sx_slabs_get(sx_ctx *ctx, const sx_matrix *A, int which, const sx_slabs **out) {
    *out = nullptr;
    const int opt = ctx->opt_slabs;
    if (opt == 0) return SX_OK;
    const int64_t *ptr = which ? A->csc_ptr : A->csr_ptr;
    if (!ptr || A->nnz == 0) return SX_OK;
    const int64_t nseg = which ? A->n : A->m, bound = which ? A->m : A->n;
    int R = opt;
    if (opt < 0) { // automatic: the operand must not fit an XCD's L2, and the carries must not outweigh the stream
        R = 0;
        const int64_t operand = 8 * bound;
        constexpr int64_t SLICE = 3200000; // bytes of operand per slab: measured optimum at config-5 size (kbench_slabs*.txt:
                                           // K1 8 MB / 3, K2 80 MB / 24..28; 4 MB slices no longer stay in a 4 MiB L2)
        // (a column walk whose tiles gather from a narrow range of rows has locality whatever the size of y: the LDS window
        //  may not pay for it -- sx_window.hip's rule, or the timing of window_autotune -- but slabs are for walks WITHOUT it:
        //  netlib_lp at config-5 size, plain walk 0.26 ms, three slabs 0.69 ms)
        bool local = false;
        if (which == 1) {
            int run = 0;
            SX_TRY(sx_window_run_csc(ctx, A, &run)); // (builds the sampling table on first use)
            local = A->csc_win_local != 0;
        }
        if (!local && operand > SLICE + SLICE / 8 && A->nnz >= (1 << 22)) {
            const int64_t want = (operand + SLICE - 1) / SLICE;
            if (want <= 256 && 24 * nseg * want * 4 <= 12 * A->nnz * 5) R = static_cast<int>(want);
        }
    }
    if (R < 2 || bound < 16 * R) return SX_OK;
    if (A->slabs_tried[which] != R) { // (a layout built for another R is replaced; a refusal is remembered as well)
        sx_slabs_free(A->slabs[which]);
        A->slabs[which] = nullptr;
        A->slabs_tried[which] = R;
        sx_slabs *S = nullptr;
        SX_TRY(build(ctx, ptr, which ? A->csc_idx : A->csr_idx, which ? A->csc_val : A->csr_val, nseg, bound, A->nnz, R, &S));
        A->slabs[which] = S;
    }
    *out = A->slabs[which];
    return SX_OK;
}

// ------------------------------------------------------------------ introspection (tests, tools, bench.py)
SX_API int sx_matrix_slabs_info(sx_ctx *ctx, const sx_matrix *A, int which, int64_t *info /* [3] */) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr && info != nullptr, "NULL argument");
    SX_REQUIRE(which == 0 || which == 1, "which must be 0 (rows) or 1 (columns)");
    const sx_slabs *S = nullptr;
    SX_TRY(sx_slabs_get(ctx, A, which, &S));
    info[0] = S ? S->R : 0;
    info[1] = S ? S->width : 0;
    info[2] = S ? S->nseg : 0;
    return SX_OK;
}
