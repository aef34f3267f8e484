// Tile table of a pointer array (see sx_segwalk.h): tile boundaries sit at every 256th segment
// and wherever the entry offset, counted from the enclosing 256-segment block, crosses a multiple
// of SX_TILE_BUDGET.  The predicate is local to a segment, so the table is built with one flag
// kernel and the same stream compaction that serves np.where (sx_select_indices_dev).
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

__global__ __launch_bounds__(SX_WG) void k_tile_flags(const int64_t *__restrict__ ptr, int64_t nseg,
                                                      int64_t budget, uint8_t *__restrict__ flag) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (s >= nseg) return;
    const int64_t sb = s - (s % SX_WG);
    bool cut = (s == sb);
    if (!cut) {
        const int64_t origin = ptr[sb];
        cut = ((ptr[s] - origin) / budget) != ((ptr[s - 1] - origin) / budget);
    }
    flag[s] = cut ? 1 : 0;
}

// serial cost of the eight XCD-contiguous tile ranges (the map of sx_tile_of_block): cost[k] += avg segment length + 64
__global__ __launch_bounds__(SX_WG) void k_tile_range_cost(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                           const int64_t *__restrict__ ptr, unsigned long long *__restrict__ cost) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (t >= ntiles) return;
    const int64_t s0 = tiles[t], s1 = tiles[t + 1];
    const int64_t per = (ntiles + 7) >> 3;
    const int64_t segs = s1 > s0 ? s1 - s0 : 1;
    atomicAdd(&cost[t / per], static_cast<unsigned long long>((ptr[s1] - ptr[s0]) / segs + 64));
}

// Entry budget of a tile: SX_TILE_BUDGET for large matrices; halved (down to 1024) while the matrix
// would otherwise yield fewer than ~2048 tiles, so that cache-resident problems still put several
// workgroups on each of the 256 CUs (config 2: 250 row tiles -> 2000).
inline int64_t tile_budget(int64_t nnz) {
    int64_t budget = SX_TILE_BUDGET;
    while (budget > 1024 && nnz / budget < 2048) budget >>= 1;
    return budget;
}

} // namespace

int sx_build_tiles(sx_ctx *ctx, const int64_t *ptr_dev, int64_t nseg, int64_t **tiles_out,
                   int64_t *ntiles_out, double *imbalance_out) {
    *tiles_out = nullptr;
    *ntiles_out = 0;
    if (imbalance_out) *imbalance_out = 1.0;
    int64_t *tiles = nullptr;
    if (nseg == 0) {
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&tiles), sizeof(int64_t)));
        SX_HIP(hipMemsetAsync(tiles, 0, sizeof(int64_t), ctx->stream));
        *tiles_out = tiles;
        return SX_OK;
    }
    uint8_t *flag = nullptr;
    int64_t *idx = nullptr, *count = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&flag), static_cast<size_t>(nseg)));
    int rc = SX_OK;
    hipError_t e;
    if ((e = sx_dmalloc(reinterpret_cast<void **>(&idx), sizeof(int64_t) * nseg)) != hipSuccess ||
        (e = sx_dmalloc(reinterpret_cast<void **>(&count), sizeof(int64_t))) != hipSuccess) {
        sx_set_error("hipMalloc failed while building tiles: %s", hipGetErrorString(e));
        rc = SX_ERR_NOMEM;
    }
    int64_t nt = 0, nnz = 0;
    if (rc == SX_OK &&
        (hipMemcpyAsync(&nnz, ptr_dev + nseg, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
         hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        sx_set_error("nnz download failed while building tiles");
        rc = SX_ERR_HIP;
    }
    if (rc == SX_OK) {
        const unsigned nb = static_cast<unsigned>((nseg + SX_WG - 1) / SX_WG);
        hipLaunchKernelGGL(k_tile_flags, dim3(nb), dim3(SX_WG), 0, ctx->stream, ptr_dev, nseg, tile_budget(nnz), flag);
        rc = sx_select_indices_dev(ctx, nseg, flag, 1, idx, count);
    }
    if (rc == SX_OK) {
        if (hipMemcpyAsync(&nt, count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            sx_set_error("tile count download failed");
            rc = SX_ERR_HIP;
        }
    }
    if (rc == SX_OK) {
        if (sx_dmalloc(reinterpret_cast<void **>(&tiles), sizeof(int64_t) * (nt + 1)) != hipSuccess) {
            sx_set_error("hipMalloc failed for %lld tiles", (long long)nt);
            rc = SX_ERR_NOMEM;
        } else if (hipMemcpyAsync(tiles, idx, sizeof(int64_t) * nt, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                   hipMemcpyAsync(tiles + nt, &nseg, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                   hipStreamSynchronize(ctx->stream) != hipSuccess) {
            sx_set_error("tile table copy failed");
            rc = SX_ERR_HIP;
        }
    }
    if (rc == SX_OK && imbalance_out && nt >= 64) { // (reuses `idx`: 8 counters)
        unsigned long long cost[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long *d = reinterpret_cast<unsigned long long *>(idx);
        if (hipMemsetAsync(d, 0, sizeof(cost), ctx->stream) == hipSuccess) {
            hipLaunchKernelGGL(k_tile_range_cost, dim3(static_cast<unsigned>((nt + SX_WG - 1) / SX_WG)), dim3(SX_WG), 0, ctx->stream,
                               tiles, nt, ptr_dev, d);
            if (hipMemcpyAsync(cost, d, sizeof(cost), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess) {
                double sum = 0.0, mx = 0.0;
                for (unsigned long long c : cost) {
                    sum += static_cast<double>(c);
                    mx = mx > static_cast<double>(c) ? mx : static_cast<double>(c);
                }
                if (sum > 0.0) *imbalance_out = mx / (sum / 8.0);
            }
        }
    }
    if (flag) (void)sx_dfree(flag);
    if (idx) (void)sx_dfree(idx);
    if (count) (void)sx_dfree(count);
    if (rc != SX_OK) {
        if (tiles) (void)sx_dfree(tiles);
        return rc;
    }
    *tiles_out = tiles;
    *ntiles_out = nt;
    return SX_OK;
}
