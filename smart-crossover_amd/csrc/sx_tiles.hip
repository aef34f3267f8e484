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
                   int64_t *ntiles_out) {
    *tiles_out = nullptr;
    *ntiles_out = 0;
    int64_t *tiles = nullptr;
    if (nseg == 0) {
        SX_HIP(hipMalloc(reinterpret_cast<void **>(&tiles), sizeof(int64_t)));
        SX_HIP(hipMemsetAsync(tiles, 0, sizeof(int64_t), ctx->stream));
        *tiles_out = tiles;
        return SX_OK;
    }
    uint8_t *flag = nullptr;
    int64_t *idx = nullptr, *count = nullptr;
    SX_HIP(hipMalloc(reinterpret_cast<void **>(&flag), static_cast<size_t>(nseg)));
    int rc = SX_OK;
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void **>(&idx), sizeof(int64_t) * nseg)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void **>(&count), sizeof(int64_t))) != hipSuccess) {
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
        if (hipMalloc(reinterpret_cast<void **>(&tiles), sizeof(int64_t) * (nt + 1)) != hipSuccess) {
            sx_set_error("hipMalloc failed for %lld tiles", (long long)nt);
            rc = SX_ERR_NOMEM;
        } else if (hipMemcpyAsync(tiles, idx, sizeof(int64_t) * nt, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                   hipMemcpyAsync(tiles + nt, &nseg, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                   hipStreamSynchronize(ctx->stream) != hipSuccess) {
            sx_set_error("tile table copy failed");
            rc = SX_ERR_HIP;
        }
    }
    if (flag) (void)hipFree(flag);
    if (idx) (void)hipFree(idx);
    if (count) (void)hipFree(count);
    if (rc != SX_OK) {
        if (tiles) (void)hipFree(tiles);
        return rc;
    }
    *tiles_out = tiles;
    *ntiles_out = nt;
    return SX_OK;
}
