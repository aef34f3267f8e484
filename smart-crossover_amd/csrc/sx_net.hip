// Network-side scoring kernels (gfx950):
//   K7  MCF flow indicators   (MCFManagerStd.get_sorted_flows, network_methods/net_manager.py:165-182)
//   K8  OT flow indicators    (OTManager.get_sorted_flows,     network_methods/net_manager.py:377-378)
//   OT pricing on the implicit incidence structure (net_manager.py:483,496 with formats.py:156-159)
// Built with -ffp-contract=off; every product / quotient / sum is a separately rounded operation in
// the order the reference's numpy expression evaluates it.
#include "sx_internal.h"
#include "sx_segwalk.h"

#include <cmath>

namespace {

__device__ __forceinline__ double np_maximum(double a, double b) {
    return (a > b || a != a) ? a : b; // numpy.maximum: NaN wins
}

// ------------------------------------------------------------------------------------- K7 (a)
// x_hat = x*(~mask) + u*mask - x*mask with mask = x > u/2, zeroed outside [0, u]
// (net_manager.py:166-168; the three-term form is kept: it yields NaN for u = inf like numpy does)
__global__ __launch_bounds__(SX_WG) void k_mcf_xhat(int64_t E, const double *__restrict__ x,
                                                    const double *__restrict__ u,
                                                    double *__restrict__ xhat,
                                                    uint8_t *__restrict__ mask) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < E;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double xj = x[j], uj = u[j];
        const bool big = xj > uj / 2;
        const double keep = big ? 0.0 : 1.0, flip = big ? 1.0 : 0.0;
        double xh = (xj * keep + uj * flip) - xj * flip;
        if (xj < 0 || xj > uj) xh = 0.0;
        xhat[j] = xh;
        mask[j] = big ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------- K7 (b)
// per node: f1 = sum over a_bar > 0 of a_bar*x_hat, f2 = sum over a_bar < 0 of (-a_bar)*x_hat, each
// in ascending arc order (net_manager.py:169-176); f_inv = 1/max(f1,f2) where that is non-zero
struct StageThroughput {
    const double *__restrict__ xhat;
    const uint8_t *__restrict__ mask;
    __device__ __forceinline__ void operator()(double v, int32_t j, double (&o)[2]) const {
        const double abar = mask[j] ? -v : v;
        const double xh = xhat[j];
        o[0] = (abar > 0) ? abar * xh : 0.0;
        o[1] = (abar < 0) ? (-abar) * xh : 0.0;
    }
};

__global__ __launch_bounds__(SX_WG) void k_mcf_throughput(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                          int swizzle, const int64_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ colidx,
                                                          const double *__restrict__ val,
                                                          const double *__restrict__ xhat,
                                                          const uint8_t *__restrict__ mask,
                                                          double *__restrict__ f_inv,
                                                          double *__restrict__ f_out) {
    __shared__ sx_walk_lds<2, 2048> lds;
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    double acc[2];
    int64_t i;
    bool valid;
    sx_segwalk<2, 2048>(tiles, tile, rowptr, colidx, val, StageThroughput{xhat, mask}, lds, i, valid, acc);
    if (!valid) return;
    const double f = np_maximum(acc[0], acc[1]);
    if (f_out) f_out[i] = f;
    f_inv[i] = (f != 0) ? 1 / f : 0.0;
}

// ------------------------------------------------------------------------------------- K7 (c)
// ind_j = max over the arc's entries of | (f_inv[i] * x_hat[j]) * a_bar_ij |  (net_manager.py:178-182)
// one lane per arc: incidence columns hold two entries, max is exact so no ordering is needed
__global__ __launch_bounds__(SX_WG) void k_mcf_indicator(int64_t E, const int64_t *__restrict__ colptr,
                                                         const int32_t *__restrict__ rowidx,
                                                         const double *__restrict__ val,
                                                         const double *__restrict__ xhat,
                                                         const uint8_t *__restrict__ mask,
                                                         const double *__restrict__ f_inv,
                                                         double *__restrict__ ind) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < E;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double xh = xhat[j];
        const bool big = mask[j] != 0;
        double best = 0.0;
        for (int64_t e = colptr[j]; e < colptr[j + 1]; ++e) {
            const double a = val[e];
            const double abar = big ? -a : a;
            if (abar == 0) continue; // explicit zeros are not entries of A_bar
            const double r = fabs((f_inv[rowidx[e]] * xh) * abar);
            best = np_maximum(r, best);
        }
        ind[j] = best;
    }
}

// ------------------------------------------------------------------------------------- K8
__global__ __launch_bounds__(SX_WG) void k_ot_indicator(int64_t S, int64_t D, const double *__restrict__ X,
                                                        const double *__restrict__ s,
                                                        const double *__restrict__ d,
                                                        double *__restrict__ ind) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t i = e / D, j = e - i * D;
        const double xe = X[e];
        ind[e] = np_maximum(xe / s[i], xe / d[j]);
    }
}

// ------------------------------------------------------------------------------------- OT pricing
// rc_ij = M_ij - ((0 + (-1)*y_i) + (+1)*y_{S+j});  all(rc >= -tol) and the most negative one
struct OtPartial {
    double min_rc;
    long long argmin;
    long long n_bad;
};

__device__ __forceinline__ void ot_combine(double &v, long long &ix, double v2, long long ix2) {
    if (ix2 >= 0 && (ix < 0 || v2 < v || (v2 == v && ix2 < ix))) {
        v = v2;
        ix = ix2;
    }
}

__device__ __forceinline__ void ot_block_reduce(double v, long long ix, long long bad, OtPartial *slot) {
    __shared__ double sv[SX_WG / 64];
    __shared__ long long si[SX_WG / 64];
    __shared__ long long sb[SX_WG / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double v2 = __shfl_down(v, o, 64);
        long long i2 = __shfl_down(ix, o, 64);
        ot_combine(v, ix, v2, i2);
        bad += __shfl_down(bad, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = v;
        si[threadIdx.x >> 6] = ix;
        sb[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SX_WG / 64; ++w) {
            ot_combine(v, ix, sv[w], si[w]);
            bad += sb[w];
        }
        slot->min_rc = v;
        slot->argmin = ix;
        slot->n_bad = bad;
    }
}

__global__ __launch_bounds__(SX_WG) void k_ot_price(int64_t S, int64_t D, const double *__restrict__ M,
                                                    const double *__restrict__ y, double tol,
                                                    double *__restrict__ rc_out,
                                                    OtPartial *__restrict__ partial) {
    const int64_t n = S * D;
    double v = 0.0;
    long long ix = -1, bad = 0;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t i = e / D, j = e - i * D;
        const double aty = (0.0 + (-1.0) * y[i]) + (1.0) * y[S + j];
        const double rc = M[e] - aty;
        if (rc_out) rc_out[e] = rc;
        bad += (rc >= -tol) ? 0 : 1;
        if (rc == rc) ot_combine(v, ix, rc, e);
    }
    ot_block_reduce(v, ix, bad, &partial[blockIdx.x]);
}

__global__ __launch_bounds__(SX_WG) void k_ot_price_final(const OtPartial *__restrict__ partial, int np,
                                                          sx_price_result *out) {
    double v = 0.0;
    long long ix = -1, bad = 0;
    for (int b = threadIdx.x; b < np; b += SX_WG) {
        ot_combine(v, ix, partial[b].min_rc, partial[b].argmin);
        bad += partial[b].n_bad;
    }
    __shared__ OtPartial one;
    ot_block_reduce(v, ix, bad, &one);
    __syncthreads();
    if (threadIdx.x == 0) {
        out->min_rc = (one.argmin >= 0) ? one.min_rc : NAN;
        out->argmin = one.argmin;
        out->n_violating = one.n_bad;
    }
}

inline unsigned grid1d(int64_t n, int64_t cap = 8192) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

} // namespace

SX_API int sx_flow_indicator_mcf_dev(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *u,
                                     double *ind, double *xhat_out, double *f_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && x && u && ind, "NULL argument");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the MCF indicator needs both layouts of A");
    const int64_t V = A->m, E = A->n;
    if (E == 0) return SX_OK;
    // scratch: xhat[E] (or caller's), mask[E], f_inv[V]
    const size_t need = sizeof(double) * (static_cast<size_t>(E) + static_cast<size_t>(V)) + static_cast<size_t>(E) + 512;
    SX_TRY(sx_reserve(ctx, need));
    char *base = static_cast<char *>(ctx->ws);
    double *xhat = xhat_out ? xhat_out : reinterpret_cast<double *>(base);
    double *f_inv = reinterpret_cast<double *>(base) + E;
    uint8_t *mask = reinterpret_cast<uint8_t *>(f_inv + V);
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_mcf_xhat, dim3(grid1d(E)), dim3(SX_WG), 0, s, E, x, u, xhat, mask);
    if (V > 0) {
        const int swz = ctx->opt_xcd_swizzle;
        const unsigned grid = swz ? static_cast<unsigned>(((A->n_csr_tiles + 7) >> 3) << 3)
                                  : static_cast<unsigned>(A->n_csr_tiles);
        hipLaunchKernelGGL(k_mcf_throughput, dim3(grid), dim3(SX_WG), 0, s, A->csr_tiles, A->n_csr_tiles, swz,
                           A->csr_ptr, A->csr_idx, A->csr_val, xhat, mask, f_inv, f_out);
    }
    hipLaunchKernelGGL(k_mcf_indicator, dim3(grid1d(E)), dim3(SX_WG), 0, s, E, A->csc_ptr, A->csc_idx, A->csc_val,
                       xhat, mask, f_inv, ind);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

// The three steps of K7 one by one, for arcs sharded over ranks (BASELINE config 4): a rank owns a
// block of arcs (x, u, its incidence columns) and a block of nodes (their rows over ALL arcs).  x_hat of the own
// arcs -> all-gather -> node throughputs of the own nodes from the full x_hat (each node sums its arcs in
// ascending order: the sums of the single-process kernel, bit for bit) -> all-gather of f_inv -> indicators of
// the own arcs.  No floating-point value is ever added across ranks.
SX_API int sx_mcf_xhat_dev(sx_ctx *ctx, int64_t E, const double *x, const double *u, double *xhat, uint8_t *mask) {
    SX_ENTER(ctx);
    SX_REQUIRE(E >= 0, "E < 0");
    if (E == 0) return SX_OK;
    SX_REQUIRE(x && u && xhat && mask, "NULL argument");
    hipLaunchKernelGGL(k_mcf_xhat, dim3(grid1d(E)), dim3(SX_WG), 0, ctx->stream, E, x, u, xhat, mask);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_mcf_node_flows_dev(sx_ctx *ctx, const sx_matrix *A_rows, const double *xhat_all, const uint8_t *mask_all,
                                 double *f_inv, double *f_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(A_rows && xhat_all && mask_all && f_inv, "NULL argument");
    SX_REQUIRE(A_rows->csr_ptr != nullptr, "the node block needs the row layout");
    if (A_rows->m == 0) return SX_OK;
    const int swz = ctx->opt_xcd_swizzle;
    const unsigned grid = swz ? static_cast<unsigned>(((A_rows->n_csr_tiles + 7) >> 3) << 3)
                              : static_cast<unsigned>(A_rows->n_csr_tiles);
    hipLaunchKernelGGL(k_mcf_throughput, dim3(grid), dim3(SX_WG), 0, ctx->stream, A_rows->csr_tiles, A_rows->n_csr_tiles,
                       swz, A_rows->csr_ptr, A_rows->csr_idx, A_rows->csr_val, xhat_all, mask_all, f_inv, f_out);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_mcf_arc_indicator_dev(sx_ctx *ctx, const sx_matrix *A_cols, const double *xhat_loc, const uint8_t *mask_loc,
                                    const double *f_inv_all, double *ind) {
    SX_ENTER(ctx);
    SX_REQUIRE(A_cols && xhat_loc && mask_loc && f_inv_all && ind, "NULL argument");
    SX_REQUIRE(A_cols->csc_ptr != nullptr, "the arc block needs the column layout");
    if (A_cols->n == 0) return SX_OK;
    hipLaunchKernelGGL(k_mcf_indicator, dim3(grid1d(A_cols->n)), dim3(SX_WG), 0, ctx->stream, A_cols->n, A_cols->csc_ptr,
                       A_cols->csc_idx, A_cols->csc_val, xhat_loc, mask_loc, f_inv_all, ind);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_flow_indicator_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *X, const double *s,
                                    const double *d, double *ind) {
    SX_ENTER(ctx);
    SX_REQUIRE(S >= 0 && D >= 0, "negative size");
    if (S == 0 || D == 0) return SX_OK;
    SX_REQUIRE(X && s && d && ind, "NULL argument");
    hipLaunchKernelGGL(k_ot_indicator, dim3(grid1d(S * D)), dim3(SX_WG), 0, ctx->stream, S, D, X, s, d, ind);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_price_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *M, const double *y, double tol,
                           double *rc, sx_price_result *result_dev) {
    SX_ENTER(ctx);
    SX_REQUIRE(S > 0 && D > 0, "S and D must be positive");
    SX_REQUIRE(M && y && result_dev, "NULL argument");
    const unsigned nb = grid1d(S * D, 2048);
    SX_TRY(sx_reserve(ctx, sizeof(OtPartial) * nb));
    OtPartial *partial = static_cast<OtPartial *>(ctx->ws);
    hipLaunchKernelGGL(k_ot_price, dim3(nb), dim3(SX_WG), 0, ctx->stream, S, D, M, y, tol, rc, partial);
    hipLaunchKernelGGL(k_ot_price_final, dim3(1), dim3(SX_WG), 0, ctx->stream, partial, static_cast<int>(nb),
                       result_dev);
    SX_HIP(hipGetLastError());
    return SX_OK;
}
