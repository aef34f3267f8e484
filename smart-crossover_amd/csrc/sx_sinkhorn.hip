// Entropic OT warm start (Sinkhorn-Knopp): the step the reference's driver runs before the OT crossover
// (scripts/run_network_crossover.py:95-97, `sinkhorn(ot.s, ot.d, ot.M, reg=10, numItermax=1000)` from
// the third-party package POT).  SURVEY.md 8(f) rank 2.  POT is absent and unpinned: the algorithm
// below is its published `sinkhorn_knopp` (oracle/sinkhorn.py restates it), parity unpinned.
//
//   K = exp(M / (-reg));   u = 1/S, v = 1/D
//   repeat:  v = b ./ (K^T u);   u = 1 ./ (((1/a) .* K) v)          (breakdown -> previous pair, stop)
//            every 10th iteration (0, 10, ...):  err = || v .* (K^T u) - b ||_2 ;  stop when err < stopThr
//   plan = (u .* K) .* v
//
// Two dense matrix-vector products per iteration over a matrix that stays in L2 / Infinity Cache (784 x 784
// doubles = 4.9 MB at config 3): the loop is launch-bound, not bandwidth-bound, so it is plain FMA code
// (no MFMA: a single right-hand side is a GEMV) replayed as a hipGraph of 10 iterations.  K^T u uses
// lanes along the columns with the rows cut into slices (coalesced, no atomics, fixed summation order);
// K v uses one wave per row.
#include "sx_internal.h"
#include "sx_segwalk.h"

#include <cmath>

namespace {

struct SkState {
    long long iters;
    int done;      // 1 converged, 2 numerical breakdown
    int trouble;   // raised by the update kernels of the current iteration
    double err;
    double sumsq;
};

__global__ __launch_bounds__(SX_WG) void k_sk_kernel(int64_t n, const double *__restrict__ M, double reg,
                                                     double *__restrict__ K) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG)
        K[e] = exp(M[e] / (-reg));
}

__global__ __launch_bounds__(SX_WG) void k_sk_init(int64_t S, int64_t D, double *__restrict__ u, double *__restrict__ v) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (i < S) u[i] = 1.0 / static_cast<double>(S);
    if (i < D) v[i] = 1.0 / static_cast<double>(D);
}

__global__ __launch_bounds__(SX_WG) void k_sk_save(const SkState *st, int64_t S, int64_t D, const double *__restrict__ u,
                                                   const double *__restrict__ v, double *__restrict__ up,
                                                   double *__restrict__ vp) {
    if (st->done) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (i < S) up[i] = u[i];
    if (i < D) vp[i] = v[i];
}

// part[slice][j] = sum over the rows of the slice of K[i][j] * u[i]   (lane = column, ascending rows)
__global__ __launch_bounds__(SX_WG) void k_sk_cols(const SkState *st, int64_t S, int64_t D,
                                                   const double *__restrict__ K, const double *__restrict__ u,
                                                   double *__restrict__ part) {
    if (st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + lane;
    const int64_t slice = static_cast<int64_t>(blockIdx.y) * (SX_WG / 64) + wave;
    const int64_t nslices = static_cast<int64_t>(gridDim.y) * (SX_WG / 64);
    if (j >= D) return;
    const int64_t rows = (S + nslices - 1) / nslices;
    const int64_t i0 = slice * rows, i1 = (i0 + rows < S) ? i0 + rows : S;
    double acc = 0.0;
    for (int64_t i = i0; i < i1; ++i) acc += K[i * D + j] * u[i];
    part[slice * D + j] = acc;
}

// mode 0: v[j] = b[j] / KtU[j];  mode 1: partial sums of (v[j] * KtU[j] - b[j])^2 for the stopping test
__global__ __launch_bounds__(SX_WG) void k_sk_cols_finish(SkState *st, int64_t D, int64_t nslices,
                                                          const double *__restrict__ part,
                                                          const double *__restrict__ b, double *__restrict__ v,
                                                          int mode, double *__restrict__ err_part) {
    if (st->done) return;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    double sq = 0.0;
    if (j < D) {
        double ktu = 0.0;
        for (int64_t q = 0; q < nslices; ++q) ktu += part[q * D + j];
        if (mode == 0) {
            const double vj = b[j] / ktu;
            v[j] = vj;
            if (ktu == 0.0 || vj != vj || fabs(vj) == INFINITY) st->trouble = 1; // every writer stores the same value
        } else {
            const double d = v[j] * ktu - b[j];
            sq = d * d;
        }
    }
    if (mode == 1) {
        __shared__ double ws[SX_WG / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = ws[0];
            for (int w = 1; w < SX_WG / 64; ++w) t += ws[w];
            err_part[blockIdx.x] = t;
        }
    }
}

// u[i] = 1 / sum_j ((1/a[i]) * K[i][j]) * v[j]      (one wave per row)
__global__ __launch_bounds__(SX_WG) void k_sk_rows(SkState *st, int64_t S, int64_t D, const double *__restrict__ K,
                                                   const double *__restrict__ a, const double *__restrict__ v,
                                                   double *__restrict__ u) {
    if (st->done) return;
    const int lane = threadIdx.x & 63;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * (SX_WG / 64) + (threadIdx.x >> 6);
    if (i >= S) return;
    const double inv_a = 1.0 / a[i];
    double acc = 0.0;
    for (int64_t j = lane; j < D; j += 64) acc += (inv_a * K[i * D + j]) * v[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
        const double ui = 1.0 / acc;
        u[i] = ui;
        if (ui != ui || fabs(ui) == INFINITY) st->trouble = 1;
    }
}

// end of an iteration: on breakdown restore the previous pair and stop, else count the iteration
__global__ __launch_bounds__(SX_WG) void k_sk_guard(SkState *st, int64_t S, int64_t D, double *__restrict__ u,
                                                    double *__restrict__ v, const double *__restrict__ up,
                                                    const double *__restrict__ vp) {
    if (st->done) return;
    const int trouble = st->trouble;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (trouble) {
        if (i < S) u[i] = up[i];
        if (i < D) v[i] = vp[i];
    }
}

__global__ void k_sk_count(SkState *st) {
    if (st->done) return;
    if (st->trouble) st->done = 2;
    else st->iters += 1;
}

__global__ void k_sk_err(SkState *st, const double *__restrict__ err_part, int nparts, double stop_thr) {
    if (st->done) return;
    double t = 0.0;
    for (int k = 0; k < nparts; ++k) t += err_part[k];
    st->err = sqrt(t);
    if (st->err < stop_thr) st->done = 1;
}

__global__ __launch_bounds__(SX_WG) void k_sk_plan(int64_t S, int64_t D, const double *__restrict__ K,
                                                   const double *__restrict__ u, const double *__restrict__ v,
                                                   double *__restrict__ plan) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t i = e / D;
        plan[e] = (u[i] * K[e]) * v[e - i * D];
    }
}

inline unsigned grid1d(int64_t n, int64_t cap = 4096) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

} // namespace

SX_API int sx_sinkhorn_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *a, const double *b, const double *M,
                           double reg, int64_t max_iter, double stop_thr, double *plan, double *u_out, double *v_out,
                           sx_sinkhorn_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(S > 0 && D > 0, "S and D must be positive");
    SX_REQUIRE(a && b && M && result, "NULL argument");
    SX_REQUIRE(reg > 0 && max_iter >= 0, "reg must be positive and max_iter non-negative");
    memset(result, 0, sizeof(*result));
    hipStream_t s = ctx->stream;
    int64_t ysl = (S + 127) / 128;
    ysl = ysl < 1 ? 1 : (ysl > 16 ? 16 : ysl);
    const int64_t nslices = ysl * (SX_WG / 64);
    const int gD = static_cast<int>(grid1d(D));
    // workspace: state | K[S*D] | u up [S] | v vp [D] | part[nslices*D] | err_part[gD]
    const size_t uS = static_cast<size_t>(S), uD = static_cast<size_t>(D);
    const size_t bytes = 256 + sizeof(double) * (uS * uD + 2 * uS + 2 * uD + static_cast<size_t>(nslices) * uD + gD) + 64;
    SX_TRY(sx_reserve2(ctx, bytes));
    char *base = static_cast<char *>(ctx->ws2);
    SkState *st = reinterpret_cast<SkState *>(base);
    double *K = reinterpret_cast<double *>(base + 256);
    double *u = K + uS * uD, *up = u + uS, *v = up + uS, *vp = v + uD, *part = vp + uD;
    double *err_part = part + static_cast<size_t>(nslices) * uD;
    SX_HIP(hipMemsetAsync(st, 0, sizeof(SkState), s));
    hipLaunchKernelGGL(k_sk_kernel, dim3(grid1d(S * D)), dim3(SX_WG), 0, s, S * D, M, reg, K);
    const unsigned gSD = grid1d(S > D ? S : D);
    hipLaunchKernelGGL(k_sk_init, dim3(gSD), dim3(SX_WG), 0, s, S, D, u, v);
    const dim3 gc(static_cast<unsigned>((D + 63) / 64), static_cast<unsigned>(ysl));
    const unsigned gr = static_cast<unsigned>((S + SX_WG / 64 - 1) / (SX_WG / 64));

    auto iteration = [&](bool test) {
        hipLaunchKernelGGL(k_sk_save, dim3(gSD), dim3(SX_WG), 0, s, st, S, D, u, v, up, vp);
        hipLaunchKernelGGL(k_sk_cols, gc, dim3(SX_WG), 0, s, st, S, D, K, u, part);
        hipLaunchKernelGGL(k_sk_cols_finish, dim3(gD), dim3(SX_WG), 0, s, st, D, nslices, part, b, v, 0, err_part);
        hipLaunchKernelGGL(k_sk_rows, dim3(gr), dim3(SX_WG), 0, s, st, S, D, K, a, v, u);
        hipLaunchKernelGGL(k_sk_guard, dim3(gSD), dim3(SX_WG), 0, s, st, S, D, u, v, up, vp);
        hipLaunchKernelGGL(k_sk_count, dim3(1), dim3(1), 0, s, st);
        if (test) {
            hipLaunchKernelGGL(k_sk_cols, gc, dim3(SX_WG), 0, s, st, S, D, K, u, part);
            hipLaunchKernelGGL(k_sk_cols_finish, dim3(gD), dim3(SX_WG), 0, s, st, D, nslices, part, b, v, 1, err_part);
            hipLaunchKernelGGL(k_sk_err, dim3(1), dim3(1), 0, s, st, err_part, gD, stop_thr);
        }
    };
    // iterations come in groups of ten, the first of each group carrying the stopping test (ii % 10 == 0);
    // a group is captured once into a hipGraph and replayed, the host reads the state between groups
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (ctx->opt_graph && max_iter >= 20 && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int k = 0; k < 10; ++k) iteration(k == 0);
        if (hipStreamEndCapture(s, &graph) != hipSuccess || graph == nullptr ||
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
            exec = nullptr;
    }
    (void)hipGetLastError();
    SkState host;
    memset(&host, 0, sizeof(host));
    int rc = SX_OK;
    int64_t launched = 0;
    while (launched < max_iter && !host.done) {
        if (exec && launched + 10 <= max_iter) {
            if (hipGraphLaunch(exec, s) != hipSuccess) rc = SX_ERR_HIP;
            launched += 10;
        } else {
            const int64_t upto = (launched + 10 < max_iter) ? launched + 10 : max_iter;
            for (; launched < upto; ++launched) iteration(launched % 10 == 0);
        }
        if (rc != SX_OK || hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            sx_set_error("Sinkhorn iteration batch failed");
            rc = SX_ERR_HIP;
            break;
        }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != SX_OK) return rc;
    if (plan) hipLaunchKernelGGL(k_sk_plan, dim3(grid1d(S * D)), dim3(SX_WG), 0, s, S, D, K, u, v, plan);
    if (u_out) SX_HIP(hipMemcpyAsync(u_out, u, sizeof(double) * uS, hipMemcpyDeviceToDevice, s));
    if (v_out) SX_HIP(hipMemcpyAsync(v_out, v, sizeof(double) * uD, hipMemcpyDeviceToDevice, s));
    SX_HIP(hipGetLastError());
    SX_HIP(hipStreamSynchronize(s));
    result->iters = host.iters;
    result->err = host.iters > 0 || host.done ? host.err : 1.0;
    result->status = host.done; // 0 iteration limit, 1 converged, 2 numerical breakdown
    return SX_OK;
}
