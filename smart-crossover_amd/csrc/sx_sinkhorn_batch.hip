// Batched entropic OT warm start: B instance pairs that share one cost matrix -- the reference's driver runs
// POT's sinkhorn on ten MNIST image pairs over the same 28 x 28 grid, one after the other
// (scripts/run_network_crossover.py:95-101) -- solved together.  SURVEY.md 8(f) rank 2: with the scaling
// vectors of the B instances side by side, K^T U and K V are (D x S)(S x B) and (S x D)(D x B) products, the
// one genuinely dense contraction near the path, and they run on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64).  An instance's marginals live on the shared grid; a pixel it does not use has
// mass 0 and IEEE arithmetic takes it out by itself (u_i = 1 / ((1/0) * (K v)_i) = 0, v_j = 0 / (K^T u)_j = 0);
// with the scaling vectors started on the support too, every instance walks through exactly the iterates POT
// computes on its own support.
//
// Algorithm per instance: POT's published sinkhorn_knopp, as sx_sinkhorn.hip / oracle/sinkhorn.py state it;
// the instances stop independently (per-instance done flags mask their columns).
//
// Layout: U[S][16], V[D][16] -- instance-minor, 16 = one MFMA tile of instances (B <= 16) -- so that the B
// operand of a k-step (4 rows x 16 instances) is 512 contiguous bytes; K and K^T both resident (4.9 MB each
// at 784 x 784) so that the A operand of either product is read along rows.  One workgroup per 16 output
// rows; its four waves split the contraction and their tiles are added in wave order (fixed order).
#include "sx_internal.h"
#include "sx_segwalk.h"

#include <cmath>

namespace {

constexpr int SKB = 16; // instances per batch = columns of one MFMA tile

typedef double skb_v4d __attribute__((ext_vector_type(4)));

struct SkbState {
    long long iters[SKB];
    int done[SKB];    // 1 converged, 2 numerical breakdown, 3 padding column (never iterates)
    int trouble[SKB]; // raised by the update kernels of the current iteration
    double err[SKB];
    int all_done;
};

__global__ __launch_bounds__(SX_WG) void k_skb_kernel(int64_t S, int64_t D, const double *__restrict__ M, double reg,
                                                      double *__restrict__ K, double *__restrict__ KT) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t i = e / D, j = e - i * D;
        const double k = exp(M[e] / (-reg));
        K[e] = k;
        KT[j * S + i] = k;
    }
}

// support sizes: cnt[c] = #{i : a[c][i] != 0}, cnt[16 + c] = #{j : b[c][j] != 0}   (one workgroup per instance)
__global__ __launch_bounds__(SX_WG) void k_skb_support(int64_t S, int64_t D, const double *__restrict__ a,
                                                       const double *__restrict__ b, int *__restrict__ cnt) {
    __shared__ int ws[2][SX_WG / 64];
    const int c = blockIdx.x;
    int na = 0, nb = 0;
    for (int64_t i = threadIdx.x; i < S; i += SX_WG) na += a[static_cast<int64_t>(c) * S + i] != 0.0;
    for (int64_t j = threadIdx.x; j < D; j += SX_WG) nb += b[static_cast<int64_t>(c) * D + j] != 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        na += __shfl_down(na, o, 64);
        nb += __shfl_down(nb, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        ws[0][threadIdx.x >> 6] = na;
        ws[1][threadIdx.x >> 6] = nb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[c] = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
        cnt[SKB + c] = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
    }
}

// a[B][S] (instance-major, as the caller has them) -> inv_a[S][16] = 1 / a; likewise bb[D][16] = b.  The scaling
// vectors start at 1 / (size of the instance's support) on its support and 0 off it -- POT's start on the
// problem restricted to the support -- so that the iterates ARE those of the restricted problem
__global__ __launch_bounds__(SX_WG) void k_skb_init(SkbState *st, int64_t S, int64_t D, int B,
                                                    const double *__restrict__ a, const double *__restrict__ b,
                                                    const int *__restrict__ cnt,
                                                    double *__restrict__ inv_a, double *__restrict__ bb,
                                                    double *__restrict__ U, double *__restrict__ V) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    const int64_t r = e / SKB;
    const int c = static_cast<int>(e - r * SKB);
    if (r < S) {
        const double ai = (c < B) ? a[static_cast<int64_t>(c) * S + r] : 0.0;
        inv_a[e] = (c < B) ? 1.0 / ai : 0.0;
        U[e] = (ai != 0.0) ? 1.0 / static_cast<double>(cnt[c]) : 0.0;
    }
    if (r < D) {
        const double bj = (c < B) ? b[static_cast<int64_t>(c) * D + r] : 0.0;
        bb[e] = bj;
        V[e] = (bj != 0.0) ? 1.0 / static_cast<double>(cnt[SKB + c]) : 0.0;
    }
    if (e < SKB) {
        st->iters[e] = 0;
        st->done[e] = (e < B) ? 0 : 3;
        st->trouble[e] = 0;
        st->err[e] = 1.0;
        if (e == 0) st->all_done = 0;
    }
}

__global__ __launch_bounds__(SX_WG) void k_skb_save(const SkbState *st, int64_t nS, int64_t nD,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    double *__restrict__ Up, double *__restrict__ Vp) {
    if (st->all_done) return;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (e < nS) Up[e] = U[e];
    if (e < nD) Vp[e] = V[e];
}

// Y[r0 .. r0+16)[0..16) = A[rows r0..][:] * X[:][0..16) with A = `mat` (R x C, row-major) TRANSPOSED, i.e.
// Y[r][b] = sum_k mat[k][r] * X[k][b]: the caller passes K for K^T U and K^T for K V, so that both products read
// the matrix along its rows (lane = output row, 16 contiguous doubles per k).  One workgroup per 16 output rows,
// wave w sums k = w, w + 4, ... in steps of 4 consecutive k (one MFMA), the four tiles are added in wave order.
//   mode 0 (columns):  V[r][b] = bb[r][b] / Y          trouble when Y == 0 or V is not finite
//   mode 1 (rows):     U[r][b] = 1 / (inv_a[r][b] * Y) trouble when U is not finite
//   mode 2 (test):     err_part[tile][b] = sum over the tile's rows of (V[r][b] * Y - bb[r][b])^2
// Columns of finished instances are left untouched.
__global__ __launch_bounds__(SX_WG) void k_skb_product(SkbState *st, int mode, int64_t R, int64_t C,
                                                       const double *__restrict__ mat, const double *__restrict__ X,
                                                       const double *__restrict__ scale, double *__restrict__ out,
                                                       double *__restrict__ err_part) {
    if (st->all_done) return;
    __shared__ double tile[SX_WG / 64][16][SKB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 16;
    const int row = lane & 15, kq = lane >> 4; // A: row = lane & 15, k = lane >> 4;  B: col = lane & 15, k = lane >> 4
    const int64_t rr = (r0 + row < R) ? r0 + row : R - 1;
    skb_v4d acc = {0.0, 0.0, 0.0, 0.0};
    const int64_t ksteps = (C + 3) / 4;
    for (int64_t ks = wave; ks < ksteps; ks += SX_WG / 64) {
        const int64_t k = ks * 4 + kq;
        const bool live = k < C;
        const double a = live ? mat[k * R + rr] : 0.0;
        const double b = live ? X[k * SKB + row] : 0.0; // `row` is the instance here (B operand: col = lane & 15)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // C/D layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[wave][(lane >> 4) + 4 * q][lane & 15] = acc[q];
    __syncthreads();
    const int t = threadIdx.x; // 256 lanes = 16 rows x 16 instances
    const int lr = t >> 4, inst = t & 15;
    double y = tile[0][lr][inst];
#pragma unroll
    for (int w = 1; w < SX_WG / 64; ++w) y = y + tile[w][lr][inst];
    const int64_t r = r0 + lr;
    const bool active = r < R && st->done[inst] == 0;
    double sq = 0.0;
    if (active) {
        const int64_t e = r * SKB + inst;
        if (mode == 0) {
            const double v = scale[e] / y;
            out[e] = v;
            if (y == 0.0 || v != v || fabs(v) == INFINITY) st->trouble[inst] = 1; // every writer stores the same value
        } else if (mode == 1) {
            const double u = 1.0 / (scale[e] * y);
            out[e] = u;
            if (u != u || fabs(u) == INFINITY) st->trouble[inst] = 1;
        } else {
            const double d = out[e] * y - scale[e];
            sq = d * d;
        }
    }
    if (mode == 2) { // per instance: sum over the 16 rows of the tile, rows in ascending order
        __syncthreads();
        tile[0][lr][inst] = sq;
        __syncthreads();
        if (t < SKB) {
            double s = 0.0;
            for (int q = 0; q < 16; ++q) s += tile[0][q][t];
            err_part[static_cast<int64_t>(blockIdx.x) * SKB + t] = s;
        }
    }
}

// end of an iteration, per instance: breakdown -> previous pair back and stop; else count the iteration
__global__ __launch_bounds__(SX_WG) void k_skb_guard(SkbState *st, int64_t nS, int64_t nD, double *__restrict__ U,
                                                     double *__restrict__ V, const double *__restrict__ Up,
                                                     const double *__restrict__ Vp) {
    if (st->all_done) return;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    const int inst = static_cast<int>(e & (SKB - 1));
    if (st->done[inst] == 0 && st->trouble[inst]) {
        if (e < nS) U[e] = Up[e];
        if (e < nD) V[e] = Vp[e];
    }
}

// one lane per instance: close the iteration (and the stopping test when err_part != nullptr)
__global__ void k_skb_count(SkbState *st, const double *__restrict__ err_part, int ntiles, double stop_thr) {
    if (st->all_done) return;
    const int b = threadIdx.x;
    if (b < SKB && st->done[b] == 0) {
        if (!err_part) {
            if (st->trouble[b]) st->done[b] = 2;
            else st->iters[b] += 1;
        } else {
            double t = 0.0;
            for (int k = 0; k < ntiles; ++k) t += err_part[static_cast<int64_t>(k) * SKB + b];
            st->err[b] = sqrt(t);
            if (st->err[b] < stop_thr) st->done[b] = 1;
        }
    }
    __syncthreads();
    if (b == 0) {
        int all = 1;
        for (int k = 0; k < SKB; ++k) all &= (st->done[k] != 0);
        st->all_done = all;
    }
}

// plans[b][i][j] = (u[i][b] * K[i][j]) * v[j][b]
__global__ __launch_bounds__(SX_WG) void k_skb_plan(int64_t S, int64_t D, int B, const double *__restrict__ K,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    double *__restrict__ plans) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t i = e / D, j = e - i * D;
        const double k = K[e];
        for (int b = 0; b < B; ++b) plans[static_cast<int64_t>(b) * n + e] = (U[i * SKB + b] * k) * V[j * SKB + b];
    }
}

// [S][16] -> out[B][S]
__global__ __launch_bounds__(SX_WG) void k_skb_unpack(int64_t S, int B, const double *__restrict__ X,
                                                      double *__restrict__ out) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (e >= S * B) return;
    const int64_t b = e / S, i = e - b * S;
    out[e] = X[i * SKB + b];
}

inline unsigned grid1d(int64_t n, int64_t cap = 4096) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

} // namespace

SX_API int sx_sinkhorn_batch_dev(sx_ctx *ctx, int64_t S, int64_t D, int64_t B, const double *a, const double *b,
                                 const double *M, double reg, int64_t max_iter, double stop_thr, double *plans,
                                 double *u_out, double *v_out, sx_sinkhorn_result *results) {
    SX_ENTER(ctx);
    SX_REQUIRE(S > 0 && D > 0, "S and D must be positive");
    SX_REQUIRE(B >= 1 && B <= SKB, "1 <= B <= 16 instances per batch");
    SX_REQUIRE(a && b && M && results, "NULL argument");
    SX_REQUIRE(reg > 0 && max_iter >= 0, "reg must be positive and max_iter non-negative");
    hipStream_t s = ctx->stream;
    const size_t uS = static_cast<size_t>(S), uD = static_cast<size_t>(D);
    const int64_t tS = (S + 15) / 16, tD = (D + 15) / 16;
    const int64_t nS = S * SKB, nD = D * SKB;
    // workspace: state, support counts | K, KT [S*D] | inv_a U Up [S*16] | bb V Vp [D*16] | err_part [tD*16]
    const size_t bytes = 1024 + sizeof(double) * (2 * uS * uD + 3 * uS * SKB + 3 * uD * SKB + static_cast<size_t>(tD) * SKB) + 64;
    SX_TRY(sx_reserve2(ctx, bytes));
    char *base = static_cast<char *>(ctx->ws2);
    SkbState *st = reinterpret_cast<SkbState *>(base);
    double *K = reinterpret_cast<double *>(base + 1024);
    double *KT = K + uS * uD;
    double *inv_a = KT + uS * uD, *U = inv_a + uS * SKB, *Up = U + uS * SKB;
    double *bb = Up + uS * SKB, *V = bb + uD * SKB, *Vp = V + uD * SKB;
    double *err_part = Vp + uD * SKB;
    hipLaunchKernelGGL(k_skb_kernel, dim3(grid1d(S * D)), dim3(SX_WG), 0, s, S, D, M, reg, K, KT);
    const unsigned gE = grid1d(nS > nD ? nS : nD, 1 << 20);
    int *cnt = reinterpret_cast<int *>(base + 768); // 2 x 16 ints behind the state (sizeof(SkbState) < 512)
    hipLaunchKernelGGL(k_skb_support, dim3(static_cast<unsigned>(B)), dim3(SX_WG), 0, s, S, D, a, b, cnt);
    hipLaunchKernelGGL(k_skb_init, dim3(gE), dim3(SX_WG), 0, s, st, S, D, static_cast<int>(B), a, b, cnt, inv_a, bb, U, V);

    auto iteration = [&](bool test) {
        hipLaunchKernelGGL(k_skb_save, dim3(gE), dim3(SX_WG), 0, s, st, nS, nD, U, V, Up, Vp);
        // V = b ./ (K^T U): output rows = columns of K
        hipLaunchKernelGGL(k_skb_product, dim3(static_cast<unsigned>(tD)), dim3(SX_WG), 0, s, st, 0, D, S, K, U, bb, V,
                           static_cast<double *>(nullptr));
        // U = 1 ./ ((1/a) .* (K V)): output rows = rows of K = columns of K^T
        hipLaunchKernelGGL(k_skb_product, dim3(static_cast<unsigned>(tS)), dim3(SX_WG), 0, s, st, 1, S, D, KT, V, inv_a, U,
                           static_cast<double *>(nullptr));
        hipLaunchKernelGGL(k_skb_guard, dim3(gE), dim3(SX_WG), 0, s, st, nS, nD, U, V, Up, Vp);
        hipLaunchKernelGGL(k_skb_count, dim3(1), dim3(64), 0, s, st, static_cast<const double *>(nullptr), 0, stop_thr);
        if (test) {
            hipLaunchKernelGGL(k_skb_product, dim3(static_cast<unsigned>(tD)), dim3(SX_WG), 0, s, st, 2, D, S, K, U, bb, V,
                               err_part);
            hipLaunchKernelGGL(k_skb_count, dim3(1), dim3(64), 0, s, st, err_part, static_cast<int>(tD), stop_thr);
        }
    };
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (ctx->opt_graph && max_iter >= 20 && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int k = 0; k < 10; ++k) iteration(k == 0);
        if (hipStreamEndCapture(s, &graph) != hipSuccess || graph == nullptr ||
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
            exec = nullptr;
    }
    (void)hipGetLastError();
    SkbState host;
    memset(&host, 0, sizeof(host));
    int rc = SX_OK;
    int64_t launched = 0;
    while (launched < max_iter && !host.all_done) {
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
            sx_set_error("batched Sinkhorn iteration group failed");
            rc = SX_ERR_HIP;
            break;
        }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != SX_OK) return rc;
    if (plans) hipLaunchKernelGGL(k_skb_plan, dim3(grid1d(S * D)), dim3(SX_WG), 0, s, S, D, static_cast<int>(B), K, U, V, plans);
    if (u_out) hipLaunchKernelGGL(k_skb_unpack, dim3(grid1d(S * B, 1 << 20)), dim3(SX_WG), 0, s, S, static_cast<int>(B), U, u_out);
    if (v_out) hipLaunchKernelGGL(k_skb_unpack, dim3(grid1d(D * B, 1 << 20)), dim3(SX_WG), 0, s, D, static_cast<int>(B), V, v_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipStreamSynchronize(s));
    for (int64_t k = 0; k < B; ++k) {
        results[k].iters = host.iters[k];
        results[k].status = host.done[k]; // 0 iteration limit, 1 converged, 2 numerical breakdown
        results[k].err = (host.iters[k] > 0 || host.done[k]) ? host.err[k] : 1.0;
    }
    return SX_OK;
}
