// K4: norm of the projection (I - Y^T (Y Y^T)^+ Y) v with Y = [A, I_<] diag(xx), v = diag(xx) c_std,
// by matrix-free conjugate gradients on the device (reference: get_projector_Xc / apply_projector /
// get_scale_factor, lp_methods/algorithms.py:162-193, which forms Y Y^T explicitly and calls
// scipy's cg with tol=1e-8, maxiter=1000).
//
// With xa = xx[:n] (structural columns) and xs = xx[n:] scattered to the '<' rows (0 elsewhere):
//     (Y Y^T) p = A (xa^2 .* (A^T p)) + xs^2 .* p          b = Y v = A (xa^2 .* c)
//     proj      = [ xa .* (c - A^T z) ;  -xs .* z ]         (slack costs are zero)
// One CG iteration is four launches: CSC pass (w), CSR pass (q, partial p.q), update of z and r
// (partial r.r), update of p.  Scalars live in device memory; every kernel returns at once when the
// `done` flag is set, so the host enqueues iterations in batches and polls the flag between batches.
// Reductions are two-stage with a fixed order: results are reproducible run to run.  The recurrence
// and the stopping rule (||r|| < tol*||b||, tested before each iteration; legacy immediate exit when
// ||b|| <= tol) are those of the reference's pinned scipy.
#include "sx_internal.h"
#include "sx_rowblock.h"
#include "sx_segwalk.h"
#include "sx_window.h"

#include <cmath>

namespace {

constexpr int CG_CHUNK = 4096;
constexpr int CG_GRID = 2048; // workgroups of the grid-stride SpMV passes = number of partials

struct CgState {
    double rho[2];     // r.r of the current / next iteration (indexed by iteration parity)
    double atol;       // tol * ||b||
    double sumsq;      // final ||proj||^2
    long long iters;   // completed iterations
    int done;          // 1: converged (or finished), kernels become no-ops
    int converged;
};

struct StageDot {
    const double *__restrict__ vec;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const {
        o[0] = v * vec[i];
    }
};

__device__ __forceinline__ void walk_range(int64_t ntiles, int swizzle, int64_t &t, int64_t &t_end,
                                           int64_t &t_step) {
    t = blockIdx.x;
    t_end = ntiles;
    t_step = gridDim.x;
    if (swizzle) {
        const int64_t per = (ntiles + 7) >> 3;
        t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        t_end = ((blockIdx.x & 7) + 1) * per;
        if (t_end > ntiles) t_end = ntiles;
        t_step = gridDim.x >> 3;
    }
}

// fixed-order block sum; result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double s[SX_WG / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads(); // protect s[] against the previous use
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    return s[0] + s[1] + s[2] + s[3];
}

// every workgroup re-reduces the partial array in the same order -> identical value everywhere
__device__ __forceinline__ double reduce_partials(const double *__restrict__ partial, int np) {
    double v = 0.0;
    for (int k = threadIdx.x; k < np; k += SX_WG) v += partial[k];
    double tot = block_sum(v);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = tot;
    __syncthreads();
    return bc;
}

// out[j] = scale[j]^2 * (A^T in)[j]                                     (CSC pass)
__global__ __launch_bounds__(SX_WG) void k_cg_at(const CgState *st, const int64_t *__restrict__ tiles,
                                                 int64_t ntiles, int swizzle,
                                                 const int64_t *__restrict__ colptr,
                                                 const int32_t *__restrict__ rowidx,
                                                 const double *__restrict__ val,
                                                 const double *__restrict__ in,
                                                 const double *__restrict__ scale,
                                                 double *__restrict__ out) {
    if (st->done) return;
    __shared__ sx_walk_lds<1, CG_CHUNK> lds;
    int64_t t, t_end, t_step;
    walk_range(ntiles, swizzle, t, t_end, t_step);
    for (; t < t_end; t += t_step) {
        double acc[1];
        int64_t j;
        bool valid;
        double s = 0.0;
        auto pre = [&](int64_t seg, bool ok) {
            if (ok) s = scale[seg];
        };
        sx_segwalk<1, CG_CHUNK>(tiles, t, colptr, rowidx, val, StageDot{in}, lds, j, valid, acc, pre);
        if (valid) out[j] = (s * s) * acc[0];
    }
}

// the same pass behind the LDS operand window (sx_window.h): grid-stride over runs of RUN tiles
template <int RUN>
__global__ __launch_bounds__(SX_WG) void k_cg_at_lw(const CgState *st, const int64_t *__restrict__ tiles,
                                                    int64_t ntiles, int swizzle, const int32_t *__restrict__ win_lo,
                                                    const int64_t *__restrict__ colptr,
                                                    const int32_t *__restrict__ rowidx,
                                                    const double *__restrict__ val, int64_t m,
                                                    const double *__restrict__ in,
                                                    const double *__restrict__ scale, double *__restrict__ out) {
    if (st->done) return;
    __shared__ sx_walk_lds<1, SXL_CHUNK> lds;
    __shared__ double win[SXL_CAP];
    const int64_t nruns = (ntiles + RUN - 1) / RUN;
    int64_t r, r_end, r_step;
    walk_range(nruns, swizzle, r, r_end, r_step);
    for (; r < r_end; r += r_step) {
        const int64_t t0 = r * RUN;
        const int64_t t1 = (t0 + RUN < ntiles) ? t0 + RUN : ntiles;
        const int64_t wlo = win_lo[(t0 + t1 - 1) >> 1];
        __syncthreads(); // the previous run's gathers are done with the window
        sx_window_fill(win, in, wlo, m);
        __syncthreads();
        for (int64_t t = t0; t < t1; ++t) {
            double acc[1];
            int64_t j;
            bool valid;
            double s = 0.0;
            auto pre = [&](int64_t seg, bool ok) {
                if (ok) s = scale[seg];
            };
            sx_segwalk<1, SXL_CHUNK>(tiles, t, colptr, rowidx, val, sx_stage_win{in, win, wlo}, lds, j, valid, acc, pre);
            if (valid) out[j] = (s * s) * acc[0];
        }
    }
}

// q[i] = (A w)[i] + xs[i]^2 * p[i];  partial[block] = sum p[i]*q[i]       (CSR pass)
// with p == nullptr: q = A w only, partial = sum q[i]^2  (used for b = Y v and rho0 = b.b)
__global__ __launch_bounds__(SX_WG) void k_cg_a(const CgState *st, const int64_t *__restrict__ tiles,
                                                int64_t ntiles, int swizzle,
                                                const int64_t *__restrict__ rowptr,
                                                const int32_t *__restrict__ colidx,
                                                const double *__restrict__ val,
                                                const double *__restrict__ w,
                                                const double *__restrict__ xs,
                                                const double *__restrict__ p, double *__restrict__ q,
                                                double *__restrict__ partial) {
    if (st->done) return;
    __shared__ sx_walk_lds<1, CG_CHUNK> lds;
    int64_t t, t_end, t_step;
    walk_range(ntiles, swizzle, t, t_end, t_step);
    double dot = 0.0;
    for (; t < t_end; t += t_step) {
        double acc[1];
        int64_t i;
        bool valid;
        double s = 0.0, pi = 0.0; // epilogue operands requested before the walk
        auto pre = [&](int64_t seg, bool ok) {
            if (ok && p) {
                s = xs[seg];
                pi = p[seg];
            }
        };
        sx_segwalk<1, CG_CHUNK>(tiles, t, rowptr, colidx, val, StageDot{w}, lds, i, valid, acc, pre);
        if (valid) {
            double qi = acc[0];
            if (p) {
                qi = qi + (s * s) * pi;
                dot += pi * qi;
            } else {
                dot += qi * qi;
            }
            q[i] = qi;
        }
    }
    const double tot = block_sum(dot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// t[j] = xa[j]^2 * c[j]
__global__ __launch_bounds__(SX_WG) void k_cg_scale_c(int64_t n, const double *__restrict__ xa,
                                                      const double *__restrict__ c,
                                                      double *__restrict__ t) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG)
        t[j] = (xa[j] * xa[j]) * c[j];
}

// start: r = p = b (already in r), z = 0, rho[0] = sum(partial)
__global__ __launch_bounds__(SX_WG) void k_cg_init(CgState *st, int64_t m, const double *__restrict__ partial,
                                                   int np, const double *__restrict__ r,
                                                   double *__restrict__ p, double *__restrict__ z) {
    const double rho0 = reduce_partials(partial, np);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        p[i] = r[i];
        z[i] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rho[0] = rho0;
        st->rho[1] = 0.0;
        st->iters = 0;
        st->done = 0;
        st->converged = 0;
        st->sumsq = 0.0;
        st->atol = 0.0; // set by the host once it knows ||b||
    }
}

// alpha = rho / (p.q);  z += alpha p;  r -= alpha q;  partial_rr[block] = sum r^2
__global__ __launch_bounds__(SX_WG) void k_cg_update_zr(const CgState *st, int parity, int64_t m,
                                                        const double *__restrict__ partial_pq, int np,
                                                        const double *__restrict__ p,
                                                        const double *__restrict__ q,
                                                        double *__restrict__ z, double *__restrict__ r,
                                                        double *__restrict__ partial_rr) {
    if (st->done) return;
    const double pq = reduce_partials(partial_pq, np);
    const double alpha = st->rho[parity] / pq;
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        z[i] = z[i] + alpha * p[i];
        const double ri = r[i] - alpha * q[i];
        r[i] = ri;
        acc += ri * ri;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial_rr[blockIdx.x] = tot;
}

// rho' = sum(partial_rr); stop test; beta = rho'/rho; p = r + beta p
__global__ __launch_bounds__(SX_WG) void k_cg_update_p(CgState *st, int parity, int64_t m,
                                                       const double *__restrict__ partial_rr, int np,
                                                       const double *__restrict__ r,
                                                       double *__restrict__ p) {
    if (st->done) return;
    const double rho_new = reduce_partials(partial_rr, np);
    const double rho_old = st->rho[parity];
    const bool stop = sqrt(rho_new) < st->atol;
    if (!stop) {
        const double beta = rho_new / rho_old;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
             i += static_cast<int64_t>(gridDim.x) * SX_WG)
            p[i] = r[i] + beta * p[i]; // scipy: p *= beta; p += r  (same two roundings)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rho[parity ^ 1] = rho_new;
        st->iters = st->iters + 1;
        if (stop) {
            st->converged = 1;
            __threadfence();
            st->done = 1;
        }
    }
}

// partial[block] = sum_j (xa[j] * (c[j] - (A^T z)[j]))^2
__global__ __launch_bounds__(SX_WG) void k_cg_proj_cols(int64_t n, const double *__restrict__ xa,
                                                        const double *__restrict__ c,
                                                        const double *__restrict__ atz,
                                                        double *__restrict__ partial,
                                                        double *__restrict__ proj_out) {
    double acc = 0.0;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double pj = xa[j] * (c[j] - atz[j]);
        if (proj_out) proj_out[j] = pj;
        acc += pj * pj;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// r[i] += xs[i]^2 * cs[i];  partial = sum r[i]^2 (only when the slack columns carry a cost)
__global__ __launch_bounds__(SX_WG) void k_cg_slack_cost(int64_t m, const double *__restrict__ xs,
                                                         const double *__restrict__ cs, double *__restrict__ r,
                                                         double *__restrict__ partial) {
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double s = xs[i];
        const double ri = r[i] + (s * s) * cs[i];
        r[i] = ri;
        acc += ri * ri;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SX_WG) void k_cg_proj_rows(int64_t m, const double *__restrict__ xs,
                                                        const double *__restrict__ cs,
                                                        const double *__restrict__ z,
                                                        double *__restrict__ partial,
                                                        double *__restrict__ proj_out) {
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        // slack block of proj: xs .* (cs - z); without slack costs it is -xs .* z
        const double pi = cs ? xs[i] * (cs[i] - z[i]) : -(xs[i] * z[i]);
        if (proj_out) proj_out[i] = pi;
        acc += pi * pi;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SX_WG) void k_cg_finish(CgState *st, const double *__restrict__ pa, int na,
                                                     const double *__restrict__ pb, int nb) {
    const double a = reduce_partials(pa, na);
    const double b = reduce_partials(pb, nb);
    if (threadIdx.x == 0) st->sumsq = a + b;
}

__global__ void k_cg_set_atol(CgState *st, double atol, int done) {
    st->atol = atol;
    st->done = done;
}

__global__ void k_cg_ones(int64_t n, double *v) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG)
        v[j] = 1.0;
}

inline int grid_for(const sx_ctx *ctx, int64_t ntiles) {
    int64_t g = ntiles < CG_GRID ? ntiles : CG_GRID;
    if (ctx->opt_xcd_swizzle && ntiles >= 64) g &= ~static_cast<int64_t>(7);
    return static_cast<int>(g < 1 ? 1 : g);
}

// degenerate shapes: with no rows Y is empty and the projector is the identity, proj = xa .* c; with no
// columns the only directions are slack columns, which Y = diag(xs) pins to zero wherever xs != 0
__global__ __launch_bounds__(1024) void k_cg_no_rows(int64_t n, const double *__restrict__ xa,
                                                     const double *__restrict__ c, double *__restrict__ proj_cols,
                                                     double *__restrict__ sumsq_out) {
    __shared__ double part[16];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 1024) {
        const double v = xa[j] * c[j];
        if (proj_cols) proj_cols[j] = v;
        acc = acc + v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = part[0];
        for (int w = 1; w < 16; ++w) t += part[w];
        *sumsq_out = t;
    }
}

} // namespace

SX_API int sx_projector_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                            const double *c, double tol, int maxiter, double *proj_cols,
                            double *proj_rows, sx_cg_result *result) {
    return sx_projector_std_dev(ctx, A, xa, xs, c, nullptr, tol, maxiter, proj_cols, proj_rows, result);
}

SX_API int sx_projector_std_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                                const double *c, const double *cs, double tol, int maxiter,
                                double *proj_cols, double *proj_rows, sx_cg_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr && result != nullptr, "matrix or result is NULL");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the projector needs both layouts of A");
    SX_REQUIRE(xa && xs && c, "xa, xs or c is NULL");
    SX_REQUIRE(maxiter >= 0 && tol >= 0, "bad tol/maxiter");
    const int64_t m = A->m, n = A->n;
    memset(result, 0, sizeof(*result));
    if (m == 0 || n == 0) {
        // the reference's apply_projector on an empty Y returns v unchanged (algorithms.py:183-187)
        result->converged = 1;
        if (n == 0) {
            if (proj_rows && m > 0) SX_HIP(hipMemsetAsync(proj_rows, 0, sizeof(double) * static_cast<size_t>(m), ctx->stream));
            return SX_OK;
        }
        SX_TRY(sx_reserve(ctx, 256));
        double *sumsq = static_cast<double *>(ctx->ws);
        hipLaunchKernelGGL(k_cg_no_rows, dim3(1), dim3(1024), 0, ctx->stream, n, xa, c, proj_cols, sumsq);
        double host_sumsq = 0.0;
        SX_HIP(hipMemcpyAsync(&host_sumsq, sumsq, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SX_HIP(hipStreamSynchronize(ctx->stream));
        result->proj_norm = sqrt(host_sumsq);
        return SX_OK;
    }

    const int swzT = sx_csc_swizzle(ctx, A);
    const int swzA = (ctx->opt_xcd_swizzle && A->n_csr_tiles >= 64 && A->csr_imbalance <= SX_SWIZZLE_MAX_IMBALANCE) ? 1 : 0;
    const int gT = grid_for(ctx, A->n_csc_tiles), gA = grid_for(ctx, A->n_csr_tiles);
    // vector kernels: grid-stride with ~4 elements per lane, at most 1024 workgroups (= partials); a
    // grid sized for the larger of m and n keeps small problems from launching mostly idle blocks
    int64_t gv64 = ((m > n ? m : n) + 4 * SX_WG - 1) / (4 * SX_WG);
    const int gv = static_cast<int>(gv64 < 1 ? 1 : (gv64 > 1024 ? 1024 : gv64));

    // column pass behind the LDS operand window when the matrix's auto rule (or the option) says so; the
    // table is built here, outside any graph capture
    int win_run = 0;
    SX_TRY(sx_window_run_csc(ctx, A, &win_run));
    // workspace: state | partial_pq[CG_GRID] | partial_rr[CG_GRID] | z r p q [m] | w atz [n]
    const size_t off_state = 0;
    const size_t off_ppq = 256;
    const size_t off_prr = off_ppq + sizeof(double) * CG_GRID;
    const size_t off_vec = off_prr + sizeof(double) * CG_GRID;
    const size_t bytes =
        off_vec + sizeof(double) * (4 * static_cast<size_t>(m) + 2 * static_cast<size_t>(n)) + 256;
    SX_TRY(sx_reserve(ctx, bytes));
    char *base = static_cast<char *>(ctx->ws);
    CgState *st = reinterpret_cast<CgState *>(base + off_state);
    double *ppq = reinterpret_cast<double *>(base + off_ppq);
    double *prr = reinterpret_cast<double *>(base + off_prr);
    double *z = reinterpret_cast<double *>(base + off_vec);
    double *r = z + m, *p = r + m, *q = p + m, *w = q + m, *atz = w + n;
    hipStream_t s = ctx->stream;

    auto launch_at = [&](const double *in, const double *scale, double *out) { // out = scale^2 .* (A^T in)
        if (win_run >= 4)
            hipLaunchKernelGGL((k_cg_at_lw<4>), dim3(gT), dim3(SX_WG), 0, s, st, A->csc_tiles, A->n_csc_tiles, swzT,
                               A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, m, in, scale, out);
        else if (win_run >= 1)
            hipLaunchKernelGGL((k_cg_at_lw<1>), dim3(gT), dim3(SX_WG), 0, s, st, A->csc_tiles, A->n_csc_tiles, swzT,
                               A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, m, in, scale, out);
        else
            hipLaunchKernelGGL(k_cg_at, dim3(gT), dim3(SX_WG), 0, s, st, A->csc_tiles, A->n_csc_tiles, swzT, A->csc_ptr,
                               A->csc_idx, A->csc_val, in, scale, out);
    };
    // row pass q = A w (+ xs^2 .* p), partials of p.q (or q.q) in ppq: over the column-blocked copy of the rows
    // when the matrix has one (sx_rowblock.h), else the plain walk; nA = number of partials either leaves
    const sx_rowblock *rb = nullptr;
    SX_TRY(sx_rowblock_get(ctx, A, &rb));
    int nA = gA;
    auto launch_a = [&](const double *vec, const double *pvec, double *out) -> int {
        if (rb) return sx_rb_cg_a(ctx, rb, n, st, vec, xs, pvec, out, ppq, CG_GRID, &nA);
        hipLaunchKernelGGL(k_cg_a, dim3(gA), dim3(SX_WG), 0, s, st, A->csr_tiles, A->n_csr_tiles, swzA, A->csr_ptr,
                           A->csr_idx, A->csr_val, vec, xs, pvec, out, ppq);
        return SX_OK;
    };
    SX_HIP(hipMemsetAsync(st, 0, sizeof(CgState), s));
    // b = A (xa^2 .* c) -> r ; rho0 = b.b
    hipLaunchKernelGGL(k_cg_scale_c, dim3(gv), dim3(SX_WG), 0, s, n, xa, c, w);
    SX_TRY(launch_a(w, nullptr, r));
    int n_rho0 = nA;
    if (cs) { // slack columns carry a cost: b += xs^2 .* cs, rho0 recomputed from the completed b
        hipLaunchKernelGGL(k_cg_slack_cost, dim3(gv), dim3(SX_WG), 0, s, m, xs, cs, r, ppq);
        n_rho0 = gv;
    }
    hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(SX_WG), 0, s, st, m, ppq, n_rho0, r, p, z);
    CgState host;
    SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    const double bnrm = sqrt(host.rho[0]);
    result->b_norm = bnrm;
    const bool trivial = !(bnrm > tol); // legacy scipy: ||b|| <= tol -> return x0 = 0 at once
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, st, tol * bnrm, trivial ? 1 : 0);

    int launched = 0;
    bool finished = trivial;
    if (!trivial && !(bnrm < tol * bnrm)) { // scipy tests ||r|| < atol before the first iteration too
        auto enqueue_iteration = [&](int par) {
            launch_at(p, xa, w);
            (void)launch_a(w, p, q);
            hipLaunchKernelGGL(k_cg_update_zr, dim3(gv), dim3(SX_WG), 0, s, st, par, m, ppq, nA, p, q, z, r, prr);
            hipLaunchKernelGGL(k_cg_update_p, dim3(gv), dim3(SX_WG), 0, s, st, par, m, prr, gv, r, p);
        };
        // the loop is launch-bound on cache-resident problems (four ~4 us kernels per iteration): a
        // batch of 24 iterations (even, so the rho parity repeats) is captured once into a hipGraph and
        // replayed; the host polls the done flag between batches.  Direct launches serve the tail and
        // are the fallback when capture is unavailable.
        const int batch = 24;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (ctx->opt_graph && maxiter >= 2 * batch && getenv("SX_NO_GRAPH") == nullptr) { // (SX_NO_GRAPH: direct launches, for traces)
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                for (int k = 0; k < batch; ++k) enqueue_iteration(k & 1);
                if (hipStreamEndCapture(s, &graph) != hipSuccess || graph == nullptr ||
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
                    exec = nullptr;
            }
            (void)hipGetLastError();
        }
        int rc_loop = SX_OK;
        while (launched < maxiter && !finished) {
            if (exec && launched + batch <= maxiter) {
                if (hipGraphLaunch(exec, s) != hipSuccess) {
                    sx_set_error("hipGraphLaunch failed in the CG loop");
                    rc_loop = SX_ERR_HIP;
                    break;
                }
                launched += batch;
            } else {
                const int upto = (launched + batch < maxiter) ? launched + batch : maxiter;
                for (; launched < upto; ++launched) enqueue_iteration(launched & 1);
            }
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess) {
                sx_set_error("CG iteration batch failed");
                rc_loop = SX_ERR_HIP;
                break;
            }
            finished = host.done != 0;
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc_loop != SX_OK) return rc_loop;
    }
    SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    result->iters = host.iters;
    result->converged = trivial ? 1 : host.converged;
    result->rel_residual = bnrm > 0 ? sqrt(host.rho[host.iters & 1]) / bnrm : 0.0;

    // ||proj||: columns xa .* (c - A^T z), slack rows -xs .* z.  A^T z is taken with unit scale
    // (dividing xa^2 (A^T z) by xa would break on xa = 0), so w becomes a vector of ones first.
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, st, tol * bnrm, 0); // re-arm the kernels
    hipLaunchKernelGGL(k_cg_ones, dim3(gv), dim3(SX_WG), 0, s, n, w);
    launch_at(z, w, atz);
    hipLaunchKernelGGL(k_cg_proj_cols, dim3(gv), dim3(SX_WG), 0, s, n, xa, c, atz, ppq, proj_cols);
    hipLaunchKernelGGL(k_cg_proj_rows, dim3(gv), dim3(SX_WG), 0, s, m, xs, cs, z, prr, proj_rows);
    hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(SX_WG), 0, s, st, ppq, gv, prr, gv);
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    result->proj_norm = sqrt(host.sumsq);
    return SX_OK;
}

SX_API int sx_projector_norm_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                                 const double *c, double tol, int maxiter, sx_cg_result *result) {
    return sx_projector_dev(ctx, A, xa, xs, c, tol, maxiter, nullptr, nullptr, result);
}

SX_API int sx_projector_norm(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                             const double *c, double tol, int maxiter, sx_cg_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr && result != nullptr, "matrix or result is NULL");
    SX_REQUIRE(xa && xs && c, "xa, xs or c is NULL");
    sx_stage st(ctx);
    void *dxa, *dxs, *dc;
    SX_TRY(st.in(xa, sizeof(double) * A->n, &dxa));
    SX_TRY(st.in(xs, sizeof(double) * A->m, &dxs));
    SX_TRY(st.in(c, sizeof(double) * A->n, &dc));
    return sx_projector_norm_dev(ctx, A, (double *)dxa, (double *)dxs, (double *)dc, tol, maxiter, result);
}

// ------------------------------------------------------------------------------------------------------
// The same CG with the columns of Y sharded over ranks (reference arithmetic
// lp_methods/algorithms.py:183-187).  A rank holds a column block A_loc (m x n_loc, both layouts) with its
// slice of xa and c; every m-vector (z, r, p, q, xs) is replicated and every rank performs the same vector
// updates, so the only exchange per iteration is ONE all-reduce(SUM) of the partial product q = A_loc w_loc --
// issued by the host between sx_cg_shard_local and sx_cg_shard_update on the stream the context runs on
// (torch.distributed: RCCL over xGMI); the dot products are then taken from replicated vectors and need no
// collective at all.  The slack block xs^2 .* p is added once, after the reduction.
namespace {

// q[i] += xs[i]^2 * p[i];  partial[block] = sum p[i]*q[i]
__global__ __launch_bounds__(SX_WG) void k_cg_add_slack(const CgState *st, int64_t m, const double *__restrict__ xs,
                                                        const double *__restrict__ p, double *__restrict__ q,
                                                        double *__restrict__ partial) {
    if (st->done) return;
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double s = xs[i], pi = p[i];
        const double qi = q[i] + (s * s) * pi;
        q[i] = qi;
        acc += pi * qi;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// r = q (the reduced right-hand side), partial = sum r^2
__global__ __launch_bounds__(SX_WG) void k_cg_take_rhs(int64_t m, const double *__restrict__ q, double *__restrict__ r,
                                                       double *__restrict__ partial) {
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double v = q[i];
        r[i] = v;
        acc += v * v;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

} // namespace

struct sx_cg_shard {
    sx_ctx *ctx = nullptr;
    const sx_matrix *A = nullptr;
    const double *xa = nullptr, *xs = nullptr, *c = nullptr, *cs = nullptr;
    double tol = 0.0, bnrm = 0.0;
    int64_t m = 0, n = 0;
    char *block = nullptr; // one allocation: state | partials | z r p q | w atz
    CgState *st = nullptr;
    double *ppq = nullptr, *prr = nullptr, *z = nullptr, *r = nullptr, *p = nullptr, *q = nullptr, *w = nullptr, *atz = nullptr;
    int gT = 1, gA = 1, gv = 1, swzT = 0, swzA = 0, win_run = 0, nA = 1;
    const sx_rowblock *rb = nullptr;
};

namespace {

void shard_at(sx_cg_shard *h, const double *in, const double *scale, double *out) {
    const sx_matrix *A = h->A;
    hipStream_t s = h->ctx->stream;
    if (h->win_run >= 4)
        hipLaunchKernelGGL((k_cg_at_lw<4>), dim3(h->gT), dim3(SX_WG), 0, s, h->st, A->csc_tiles, A->n_csc_tiles, h->swzT,
                           A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, h->m, in, scale, out);
    else if (h->win_run >= 1)
        hipLaunchKernelGGL((k_cg_at_lw<1>), dim3(h->gT), dim3(SX_WG), 0, s, h->st, A->csc_tiles, A->n_csc_tiles, h->swzT,
                           A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, h->m, in, scale, out);
    else
        hipLaunchKernelGGL(k_cg_at, dim3(h->gT), dim3(SX_WG), 0, s, h->st, A->csc_tiles, A->n_csc_tiles, h->swzT, A->csc_ptr,
                           A->csc_idx, A->csc_val, in, scale, out);
}

int shard_a(sx_cg_shard *h, const double *vec, double *out) { // out = A_loc vec, partials of out.out in ppq
    const sx_matrix *A = h->A;
    if (h->rb) return sx_rb_cg_a(h->ctx, h->rb, h->n, h->st, vec, h->xs, nullptr, out, h->ppq, CG_GRID, &h->nA);
    hipLaunchKernelGGL(k_cg_a, dim3(h->gA), dim3(SX_WG), 0, h->ctx->stream, h->st, A->csr_tiles, A->n_csr_tiles, h->swzA,
                       A->csr_ptr, A->csr_idx, A->csr_val, vec, h->xs, static_cast<const double *>(nullptr), out, h->ppq);
    h->nA = h->gA;
    return SX_OK;
}

} // namespace

SX_API int sx_cg_shard_open(sx_ctx *ctx, const sx_matrix *A_loc, const double *xa_loc, const double *xs,
                            const double *c_loc, const double *cs, double tol, sx_cg_shard **out, double **reduce_vec) {
    SX_ENTER(ctx);
    SX_REQUIRE(A_loc && out && reduce_vec, "NULL argument");
    SX_REQUIRE(A_loc->csr_ptr && A_loc->csc_ptr, "the projector needs both layouts of the column block");
    SX_REQUIRE(xa_loc && xs && c_loc, "xa, xs or c is NULL");
    SX_REQUIRE(A_loc->m > 0 && tol >= 0, "empty row space or negative tolerance");
    sx_cg_shard *h = new (std::nothrow) sx_cg_shard();
    if (!h) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    h->ctx = ctx;
    h->A = A_loc;
    h->xa = xa_loc;
    h->xs = xs;
    h->c = c_loc;
    h->cs = cs;
    h->tol = tol;
    const int64_t m = h->m = A_loc->m, n = h->n = A_loc->n;
    h->swzT = sx_csc_swizzle(ctx, A_loc);
    h->swzA = (ctx->opt_xcd_swizzle && A_loc->n_csr_tiles >= 64 && A_loc->csr_imbalance <= SX_SWIZZLE_MAX_IMBALANCE) ? 1 : 0;
    h->gT = grid_for(ctx, A_loc->n_csc_tiles);
    h->gA = grid_for(ctx, A_loc->n_csr_tiles);
    int64_t gv64 = ((m > n ? m : n) + 4 * SX_WG - 1) / (4 * SX_WG);
    h->gv = static_cast<int>(gv64 < 1 ? 1 : (gv64 > 1024 ? 1024 : gv64));
    int rc = sx_window_run_csc(ctx, A_loc, &h->win_run);
    if (rc == SX_OK) rc = sx_rowblock_get(ctx, A_loc, &h->rb);
    const size_t off_ppq = 256, off_prr = off_ppq + sizeof(double) * CG_GRID, off_vec = off_prr + sizeof(double) * CG_GRID;
    const size_t bytes = off_vec + sizeof(double) * (4 * static_cast<size_t>(m) + 2 * static_cast<size_t>(n > 0 ? n : 1)) + 256;
    if (rc == SX_OK && sx_dmalloc(reinterpret_cast<void **>(&h->block), bytes) != hipSuccess) {
        sx_set_error("hipMalloc of %zu bytes failed for the sharded projector", bytes);
        rc = SX_ERR_NOMEM;
    }
    if (rc != SX_OK) {
        delete h;
        return rc;
    }
    struct Guard { // a failed launch below must not leak the handle and its block
        sx_cg_shard *h;
        ~Guard() {
            if (h) {
                (void)sx_dfree(h->block);
                delete h;
            }
        }
    } guard{h};
    h->st = reinterpret_cast<CgState *>(h->block);
    h->ppq = reinterpret_cast<double *>(h->block + off_ppq);
    h->prr = reinterpret_cast<double *>(h->block + off_prr);
    h->z = reinterpret_cast<double *>(h->block + off_vec);
    h->r = h->z + m;
    h->p = h->r + m;
    h->q = h->p + m;
    h->w = h->q + m;
    h->atz = h->w + (n > 0 ? n : 1);
    hipStream_t s = ctx->stream;
    SX_HIP(hipMemsetAsync(h->st, 0, sizeof(CgState), s));
    // this rank's share of b = Y v: A_loc (xa^2 .* c) -> q, to be summed over the ranks by the caller
    if (n > 0) {
        hipLaunchKernelGGL(k_cg_scale_c, dim3(h->gv), dim3(SX_WG), 0, s, n, xa_loc, c_loc, h->w);
        SX_TRY(shard_a(h, h->w, h->q));
    } else {
        SX_HIP(hipMemsetAsync(h->q, 0, sizeof(double) * static_cast<size_t>(m), s));
    }
    SX_HIP(hipGetLastError());
    guard.h = nullptr;
    *out = h;
    *reduce_vec = h->q;
    return SX_OK;
}

SX_API int sx_cg_shard_start(sx_cg_shard *h, double *bnorm_out, int *trivial_out) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_device_guard guard(h->ctx->device);
    hipStream_t s = h->ctx->stream;
    const int64_t m = h->m;
    hipLaunchKernelGGL(k_cg_take_rhs, dim3(h->gv), dim3(SX_WG), 0, s, m, h->q, h->r, h->ppq);
    if (h->cs) hipLaunchKernelGGL(k_cg_slack_cost, dim3(h->gv), dim3(SX_WG), 0, s, m, h->xs, h->cs, h->r, h->ppq);
    hipLaunchKernelGGL(k_cg_init, dim3(h->gv), dim3(SX_WG), 0, s, h->st, m, h->ppq, h->gv, h->r, h->p, h->z);
    CgState host;
    SX_HIP(hipMemcpyAsync(&host, h->st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    h->bnrm = sqrt(host.rho[0]);
    const bool trivial = !(h->bnrm > h->tol) || (h->bnrm < h->tol * h->bnrm);
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, h->st, h->tol * h->bnrm, trivial ? 1 : 0);
    SX_HIP(hipGetLastError());
    if (bnorm_out) *bnorm_out = h->bnrm;
    if (trivial_out) *trivial_out = trivial ? 1 : 0;
    return SX_OK;
}

// w = xa^2 .* (A_loc^T p);  q = A_loc w   -- then the caller all-reduces q (the vector sx_cg_shard_open returned)
SX_API int sx_cg_shard_local(sx_cg_shard *h) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_device_guard guard(h->ctx->device);
    if (h->n > 0) {
        shard_at(h, h->p, h->xa, h->w);
        SX_TRY(shard_a(h, h->w, h->q));
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}

// with q reduced over the ranks: q += xs^2 .* p, alpha, z, r, beta, p  (iteration number = parity of rho's slot)
SX_API int sx_cg_shard_update(sx_cg_shard *h, int parity) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_device_guard guard(h->ctx->device);
    hipStream_t s = h->ctx->stream;
    const int64_t m = h->m;
    hipLaunchKernelGGL(k_cg_add_slack, dim3(h->gv), dim3(SX_WG), 0, s, h->st, m, h->xs, h->p, h->q, h->ppq);
    hipLaunchKernelGGL(k_cg_update_zr, dim3(h->gv), dim3(SX_WG), 0, s, h->st, parity & 1, m, h->ppq, h->gv, h->p, h->q, h->z,
                       h->r, h->prr);
    hipLaunchKernelGGL(k_cg_update_p, dim3(h->gv), dim3(SX_WG), 0, s, h->st, parity & 1, m, h->prr, h->gv, h->r, h->p);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_cg_shard_poll(sx_cg_shard *h, int *done, int64_t *iters) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_device_guard guard(h->ctx->device);
    CgState host;
    SX_HIP(hipMemcpyAsync(&host, h->st, sizeof(host), hipMemcpyDeviceToHost, h->ctx->stream));
    SX_HIP(hipStreamSynchronize(h->ctx->stream));
    if (done) *done = host.done;
    if (iters) *iters = host.iters;
    return SX_OK;
}

// squared norms of this rank's part of the projection: its columns xa .* (c - A_loc^T z) (optionally written to
// proj_cols_loc) and the slack rows (identical on every rank: count them once); converged / iterations of the loop
SX_API int sx_cg_shard_finish(sx_cg_shard *h, double *proj_cols_loc, double *proj_rows, double *cols_sumsq,
                              double *rows_sumsq, sx_cg_result *result) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_device_guard guard(h->ctx->device);
    hipStream_t s = h->ctx->stream;
    const int64_t m = h->m, n = h->n;
    CgState host;
    SX_HIP(hipMemcpyAsync(&host, h->st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    const long long iters = host.iters;
    const int converged = host.converged;
    const double rel = h->bnrm > 0 ? sqrt(host.rho[host.iters & 1]) / h->bnrm : 0.0;
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, h->st, h->tol * h->bnrm, 0); // re-arm the kernels
    double cols = 0.0, rows = 0.0;
    if (n > 0) {
        hipLaunchKernelGGL(k_cg_ones, dim3(h->gv), dim3(SX_WG), 0, s, n, h->w);
        shard_at(h, h->z, h->w, h->atz);
        hipLaunchKernelGGL(k_cg_proj_cols, dim3(h->gv), dim3(SX_WG), 0, s, n, h->xa, h->c, h->atz, h->ppq, proj_cols_loc);
    } else {
        SX_HIP(hipMemsetAsync(h->ppq, 0, sizeof(double) * h->gv, s));
    }
    hipLaunchKernelGGL(k_cg_proj_rows, dim3(h->gv), dim3(SX_WG), 0, s, m, h->xs, h->cs, h->z, h->prr, proj_rows);
    // the two sums separately: k_cg_finish adds them, so run it once per term against a zeroed partner
    SX_HIP(hipMemsetAsync(h->w, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(SX_WG), 0, s, h->st, h->ppq, h->gv, h->w, 1);
    SX_HIP(hipMemcpyAsync(&host, h->st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    cols = host.sumsq;
    hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(SX_WG), 0, s, h->st, h->prr, h->gv, h->w, 1);
    SX_HIP(hipMemcpyAsync(&host, h->st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    rows = host.sumsq;
    SX_HIP(hipGetLastError());
    if (cols_sumsq) *cols_sumsq = cols;
    if (rows_sumsq) *rows_sumsq = rows;
    if (result) {
        memset(result, 0, sizeof(*result));
        result->b_norm = h->bnrm;
        result->iters = iters;
        result->converged = converged;
        result->rel_residual = rel;
    }
    return SX_OK;
}

SX_API int sx_cg_shard_close(sx_cg_shard *h) {
    if (!h) return SX_OK;
    sx_device_guard guard(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->block) (void)sx_dfree(h->block);
    delete h;
    return SX_OK;
}

// ------------------------------------------------------------------------------------------------------
// Free-variable branch of get_projector_Xc (reference lp_methods/algorithms.py:173-180):
//     t = cg(A_2^T A_2, c_free, tol 1e-8, 1000 iterations);   c' = c_std[non-free] - A_1^T (A_2 t)
//     projection of X_1 c' onto {x : A_1 X_1 x + A_2 f = 0}        (a Gurobi QP in the reference)
// with A_2 the free columns and A_1 the other columns of the standard-form matrix [A, I_<].  Everything runs on
// the device: the normal equations of the free block by the same CG kernels with the two products swapped,
// the adjusted cost by one column pass (K1), and the QP as the limit of the ordinary projector in which the free
// columns carry a large scale tau and zero cost -- the penalty form, error O((||Y|| / (tau sigma_min(A_2)))^2);
// tau = 100 x the largest ordinary column scale, corrected for the ratio of column norms.  Parity with Gurobi's
// loose-tolerance answer is unpinned (no Gurobi); tests compare with the exact minimiser of the QP.
namespace {

__global__ __launch_bounds__(SX_WG) void k_free_flags(int64_t nf, const int64_t *__restrict__ free_idx,
                                                      uint8_t *__restrict__ is_free) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < nf;
         k += static_cast<int64_t>(gridDim.x) * SX_WG)
        is_free[free_idx[k]] = 1;
}

// per column: squared norm; maxima as bit patterns of non-negative doubles: [0] all columns, [1] -(min over free
// columns) is awkward, so [1] holds the max of 1 / sqnorm over the free columns; [2] max xa over the non-free
__global__ __launch_bounds__(SX_WG) void k_free_column_stats(int64_t n, const int64_t *__restrict__ colptr,
                                                             const double *__restrict__ val,
                                                             const uint8_t *__restrict__ is_free,
                                                             const double *__restrict__ xa,
                                                             unsigned long long *__restrict__ stats) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        double s = 0.0;
        for (int64_t e = colptr[j]; e < colptr[j + 1]; ++e) s += val[e] * val[e];
        a = fmax(a, s);
        if (is_free[j]) b = fmax(b, s > 0.0 ? 1.0 / s : INFINITY);
        else c = fmax(c, xa[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = fmax(a, __shfl_down(a, o, 64));
        b = fmax(b, __shfl_down(b, o, 64));
        c = fmax(c, __shfl_down(c, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[0], static_cast<unsigned long long>(__double_as_longlong(a)));
        atomicMax(&stats[1], static_cast<unsigned long long>(__double_as_longlong(b)));
        atomicMax(&stats[2], static_cast<unsigned long long>(__double_as_longlong(c)));
    }
}

__global__ __launch_bounds__(SX_WG) void k_free_max(int64_t m, const double *__restrict__ v, unsigned long long *slot) {
    double a = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG)
        a = fmax(a, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_down(a, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, static_cast<unsigned long long>(__double_as_longlong(a)));
}

// xa2 = xa with tau on the free columns; c_adj zeroed there
__global__ __launch_bounds__(SX_WG) void k_free_apply(int64_t n, const uint8_t *__restrict__ is_free, double tau,
                                                      const double *__restrict__ xa, double *__restrict__ xa2,
                                                      double *__restrict__ c_adj) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const bool f = is_free[j] != 0;
        xa2[j] = f ? tau : xa[j];
        if (f) c_adj[j] = 0.0;
    }
}

// cs[i] = row '<' ? -g[i] : 0
__global__ __launch_bounds__(SX_WG) void k_free_slack_cost(int64_t m, const uint8_t *__restrict__ row_lt,
                                                           const double *__restrict__ g, double *__restrict__ cs) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG)
        cs[i] = row_lt[i] ? -g[i] : 0.0;
}

// partial sums of pc[j]^2 over non-free columns (mode 0) or pr[i]^2 over '<' rows (mode 1); masked-out entries are
// also zeroed in place so that the vectors handed back cover the QP's variables only
__global__ __launch_bounds__(SX_WG) void k_free_norm(int64_t n, const uint8_t *__restrict__ flag, int keep_when,
                                                     double *__restrict__ v, double *__restrict__ partial) {
    double acc = 0.0;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        if ((flag[j] != 0) == (keep_when != 0)) acc += v[j] * v[j];
        else v[j] = 0.0;
    }
    const double tot = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

inline double from_bits(unsigned long long b) {
    double d;
    memcpy(&d, &b, sizeof(d));
    return d;
}

} // namespace

SX_API int sx_projector_free_dev(sx_ctx *ctx, const sx_matrix *A, int64_t nf, const int64_t *free_idx, const double *xa,
                                 const double *xs, const double *c, const uint8_t *row_lt, double *proj_cols,
                                 double *proj_rows, sx_cg_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && free_idx && xa && xs && c && row_lt && proj_cols && proj_rows && result, "NULL argument");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the projector needs both layouts of A");
    SX_REQUIRE(nf > 0 && nf <= A->n, "nf must be in 1 .. n");
    const int64_t m = A->m, n = A->n;
    hipStream_t s = ctx->stream;
    sx_stage tmp(ctx); // device temporaries, freed on return
    void *v_isfree, *v_stats, *v_cfree, *v_t, *v_r, *v_p, *v_g, *v_w, *v_cadj, *v_xa2, *v_cs, *v_ones, *v_st, *v_pa, *v_pb;
    SX_TRY(tmp.in(nullptr, static_cast<size_t>(n), &v_isfree));
    SX_TRY(tmp.in(nullptr, 4 * sizeof(unsigned long long), &v_stats));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_cfree));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_t));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_r));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_p));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_g));   // A_2^T w  (nf)
    SX_TRY(tmp.in(nullptr, sizeof(double) * m, &v_w));    // A_2 p, later g = A_2 t  (m)
    SX_TRY(tmp.in(nullptr, sizeof(double) * n, &v_cadj));
    SX_TRY(tmp.in(nullptr, sizeof(double) * n, &v_xa2));
    SX_TRY(tmp.in(nullptr, sizeof(double) * m, &v_cs));
    SX_TRY(tmp.in(nullptr, sizeof(double) * nf, &v_ones));
    SX_TRY(tmp.in(nullptr, 256, &v_st));
    SX_TRY(tmp.in(nullptr, sizeof(double) * CG_GRID, &v_pa));
    SX_TRY(tmp.in(nullptr, sizeof(double) * CG_GRID, &v_pb));
    uint8_t *is_free = static_cast<uint8_t *>(v_isfree);
    unsigned long long *stats = static_cast<unsigned long long *>(v_stats);
    double *c_free = static_cast<double *>(v_cfree), *t = static_cast<double *>(v_t), *r = static_cast<double *>(v_r);
    double *p = static_cast<double *>(v_p), *g = static_cast<double *>(v_g), *w = static_cast<double *>(v_w);
    double *c_adj = static_cast<double *>(v_cadj), *xa2 = static_cast<double *>(v_xa2), *cs = static_cast<double *>(v_cs);
    double *ones = static_cast<double *>(v_ones), *pa = static_cast<double *>(v_pa), *pb = static_cast<double *>(v_pb);
    CgState *st = static_cast<CgState *>(v_st);
    auto g1 = [](int64_t k) {
        int64_t b = (k + 4 * SX_WG - 1) / (4 * SX_WG);
        return dim3(static_cast<unsigned>(b < 1 ? 1 : (b > 1024 ? 1024 : b)));
    };
    SX_HIP(hipMemsetAsync(is_free, 0, static_cast<size_t>(n), s));
    SX_HIP(hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), s));
    SX_HIP(hipMemsetAsync(st, 0, sizeof(CgState), s));
    hipLaunchKernelGGL(k_free_flags, g1(nf), dim3(SX_WG), 0, s, nf, free_idx, is_free);

    // ---- t = cg(A_2^T A_2, c_free): the CG kernels with the row pass first (w = A_2 p, alpha = rho / (w.w))
    sx_matrix *A2 = nullptr;
    SX_TRY(sx_gather_columns_dev(ctx, A, free_idx, nf, &A2));
    struct Guard {
        sx_matrix *M;
        ~Guard() { (void)sx_matrix_destroy(M); }
    } guard{A2};
    SX_TRY(sx_gather_f64_dev(ctx, nf, free_idx, c, c_free));
    const int swzT = sx_csc_swizzle(ctx, A2);
    const int swzA = (ctx->opt_xcd_swizzle && A2->n_csr_tiles >= 64 && A2->csr_imbalance <= SX_SWIZZLE_MAX_IMBALANCE) ? 1 : 0;
    const int gT = grid_for(ctx, A2->n_csc_tiles), gA = grid_for(ctx, A2->n_csr_tiles);
    const dim3 gv = g1(nf);
    hipLaunchKernelGGL(k_cg_ones, gv, dim3(SX_WG), 0, s, nf, ones);
    hipLaunchKernelGGL(k_cg_take_rhs, gv, dim3(SX_WG), 0, s, nf, c_free, r, pa);             // r = c_free, partials r.r
    hipLaunchKernelGGL(k_cg_init, gv, dim3(SX_WG), 0, s, st, nf, pa, static_cast<int>(gv.x), r, p, t);
    CgState host;
    SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    const double bn = sqrt(host.rho[0]);
    const bool trivial = !(bn > 1e-8);
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, st, 1e-8 * bn, trivial ? 1 : 0);
    if (!trivial) {
        for (int it = 0; it < 1000; ++it) {
            hipLaunchKernelGGL(k_cg_a, dim3(gA), dim3(SX_WG), 0, s, st, A2->csr_tiles, A2->n_csr_tiles, swzA, A2->csr_ptr,
                               A2->csr_idx, A2->csr_val, p, static_cast<const double *>(nullptr),
                               static_cast<const double *>(nullptr), w, pa);                 // w = A_2 p, partials w.w
            hipLaunchKernelGGL(k_cg_at, dim3(gT), dim3(SX_WG), 0, s, st, A2->csc_tiles, A2->n_csc_tiles, swzT, A2->csc_ptr,
                               A2->csc_idx, A2->csc_val, w, ones, g);                        // g = A_2^T w
            hipLaunchKernelGGL(k_cg_update_zr, gv, dim3(SX_WG), 0, s, st, it & 1, nf, pa, gA, p, g, t, r, pb);
            hipLaunchKernelGGL(k_cg_update_p, gv, dim3(SX_WG), 0, s, st, it & 1, nf, pb, static_cast<int>(gv.x), r, p);
            if ((it % 25) == 24) {
                SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
                SX_HIP(hipStreamSynchronize(s));
                if (host.done) break;
            }
        }
    }
    // ---- g = A_2 t (m-vector), adjusted cost c - A^T g on the structurals (zero on the free ones), -g on the slacks
    hipLaunchKernelGGL(k_cg_set_atol, dim3(1), dim3(1), 0, s, st, 0.0, 0);                   // re-arm
    hipLaunchKernelGGL(k_cg_a, dim3(gA), dim3(SX_WG), 0, s, st, A2->csr_tiles, A2->n_csr_tiles, swzA, A2->csr_ptr,
                       A2->csr_idx, A2->csr_val, t, static_cast<const double *>(nullptr), static_cast<const double *>(nullptr),
                       w, pa);
    SX_TRY(sx_score_columns_dev(ctx, A, w, c, nullptr, nullptr, nullptr, 0.0, c_adj, nullptr));
    hipLaunchKernelGGL(k_free_slack_cost, g1(m), dim3(SX_WG), 0, s, m, row_lt, w, cs);
    // ---- scale of the free columns
    hipLaunchKernelGGL(k_free_column_stats, g1(n), dim3(SX_WG), 0, s, n, A->csc_ptr, A->csc_val, is_free, xa, stats);
    hipLaunchKernelGGL(k_free_max, g1(m), dim3(SX_WG), 0, s, m, xs, stats + 3);
    unsigned long long hs[4];
    SX_HIP(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    const double norm_max = sqrt(from_bits(hs[0]));
    const double inv_min_free = from_bits(hs[1]); // max of 1 / ||a_j||^2 over the free columns
    const double norm_min_free = (inv_min_free > 0 && inv_min_free < INFINITY) ? sqrt(1.0 / inv_min_free) : 0.0;
    const double scale_max = fmax(from_bits(hs[2]), from_bits(hs[3]));
    const double ratio = norm_min_free > 0 ? norm_max / norm_min_free : 1.0;
    const double tau = 100.0 * fmax(scale_max, 1e-300) * fmax(1.0, ratio);
    hipLaunchKernelGGL(k_free_apply, g1(n), dim3(SX_WG), 0, s, n, is_free, tau, xa, xa2, c_adj);
    // ---- the penalised projector, then the norm over the QP's own variables (non-free columns, '<' rows)
    SX_TRY(sx_projector_std_dev(ctx, A, xa2, xs, c_adj, cs, 1e-8, 1000, proj_cols, proj_rows, result));
    const dim3 gn = g1(n), gm = g1(m);
    hipLaunchKernelGGL(k_free_norm, gn, dim3(SX_WG), 0, s, n, is_free, 0, proj_cols, pa);
    hipLaunchKernelGGL(k_free_norm, gm, dim3(SX_WG), 0, s, m, row_lt, 1, proj_rows, pb);
    hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(SX_WG), 0, s, st, pa, static_cast<int>(gn.x), pb, static_cast<int>(gm.x));
    SX_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    SX_HIP(hipGetLastError());
    result->proj_norm = sqrt(host.sumsq);
    return SX_OK;
}
