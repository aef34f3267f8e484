// First-order stage of the LP re-solve (kernel group K16p): restarted, diagonally preconditioned primal-dual
// hybrid gradient (PDLP: Applegate et al., "Practical large-scale linear programming using primal-dual hybrid
// gradient", NeurIPS 2021) on the device.
//
// What it replaces.  The reference re-solves the perturbed sub-problem with barrier + crossover
// (lp_methods/algorithms.py:50-54 -> solver_caller/gurobi.py:111-115 run_barrier): the barrier carries the
// interior point of the ORIGINAL LP to a point next to the PERTURBED optimum, the crossover then needs few
// pivots.  Rounds 1-2 had no such stage: the device simplex walked 12 pivots per row from the unperturbed
// interior point.  This file is that stage, built from the two sparse products the crossover kernels already
// are (K1: c - A^T y over the column layout, K2: b - A x over the row layout), with the proximal steps fused
// into the walks' epilogues:
//     x+ = clip(x - tau_j (c - A^T y)_j, l, u),   tau_j = (eta / omega) dc_j^2
//     y+ = proj(y + sig_i (b - A (2 x+ - x))_i),  sig_i = (eta * omega) dr_i^2,  proj: y <= 0 on '<' rows
// (dr, dc: Ruiz + Pock-Chambolle scalings of rows and columns, applied as per-coordinate steps so that A is
// never rescaled; eta = 0.9 / ||D_r A D_c||_2 by power iteration; omega: primal weight).  Every 64 iterations
// the KKT error of the iterate and of the running average is formed (two walks with two accumulators each)
// and a single-workgroup kernel takes the restart decision ON THE DEVICE (sufficient / necessary / artificial
// decay, primal weight from the travelled distances); the host replays one hipGraph per 64 iterations and reads
// one status word every few replays.  No atomics, fixed reduction orders: run-to-run reproducible.
//
// Sign convention as everywhere in this library: reduced cost = c - A^T y, dual of a '<' row <= 0.
#include "sx_internal.h"
#include "sx_segwalk.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int PD_CHUNK = 1024;    // staged entries per chunk of the walks (8 KiB of LDS per accumulator)
constexpr int PD_PERIOD = 64;     // iterations between two restart checks
constexpr int PD_NPART = 16;      // partial sums per workgroup and check kernel
constexpr int PD_MAXBLK = 4096;   // partial records reduced by the decision kernel

struct PdState {
    double omega, eta;
    double err_restart, err_prev;
    double inv_k;          // 1 / iterations since the last restart (what the average divides by)
    double tol;
    double bnorm, cnorm;
    // outputs of the last check, candidate 0 = iterate, 1 = average
    double pr[2], du[2], gap[2], pobj[2], dobj[2], err[2];
    long long k_since, total, restarts, checks;
    int do_restart, cand, status, first; // status: 0 running, 1 converged
};

struct StageDot1 {
    const double *__restrict__ vec;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const { o[0] = v * vec[i]; }
};
struct StageDot2 { // iterate and average in one walk
    const double *__restrict__ v0;
    const double *__restrict__ v1;
    double s1;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[2]) const {
        o[0] = v * v0[i];
        o[1] = v * (v1[i] * s1);
    }
};

// ------------------------------------------------------------------------------ the two steps
__global__ __launch_bounds__(SX_WG) void k_pd_x(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                const double *__restrict__ val, const double *__restrict__ y,
                                                const double *__restrict__ c, const double *__restrict__ l,
                                                const double *__restrict__ u, const double *__restrict__ dc2,
                                                const PdState *__restrict__ st, double *__restrict__ x,
                                                double *__restrict__ xbar, double *__restrict__ xsum) {
    __shared__ sx_walk_lds<1, PD_CHUNK> lds;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles || st->status != 0) return; // (converged or at the limit: the rest of the batch is idle)
    double acc[1];
    int64_t j;
    bool valid;
    double cj = 0.0, xj = 0.0, lj = 0.0, uj = 0.0, dj = 0.0, sj = 0.0;
    auto pre = [&](int64_t seg, bool ok) {
        if (ok) {
            cj = c[seg];
            xj = x[seg];
            lj = l[seg];
            uj = u[seg];
            dj = dc2[seg];
            sj = xsum[seg];
        }
    };
    sx_segwalk<1, PD_CHUNK, 0>(tiles, tile, colptr, rowidx, val, StageDot1{y}, lds, j, valid, acc, pre);
    if (!valid) return;
    const double tau = (st->eta / st->omega) * dj;
    double xn = xj - tau * (cj - acc[0]);
    xn = fmin(fmax(xn, lj), uj);
    x[j] = xn;
    xbar[j] = 2.0 * xn - xj;
    xsum[j] = sj + xn;
}

__global__ __launch_bounds__(SX_WG) void k_pd_y(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                const double *__restrict__ val, const double *__restrict__ xbar,
                                                const double *__restrict__ b, const uint8_t *__restrict__ lt,
                                                const double *__restrict__ dr2, const PdState *__restrict__ st,
                                                double *__restrict__ y, double *__restrict__ ysum) {
    __shared__ sx_walk_lds<1, PD_CHUNK> lds;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles || st->status != 0) return;
    double acc[1];
    int64_t i;
    bool valid;
    double bi = 0.0, yi = 0.0, di = 0.0, si = 0.0;
    bool lti = false;
    auto pre = [&](int64_t seg, bool ok) {
        if (ok) {
            bi = b[seg];
            yi = y[seg];
            di = dr2[seg];
            si = ysum[seg];
            lti = lt && lt[seg];
        }
    };
    sx_segwalk<1, PD_CHUNK, 0>(tiles, tile, rowptr, colidx, val, StageDot1{xbar}, lds, i, valid, acc, pre);
    if (!valid) return;
    const double sig = (st->eta * st->omega) * di;
    double yn = yi + sig * (bi - acc[0]);
    if (lti) yn = fmin(yn, 0.0);
    y[i] = yn;
    ysum[i] = si + yn;
}

// ------------------------------------------------------------------------------ the check
// fixed-order workgroup reduction of NV values per lane; result in red[0..NV) of lane 0's view
template <int NV>
__device__ __forceinline__ void pd_block_sum(double (&v)[NV], double *sm /* [4][NV] */, double *out) {
#pragma unroll
    for (int a = 0; a < NV; ++a) {
        double t = v[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        v[a] = t;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < NV; ++a) sm[w * NV + a] = v[a];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int a = 0; a < NV; ++a) out[a] = ((sm[a] + sm[NV + a]) + sm[2 * NV + a]) + sm[3 * NV + a];
}

// rows: primal residual, b^T y, travelled distance in y -- for the iterate (0) and the average (1)
__global__ __launch_bounds__(SX_WG) void k_pd_chk_rows(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                       const int64_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colidx, const double *__restrict__ val,
                                                       const double *__restrict__ x, const double *__restrict__ xsum,
                                                       const double *__restrict__ b, const uint8_t *__restrict__ lt,
                                                       const double *__restrict__ y, const double *__restrict__ ysum,
                                                       const double *__restrict__ yr, const PdState *__restrict__ st,
                                                       double *__restrict__ part) {
    __shared__ sx_walk_lds<2, PD_CHUNK> lds;
    __shared__ double sm[4 * 8];
    const int64_t tile = blockIdx.x;
    double acc[2];
    int64_t i;
    bool valid;
    const double ik = st->inv_k;
    sx_segwalk<2, PD_CHUNK, 0>(tiles, tile, rowptr, colidx, val, StageDot2{x, xsum, ik}, lds, i, valid, acc);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        const double bi = b[i], y0 = y[i], y1 = ysum[i] * ik, yri = yr[i];
        const bool lti = lt && lt[i];
        double r0 = bi - acc[0], r1 = bi - acc[1];
        if (lti) {
            r0 = fmin(r0, 0.0);
            r1 = fmin(r1, 0.0);
        }
        v[0] = r0 * r0;
        v[1] = r1 * r1;
        v[2] = bi * y0;
        v[3] = bi * y1;
        v[4] = (y0 - yri) * (y0 - yri);
        v[5] = (y1 - yri) * (y1 - yri);
        v[6] = bi * bi;
    }
    pd_block_sum<8>(v, sm, part + static_cast<size_t>(blockIdx.x) * PD_NPART);
}

// columns: dual residual, bound part of the dual objective, c^T x, travelled distance in x, ||c - A^T y||^2
__global__ __launch_bounds__(SX_WG) void k_pd_chk_cols(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                       const int64_t *__restrict__ colptr,
                                                       const int32_t *__restrict__ rowidx, const double *__restrict__ val,
                                                       const double *__restrict__ y, const double *__restrict__ ysum,
                                                       const double *__restrict__ c, const double *__restrict__ l,
                                                       const double *__restrict__ u, const double *__restrict__ x,
                                                       const double *__restrict__ xsum, const double *__restrict__ xr,
                                                       const PdState *__restrict__ st, double *__restrict__ part) {
    __shared__ sx_walk_lds<2, PD_CHUNK> lds;
    __shared__ double sm[4 * 12];
    const int64_t tile = blockIdx.x;
    double acc[2];
    int64_t j;
    bool valid;
    const double ik = st->inv_k;
    sx_segwalk<2, PD_CHUNK, 0>(tiles, tile, colptr, rowidx, val, StageDot2{y, ysum, ik}, lds, j, valid, acc);
    double v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        const double cj = c[j], lj = l[j], uj = u[j], x0 = x[j], x1 = xsum[j] * ik, xrj = xr[j];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double rc = cj - acc[k];
            double viol = 0.0, bound = 0.0;
            if (rc > 0.0) {
                if (isinf(lj)) viol = rc;
                else bound = lj * rc;
            } else if (rc < 0.0) {
                if (isinf(uj)) viol = rc;
                else bound = uj * rc;
            }
            v[k] = viol * viol;
            v[2 + k] = bound;
        }
        v[4] = cj * x0;
        v[5] = cj * x1;
        v[6] = (x0 - xrj) * (x0 - xrj);
        v[7] = (x1 - xrj) * (x1 - xrj);
        v[8] = cj * cj;
        const double rc0 = cj - acc[0];
        v[9] = rc0 * rc0;
    }
    pd_block_sum<12>(v, sm, part + static_cast<size_t>(blockIdx.x) * PD_NPART);
}

// one workgroup: sums the partial records in a fixed order, forms the KKT errors, decides
__global__ __launch_bounds__(SX_WG) void k_pd_decide(const double *__restrict__ part_r, int nblk_r,
                                                     const double *__restrict__ part_c, int nblk_c, PdState *st,
                                                     long long max_iter) {
    __shared__ double sm[4 * PD_NPART];
    __shared__ double R[PD_NPART], Cc[PD_NPART];
    for (int pass = 0; pass < 2; ++pass) {
        const double *part = pass ? part_c : part_r;
        const int nblk = pass ? nblk_c : nblk_r;
        double v[PD_NPART];
#pragma unroll
        for (int a = 0; a < PD_NPART; ++a) v[a] = 0.0;
        for (int k = threadIdx.x; k < nblk; k += SX_WG)
#pragma unroll
            for (int a = 0; a < 12; ++a) v[a] += part[static_cast<size_t>(k) * PD_NPART + a];
        pd_block_sum<PD_NPART>(v, sm, pass ? Cc : R);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    PdState s = *st;
    if (s.status != 0) return;
    const bool first = s.first != 0;
    if (first) {
        s.bnorm = sqrt(R[6]);
        s.cnorm = sqrt(Cc[8]);
        const double rcn = sqrt(Cc[9]); // what is left of c at the starting duals: the cost the iteration sees
        double w = 1.0;
        if (rcn > 0.0 && s.bnorm > 0.0) w = rcn / s.bnorm;
        s.omega = fmin(fmax(w, 1e-8), 1e8);
    }
    for (int k = 0; k < 2; ++k) {
        s.pr[k] = sqrt(R[k]);
        s.du[k] = sqrt(Cc[k]);
        s.pobj[k] = Cc[4 + k];
        s.dobj[k] = R[2 + k] + Cc[2 + k];
        s.gap[k] = fabs(s.pobj[k] - s.dobj[k]);
        s.err[k] = sqrt(s.pr[k] * s.pr[k] + s.du[k] * s.du[k] + s.gap[k] * s.gap[k]);
    }
    s.checks += 1;
    s.do_restart = 0;
    if (first) {
        s.first = 0;
        s.err_restart = s.err[0];
        s.err_prev = s.err[0];
        s.cand = 0;
    } else {
        const int cand = (s.err[1] < s.err[0]) ? 1 : 0;
        const double e = s.err[cand];
        bool go = false;
        if (e <= 0.2 * s.err_restart) go = true;
        else if (e <= 0.8 * s.err_restart && e > s.err_prev) go = true;
        else if (static_cast<double>(s.k_since) >= 0.36 * static_cast<double>(s.total) && s.total > 1000) go = true;
        s.err_prev = e;
        s.cand = cand;
        const bool conv = s.pr[cand] <= s.tol * (1.0 + s.bnorm) && s.du[cand] <= s.tol * (1.0 + s.cnorm) &&
                          s.gap[cand] <= s.tol * (1.0 + fabs(s.pobj[cand]) + fabs(s.dobj[cand]));
        if (conv) s.status = 1;
        else if (s.total >= max_iter) s.status = 2;
        if (s.status != 0) go = true; // the better candidate becomes the iterate that is handed back
        if (go) {
            const double dx = sqrt(Cc[6 + cand]), dy = sqrt(R[4 + cand]);
            if (dx > 1e-300 && dy > 1e-300 && !conv) s.omega = exp(0.5 * log(dy / dx) + 0.5 * log(s.omega));
            s.omega = fmin(fmax(s.omega, 1e-10), 1e10);
            s.do_restart = 1;
            s.restarts += 1;
            s.err_restart = e;
        }
    }
    *st = s;
}

// after the decision: the candidate becomes iterate and restart point, the sums start over
__global__ __launch_bounds__(SX_WG) void k_pd_apply(int64_t n, int64_t m, PdState *st, const double *__restrict__ l,
                                                    const double *__restrict__ u, const uint8_t *__restrict__ lt,
                                                    double *__restrict__ x, double *__restrict__ xsum,
                                                    double *__restrict__ xr, double *__restrict__ y,
                                                    double *__restrict__ ysum, double *__restrict__ yr) {
    const bool go = st->do_restart != 0;
    const bool avg = st->cand == 1;
    const double ik = st->inv_k;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (go) {
        if (t < n) {
            // (an average of points of the box can leave it by a rounding: put it back)
            const double v = avg ? fmin(fmax(xsum[t] * ik, l[t]), u[t]) : x[t];
            x[t] = v;
            xr[t] = v;
            xsum[t] = 0.0;
        }
        if (t < m) {
            double v = y[t];
            if (avg) {
                v = ysum[t] * ik;
                if (lt && lt[t]) v = fmin(v, 0.0);
            }
            y[t] = v;
            yr[t] = v;
            ysum[t] = 0.0;
        }
    }
}

// bookkeeping of the counters: after the apply (which still needs the old inv_k), before the next period
__global__ void k_pd_advance(PdState *st, int period, int before) {
    if (st->status != 0 && before) return;
    if (before) { // in front of a period's check: the period has run
        st->k_since += period;
        st->total += period;
        st->inv_k = 1.0 / static_cast<double>(st->k_since);
    } else if (st->do_restart) {
        st->k_since = 0;
        st->inv_k = 1.0;
        st->do_restart = 0;
    }
}

// ------------------------------------------------------------------------------ set-up (plain per-segment loops)
// mode 0: out[s] = max_k |a_k| r[s] o[idx_k];  mode 1: out[s] = sum_k |a_k| r[s] o[idx_k]
__global__ __launch_bounds__(SX_WG) void k_pd_segnorm(int64_t nseg, const int64_t *__restrict__ ptr,
                                                      const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                      const double *__restrict__ mine, const double *__restrict__ other,
                                                      int mode, double *__restrict__ out) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (s >= nseg) return;
    double acc = 0.0;
    const double ms = mine[s];
    for (int64_t k = ptr[s]; k < ptr[s + 1]; ++k) {
        const double a = fabs(val[k]) * ms * other[idx[k]];
        acc = mode ? acc + a : fmax(acc, a);
    }
    out[s] = acc;
}
__global__ __launch_bounds__(SX_WG) void k_pd_rescale(int64_t n, const double *__restrict__ nrm, double *__restrict__ d) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (t < n && nrm[t] > 0.0) d[t] = d[t] / sqrt(nrm[t]);
}
__global__ __launch_bounds__(SX_WG) void k_pd_fill(int64_t n, double v, double *__restrict__ d) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (t < n) d[t] = v;
}
__global__ __launch_bounds__(SX_WG) void k_pd_square(int64_t n, const double *__restrict__ d, double *__restrict__ d2) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (t < n) d2[t] = d[t] * d[t];
}
// power iteration on (D_r A D_c)^T (D_r A D_c):  w = dr2 .* (A (dc .* v));  v' = dc .* (A^T w)
__global__ __launch_bounds__(SX_WG) void k_pd_pow_rows(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                       const int64_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colidx, const double *__restrict__ val,
                                                       const double *__restrict__ v, const double *__restrict__ dr2,
                                                       double *__restrict__ w) {
    __shared__ sx_walk_lds<1, PD_CHUNK> lds;
    double acc[1];
    int64_t i;
    bool valid;
    sx_segwalk<1, PD_CHUNK, 0>(tiles, blockIdx.x, rowptr, colidx, val, StageDot1{v}, lds, i, valid, acc);
    if (valid) w[i] = dr2[i] * acc[0];
}
__global__ __launch_bounds__(SX_WG) void k_pd_pow_cols(const int64_t *__restrict__ tiles, int64_t ntiles,
                                                       const int64_t *__restrict__ colptr,
                                                       const int32_t *__restrict__ rowidx, const double *__restrict__ val,
                                                       const double *__restrict__ w, const double *__restrict__ dc,
                                                       double *__restrict__ vraw, double *__restrict__ part) {
    __shared__ sx_walk_lds<1, PD_CHUNK> lds;
    __shared__ double sm[4];
    double acc[1];
    int64_t j;
    bool valid;
    sx_segwalk<1, PD_CHUNK, 0>(tiles, blockIdx.x, colptr, rowidx, val, StageDot1{w}, lds, j, valid, acc);
    double v[1] = {0.0};
    if (valid) {
        const double t = dc[j] * acc[0]; // = (M^T M u)_j with v = dc .* u handed to the row pass
        vraw[j] = t;
        v[0] = t * t;
    }
    pd_block_sum<1>(v, sm, part + blockIdx.x);
}
// v = dc .* (vraw / ||vraw||); the norm is the sum of the partials (one lane sums them in order)
__global__ __launch_bounds__(SX_WG) void k_pd_pow_norm(int nblk, const double *__restrict__ part, double *__restrict__ nrm) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += part[k];
        nrm[0] = sqrt(s);
    }
}
__global__ __launch_bounds__(SX_WG) void k_pd_pow_scale(int64_t n, const double *__restrict__ vraw,
                                                        const double *__restrict__ dc, const double *__restrict__ nrm,
                                                        double *__restrict__ v) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (t < n) v[t] = nrm[0] > 0.0 ? dc[t] * (vraw[t] / nrm[0]) : 0.0;
}

inline unsigned grid_of(int64_t n) { return static_cast<unsigned>((n + SX_WG - 1) / SX_WG > 0 ? (n + SX_WG - 1) / SX_WG : 1); }

} // namespace

SX_API int sx_pdlp_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l, const double *u,
                       const uint8_t *row_is_lt, const double *x0, const double *y0, int64_t max_iter, double tol,
                       double *x, double *y, sx_pdlp_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && b && c && l && u && x && y && result, "NULL argument");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr && A->csr_tiles && A->csc_tiles, "the first-order stage needs both layouts of A");
    const int64_t m = A->m, n = A->n;
    SX_REQUIRE(m > 0 && n > 0, "empty problem");
    SX_REQUIRE(A->n_csr_tiles <= PD_MAXBLK * 64 && A->n_csc_tiles <= PD_MAXBLK * 64, "problem too large for the check's partial records");
    if (max_iter <= 0) max_iter = 20000;
    max_iter = (max_iter + PD_PERIOD - 1) / PD_PERIOD * PD_PERIOD;
    hipStream_t s = ctx->stream;
    const int gr = static_cast<int>(A->n_csr_tiles), gc = static_cast<int>(A->n_csc_tiles);
    // ---- one block of device memory for the call
    const size_t nn = static_cast<size_t>(n), mm = static_cast<size_t>(m);
    const size_t doubles = 5 * nn + 5 * mm /* xbar xsum xr dc dc2 | ysum yr dr dr2 w */ + nn /* tmp n */ + mm /* tmp m */ +
                           static_cast<size_t>(gr + gc) * PD_NPART + 64;
    char *block = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&block), doubles * sizeof(double) + 1024));
    struct Free {
        void *p;
        ~Free() { (void)sx_dfree(p); }
    } guard{block};
    SX_HIP(hipMemsetAsync(block, 0, doubles * sizeof(double) + 1024, s));
    PdState *st = reinterpret_cast<PdState *>(block);
    double *base = reinterpret_cast<double *>(block + 1024);
    double *xbar = base, *xsum = xbar + nn, *xr = xsum + nn, *dc = xr + nn, *dc2 = dc + nn, *tn = dc2 + nn;
    double *ysum = tn + nn, *yr = ysum + mm, *dr = yr + mm, *dr2 = dr + mm, *w = dr2 + mm, *tm = w + mm;
    double *part_r = tm + mm, *part_c = part_r + static_cast<size_t>(gr) * PD_NPART, *scal = part_c + static_cast<size_t>(gc) * PD_NPART;
    // ---- scalings: 10 sweeps of Ruiz (max norms), one of Pock-Chambolle (1-norms)
    hipLaunchKernelGGL(k_pd_fill, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, 1.0, dc);
    hipLaunchKernelGGL(k_pd_fill, dim3(grid_of(m)), dim3(SX_WG), 0, s, m, 1.0, dr);
    for (int sweep = 0; sweep < 11; ++sweep) {
        const int mode = sweep == 10 ? 1 : 0;
        hipLaunchKernelGGL(k_pd_segnorm, dim3(grid_of(m)), dim3(SX_WG), 0, s, m, A->csr_ptr, A->csr_idx, A->csr_val, dr, dc, mode, tm);
        hipLaunchKernelGGL(k_pd_segnorm, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, A->csc_ptr, A->csc_idx, A->csc_val, dc, dr, mode, tn);
        hipLaunchKernelGGL(k_pd_rescale, dim3(grid_of(m)), dim3(SX_WG), 0, s, m, tm, dr);
        hipLaunchKernelGGL(k_pd_rescale, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, tn, dc);
    }
    hipLaunchKernelGGL(k_pd_square, dim3(grid_of(m)), dim3(SX_WG), 0, s, m, dr, dr2);
    hipLaunchKernelGGL(k_pd_square, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, dc, dc2);
    // ---- ||D_r A D_c||_2 by power iteration (v kept as dc .* u so that the row pass needs no scaling of its own)
    hipLaunchKernelGGL(k_pd_fill, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, 1.0, tn);
    hipLaunchKernelGGL(k_pd_fill, dim3(1), dim3(SX_WG), 0, s, 1, 1.0, scal);
    hipLaunchKernelGGL(k_pd_pow_scale, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, tn, dc, scal, xbar); // v = dc .* 1
    for (int it = 0; it < 40; ++it) {
        hipLaunchKernelGGL(k_pd_pow_rows, dim3(gr), dim3(SX_WG), 0, s, A->csr_tiles, A->n_csr_tiles, A->csr_ptr, A->csr_idx, A->csr_val, xbar, dr2, w);
        hipLaunchKernelGGL(k_pd_pow_cols, dim3(gc), dim3(SX_WG), 0, s, A->csc_tiles, A->n_csc_tiles, A->csc_ptr, A->csc_idx, A->csc_val, w, dc, tn, part_c);
        hipLaunchKernelGGL(k_pd_pow_norm, dim3(1), dim3(SX_WG), 0, s, gc, part_c, scal);
        hipLaunchKernelGGL(k_pd_pow_scale, dim3(grid_of(n)), dim3(SX_WG), 0, s, n, tn, dc, scal, xbar);
    }
    double lam = 0.0; // ||M^T M u|| for the unit vector u of the last sweep ~ sigma_max^2
    SX_HIP(hipMemcpyAsync(&lam, scal, sizeof(double), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    SX_HIP(hipGetLastError());
    const double sigma = lam > 0.0 ? std::sqrt(lam) : 1.0;
    // ---- state and start
    PdState h{};
    h.omega = 1.0;
    h.eta = 0.9 / (1.02 * sigma);
    h.inv_k = 1.0;
    h.tol = tol > 0 ? tol : 1e-8;
    h.first = 1;
    SX_HIP(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    if (x0) SX_HIP(hipMemcpyAsync(x, x0, sizeof(double) * nn, hipMemcpyDeviceToDevice, s));
    else SX_HIP(hipMemsetAsync(x, 0, sizeof(double) * nn, s));
    if (y0) SX_HIP(hipMemcpyAsync(y, y0, sizeof(double) * mm, hipMemcpyDeviceToDevice, s));
    else SX_HIP(hipMemsetAsync(y, 0, sizeof(double) * mm, s));
    SX_HIP(hipMemsetAsync(xsum, 0, sizeof(double) * nn, s));
    SX_HIP(hipMemsetAsync(ysum, 0, sizeof(double) * mm, s));
    auto check = [&]() {
        hipLaunchKernelGGL(k_pd_chk_rows, dim3(gr), dim3(SX_WG), 0, s, A->csr_tiles, A->n_csr_tiles, A->csr_ptr, A->csr_idx, A->csr_val,
                           x, xsum, b, row_is_lt, y, ysum, yr, st, part_r);
        hipLaunchKernelGGL(k_pd_chk_cols, dim3(gc), dim3(SX_WG), 0, s, A->csc_tiles, A->n_csc_tiles, A->csc_ptr, A->csc_idx, A->csc_val,
                           y, ysum, c, l, u, x, xsum, xr, st, part_c);
        hipLaunchKernelGGL(k_pd_decide, dim3(1), dim3(SX_WG), 0, s, part_r, gr, part_c, gc, st, static_cast<long long>(max_iter));
    };
    const unsigned gv = grid_of(n > m ? n : m);
    SX_HIP(hipMemcpyAsync(xr, x, sizeof(double) * nn, hipMemcpyDeviceToDevice, s));
    SX_HIP(hipMemcpyAsync(yr, y, sizeof(double) * mm, hipMemcpyDeviceToDevice, s));
    check(); // errors of the start, primal weight
    auto period = [&]() {
        for (int it = 0; it < PD_PERIOD; ++it) {
            hipLaunchKernelGGL(k_pd_x, dim3(gc), dim3(SX_WG), 0, s, A->csc_tiles, A->n_csc_tiles, A->csc_ptr, A->csc_idx, A->csc_val, y, c, l, u,
                               dc2, st, x, xbar, xsum);
            hipLaunchKernelGGL(k_pd_y, dim3(gr), dim3(SX_WG), 0, s, A->csr_tiles, A->n_csr_tiles, A->csr_ptr, A->csr_idx, A->csr_val, xbar, b,
                               row_is_lt, dr2, st, y, ysum);
        }
        hipLaunchKernelGGL(k_pd_advance, dim3(1), dim3(1), 0, s, st, PD_PERIOD, 1);
        check();
        hipLaunchKernelGGL(k_pd_apply, dim3(gv), dim3(SX_WG), 0, s, n, m, st, l, u, row_is_lt, x, xsum, xr, y, ysum, yr);
        hipLaunchKernelGGL(k_pd_advance, dim3(1), dim3(1), 0, s, st, PD_PERIOD, 0);
    };
    // ---- one hipGraph per period of 64 iterations (+ check, decision, restart)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool use_graph = ctx->opt_graph != 0 && getenv("SX_NO_GRAPH") == nullptr &&
                     (getenv("ROCP_TOOL_LIBRARIES") == nullptr || getenv("SX_GRAPH_UNDER_PROFILER") != nullptr);
    // (rocprofv3 --kernel-trace faults on graph replay: profiles/r03/hipgraph_rocprofv3.md, frames named in profiles/r04/hipgraph_frames.md;
    //  SX_GRAPH_UNDER_PROFILER=1 keeps the graph path there -- to reproduce the fault, not to measure)
    if (use_graph) {
        SX_HIP(hipStreamSynchronize(s));
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            period();
            if (hipStreamEndCapture(s, &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                use_graph = false;
                (void)hipGetLastError();
            }
        } else {
            use_graph = false;
            (void)hipGetLastError();
        }
    }
    struct GraphGuard {
        hipGraph_t &g;
        hipGraphExec_t &e;
        ~GraphGuard() {
            if (e) (void)hipGraphExecDestroy(e);
            if (g) (void)hipGraphDestroy(g);
        }
    } gguard{graph, exec};
    const int64_t periods = max_iter / PD_PERIOD;
    const int poll = getenv("SX_PDLP_POLL") ? std::max(1, atoi(getenv("SX_PDLP_POLL"))) : 8; // periods enqueued between two looks at the state
    for (int64_t p = 0; p < periods;) {
        const int64_t upto = std::min<int64_t>(p + poll, periods);
        for (; p < upto; ++p) {
            if (use_graph) SX_HIP(hipGraphLaunch(exec, s));
            else period();
        }
        SX_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        if (h.status != 0) break;
    }
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    // the iterate in (x, y) is the candidate of the last restart when the run ended on one (converged / limit)
    const int k = h.cand;
    result->status = h.status == 1 ? 0 : 3;
    result->iters = h.total;
    result->restarts = h.restarts;
    result->primal_residual = h.pr[k];
    result->dual_residual = h.du[k];
    result->gap = h.gap[k];
    result->primal_obj = h.pobj[k];
    result->dual_obj = h.dobj[k];
    result->b_norm = h.bnorm;
    result->c_norm = h.cnorm;
    result->step = h.eta;
    result->primal_weight = h.omega;
    return SX_OK;
}
