// K16: bounded revised primal simplex on the device for the restricted re-solves of the crossover
// (reference: the solve_lp / solve_mcf calls at lp_methods/algorithms.py:69-74 and
// network_methods/net_manager.py:222,468, which go to Gurobi's primal / network simplex with a warm
// basis; backend seam solver_caller/solving.py:32-68).
//
//   min c^T x   s.t.  A x + s = b,  l <= x <= u,   s_i = 0 ('=' rows)  or  s_i >= 0 ('<' rows)
//
// Variables 0..n-1 are the structural columns of A, variables n..n+m-1 the logical (slack) columns
// e_i, which are implicit.  The basis inverse is kept *explicit and dense* in HBM (m x m doubles,
// column-major): with 288 GB per GPU that is viable for the sub-problems the crossover produces
// (m <= 16384 -> 2 GiB) and turns FTRAN / the pivot row / the update into streaming kernels:
//
//   price        rc_j = c_j - a_j^T y for all structurals -- the same sequential CSC segment walk
//                as K1/K10 -- plus the logicals, in one launch; Dantzig rule (Bland's rule after 100
//                consecutive degenerate pivots); one partial per workgroup
//   ftran        every workgroup reduces the partials to the entering column q (same deterministic
//                reduction everywhere, no extra launch), then d = Binv a_q: linear combination of the
//                few columns of Binv that a_q touches
//   ratio        bounded ratio test incl. the entering variable's own bound flip; one workgroup,
//                wavefront min-reductions, ties to the larger pivot (Bland: to the smaller index)
//   update       x_B -= t*dir*d;  y += (rc_q/d_r) * rho_r;  Binv -= dhat rho_r^T   (rank one, the
//                only O(m^2) step: 16 m^2 bytes of HBM traffic per pivot)
//
// Scalars (entering column, step, leaving row, flags) live in a device struct; every kernel is a
// no-op once `done` is set, so the host enqueues pivots in batches of 32 and polls.  Every 64 pivots
// x_B = Binv (b - N x_N) and y = Binv^T c_B are recomputed from scratch.  Phase 1 relaxes the bounds
// of infeasible basic logicals and minimises their distance to feasibility; a warm basis
// (Gurobi-style codes) is installed by pivoting its structural columns into the identity, picking
// the largest available pivot each time, and is dropped if it turns out singular or infeasible.
// A session (sx_simplex_session_*) keeps Binv between the solves of a column-generation sequence:
// when the warm basis is exactly the basis the kept inverse belongs to (columns matched through
// caller-supplied stable ids), it is installed without any pivot.
//
// Multi-GPU note: with columns sharded, `price` partials are per rank and the selection becomes the
// all-gather + lexicographic min of smart_crossover/distributed.py (one small exchange per pivot);
// this file implements the single-device solver.
#include "sx_internal.h"
#include "sx_segwalk.h"

#include <cmath>
#include <unordered_map>
#include <vector>

namespace {

constexpr int SPX_CHUNK = 4096;
constexpr int SPX_GRID = 1024;   // price workgroups (= partials) at most
constexpr double PIV_TOL = 1e-9; // smallest |pivot| accepted
constexpr int BLAND_AFTER = 100;

enum : int { ST_BASIC = 0, ST_LOWER = -1, ST_UPPER = -2, ST_FREE = -3 };

struct SpxState {
    long long iters;
    int done;  // 0 running, 1 no candidate (optimal for the current costs), 2 unbounded, 3 breakdown
    int q, dir, r, flip, bland, degenerate_run;
    double rc_q, t, alpha;
    double obj;
    long long n_relaxed; // phase 1: basic logicals still outside their true bounds
};

struct Spx {
    int64_t m, n;
    // per variable (n + m)
    int8_t *status;
    uint8_t *relaxed; // logical with phase-1 bounds
    double *lo, *up, *cost, *x;
    // true data of the logicals / costs for the phase switch
    const uint8_t *row_lt;
    const double *c_true;
    // per row
    int32_t *head;
    double *y, *d, *rho, *rhs, *b;
    double *Binv;
    SpxState *st;
    // pricing partials
    double *p_score, *p_rc;
    long long *p_idx;
};

struct StageDot {
    const double *__restrict__ vec;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const { o[0] = v * vec[i]; }
};

__device__ __forceinline__ void better(double &s, long long &j, double &rc, double s2, long long j2, double rc2) {
    if (j2 >= 0 && (j < 0 || s2 > s || (s2 == s && j2 < j))) {
        s = s2;
        j = j2;
        rc = rc2;
    }
}

__device__ __forceinline__ void block_best(double s, long long j, double rc, double *ps, long long *pj, double *prc,
                                           int slot) {
    __shared__ double ss[SX_WG / 64], srcv[SX_WG / 64];
    __shared__ long long sj[SX_WG / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double s2 = __shfl_down(s, o, 64), r2 = __shfl_down(rc, o, 64);
        const long long j2 = __shfl_down(j, o, 64);
        better(s, j, rc, s2, j2, r2);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        ss[threadIdx.x >> 6] = s;
        sj[threadIdx.x >> 6] = j;
        srcv[threadIdx.x >> 6] = rc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SX_WG / 64; ++w) better(s, j, rc, ss[w], sj[w], srcv[w]);
        ps[slot] = s;
        pj[slot] = j;
        prc[slot] = rc;
    }
}

// attractiveness of a non-basic variable with reduced cost rc (0 = not a candidate)
__device__ __forceinline__ double score_of(int st, double rc, double lo, double up, double tol, int bland) {
    if (st == ST_BASIC || lo == up) return 0.0;
    double s = 0.0;
    if (st == ST_LOWER) s = (rc < -tol) ? -rc : 0.0;
    else if (st == ST_UPPER) s = (rc > tol) ? rc : 0.0;
    else s = (fabs(rc) > tol) ? fabs(rc) : 0.0;
    return (bland && s > 0.0) ? 1.0 : s;
}

// ------------------------------------------------------------------ pricing
// one launch prices everything: workgroups [0, gP) walk the structural columns, [gP, gridDim) the logicals
__global__ __launch_bounds__(SX_WG) void k_spx_price(Spx P, const int64_t *__restrict__ tiles, int64_t ntiles,
                                                     const int64_t *__restrict__ colptr,
                                                     const int32_t *__restrict__ rowidx,
                                                     const double *__restrict__ val, double tol, int gP) {
    if (P.st->done) return;
    __shared__ sx_walk_lds<1, SPX_CHUNK> lds;
    const int bland = P.st->bland;
    double s = 0.0, rc = 0.0;
    long long j = -1;
    if (static_cast<int>(blockIdx.x) >= gP) { // logical columns e_i: rc = cost - y_i
        const int gL = gridDim.x - gP;
        for (int64_t i = static_cast<int64_t>(blockIdx.x - gP) * SX_WG + threadIdx.x; i < P.m;
             i += static_cast<int64_t>(gL) * SX_WG) {
            const int64_t k = P.n + i;
            const double r = P.cost[k] - P.y[i];
            const double sc = score_of(P.status[k], r, P.lo[k], P.up[k], tol, bland);
            if (sc > 0.0) better(s, j, rc, sc, k, r);
        }
        block_best(s, j, rc, P.p_score, P.p_idx, P.p_rc, blockIdx.x);
        return;
    }
    for (int64_t t = blockIdx.x; t < ntiles; t += gP) {
        double acc[1];
        int64_t col;
        bool valid;
        sx_segwalk<1, SPX_CHUNK>(tiles, t, colptr, rowidx, val, StageDot{P.y}, lds, col, valid, acc);
        if (valid) {
            const double r = P.cost[col] - acc[0];
            const double sc = score_of(P.status[col], r, P.lo[col], P.up[col], tol, bland);
            if (sc > 0.0) better(s, j, rc, sc, col, r);
        }
    }
    block_best(s, j, rc, P.p_score, P.p_idx, P.p_rc, blockIdx.x);
}

// ------------------------------------------------------------------ ftran: d = Binv * a_q
// In the pivot loop (q_override < 0) every workgroup first repeats the selection of the entering column
// from the pricing partials -- a deterministic reduction, so all arrive at the same q without another
// launch -- and workgroup 0 records it in the state.
__global__ __launch_bounds__(SX_WG) void k_spx_ftran(Spx P, const int64_t *__restrict__ colptr,
                                                     const int32_t *__restrict__ rowidx,
                                                     const double *__restrict__ val, int q_override, int nslots) {
    __shared__ double sh_s[SX_WG / 64], sh_rc[SX_WG / 64];
    __shared__ long long sh_j[SX_WG / 64];
    int64_t q = q_override;
    if (q_override < 0) {
        SpxState *st = P.st;
        if (st->done) return;
        double s = 0.0, rc = 0.0;
        long long j = -1;
        for (int k = threadIdx.x; k < nslots; k += SX_WG) better(s, j, rc, P.p_score[k], P.p_idx[k], P.p_rc[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double s2 = __shfl_down(s, o, 64), r2 = __shfl_down(rc, o, 64);
            const long long j2 = __shfl_down(j, o, 64);
            better(s, j, rc, s2, j2, r2);
        }
        if ((threadIdx.x & 63) == 0) {
            sh_s[threadIdx.x >> 6] = s;
            sh_j[threadIdx.x >> 6] = j;
            sh_rc[threadIdx.x >> 6] = rc;
        }
        __syncthreads();
        s = sh_s[0];
        j = sh_j[0];
        rc = sh_rc[0];
        for (int w = 1; w < SX_WG / 64; ++w) better(s, j, rc, sh_s[w], sh_j[w], sh_rc[w]);
        const bool none = j < 0 || s <= 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (none) {
                st->done = 1;
                st->q = -1;
            } else {
                st->q = static_cast<int>(j);
                st->rc_q = rc;
                const int stq = P.status[j];
                st->dir = (stq == ST_LOWER) ? 1 : (stq == ST_UPPER) ? -1 : (rc < 0 ? 1 : -1);
            }
        }
        if (none) return;
        q = j;
    }
    const int64_t m = P.m;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        double acc = 0.0;
        if (q >= P.n) {
            acc = P.Binv[i + (q - P.n) * m];
        } else {
            for (int64_t e = colptr[q]; e < colptr[q + 1]; ++e) acc = fma(val[e], P.Binv[i + rowidx[e] * m], acc);
        }
        P.d[i] = acc;
    }
}

// ------------------------------------------------------------------ ratio test (one workgroup)
__global__ __launch_bounds__(1024) void k_spx_ratio(Spx P) {
    SpxState *st = P.st;
    if (st->done) return;
    const int q = st->q, dir = st->dir, bland = st->bland;
    double best_t = INFINITY, best_piv = 0.0;
    int best_r = -1;
    for (int i = threadIdx.x; i < P.m; i += 1024) {
        const double delta = dir * P.d[i];
        const int k = P.head[i];
        const double xi = P.x[k];
        double ratio = INFINITY;
        if (delta > PIV_TOL) {
            if (P.lo[k] > -INFINITY) ratio = (xi - P.lo[k]) / delta;
        } else if (delta < -PIV_TOL) {
            if (P.up[k] < INFINITY) ratio = (P.up[k] - xi) / (-delta);
        }
        if (ratio < 0.0) ratio = 0.0; // slightly infeasible basic: degenerate step
        if (ratio < INFINITY) {
            const double piv = fabs(delta);
            bool take;
            if (best_r < 0) take = true;
            else if (bland) take = ratio < best_t || (ratio == best_t && k < P.head[best_r]);
            else take = ratio < best_t || (ratio == best_t && piv > best_piv);
            if (take) {
                best_t = ratio;
                best_piv = piv;
                best_r = i;
            }
        }
    }
    __shared__ double st_t[16], st_p[16];
    __shared__ int st_r[16];
    for (int o = 32; o > 0; o >>= 1) {
        const double t2 = __shfl_down(best_t, o, 64), p2 = __shfl_down(best_piv, o, 64);
        const int r2 = __shfl_down(best_r, o, 64);
        bool take = false;
        if (r2 >= 0) {
            if (best_r < 0) take = true;
            else if (bland) take = t2 < best_t || (t2 == best_t && P.head[r2] < P.head[best_r]);
            else take = t2 < best_t || (t2 == best_t && p2 > best_piv);
        }
        if (take) {
            best_t = t2;
            best_piv = p2;
            best_r = r2;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        st_t[threadIdx.x >> 6] = best_t;
        st_p[threadIdx.x >> 6] = best_piv;
        st_r[threadIdx.x >> 6] = best_r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            const int r2 = st_r[w];
            if (r2 < 0) continue;
            bool take;
            if (best_r < 0) take = true;
            else if (bland) take = st_t[w] < best_t || (st_t[w] == best_t && P.head[r2] < P.head[best_r]);
            else take = st_t[w] < best_t || (st_t[w] == best_t && st_p[w] > best_piv);
            if (take) {
                best_t = st_t[w];
                best_piv = st_p[w];
                best_r = r2;
            }
        }
        const double range = P.up[q] - P.lo[q]; // inf unless the entering variable is boxed
        if (!(best_t < INFINITY) && !(range < INFINITY)) {
            st->done = 2; // unbounded ray
            return;
        }
        if (range <= best_t) {
            st->flip = 1;
            st->t = range;
            st->r = -1;
            st->alpha = 1.0;
        } else {
            st->flip = 0;
            st->t = best_t;
            st->r = best_r;
            st->alpha = P.d[best_r];
        }
        if (st->t <= 1e-12) {
            if (++st->degenerate_run > BLAND_AFTER) st->bland = 1;
        } else {
            st->degenerate_run = 0;
            st->bland = 0;
        }
    }
}

// rho = row r of Binv (strided gather), before Binv changes
__global__ __launch_bounds__(SX_WG) void k_spx_rho(Spx P) {
    const SpxState *st = P.st;
    if (st->done || st->flip) return;
    const int64_t m = P.m, r = st->r;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < m;
         k += static_cast<int64_t>(gridDim.x) * SX_WG)
        P.rho[k] = P.Binv[r + k * m];
}

// pivot loop: rho as above and, in the same launch, x_B -= t*dir*d ;  y += (rc_q / alpha) * rho
// (element i needs only its own rho[i])
__global__ __launch_bounds__(SX_WG) void k_spx_rho_update(Spx P) {
    const SpxState *st = P.st;
    if (st->done) return;
    const double step = st->t * st->dir;
    const bool pivot = !st->flip;
    const double mult = pivot ? st->rc_q / st->alpha : 0.0;
    const int64_t m = P.m, r = st->r;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int k = P.head[i];
        P.x[k] = P.x[k] - step * P.d[i];
        if (pivot) {
            const double rho_i = P.Binv[r + i * m];
            P.rho[i] = rho_i;
            P.y[i] = P.y[i] + mult * rho_i;
        }
    }
}

// basis bookkeeping of one pivot (single lane)
__device__ __forceinline__ void spx_commit(const Spx &P) {
    SpxState *st = P.st;
    if (st->done) return;
    const int q = st->q;
    P.x[q] = P.x[q] + st->dir * st->t;
    if (st->flip) {
        P.status[q] = (st->dir > 0) ? ST_UPPER : ST_LOWER;
        P.x[q] = (st->dir > 0) ? P.up[q] : P.lo[q];
    } else {
        const int r = st->r;
        const int k = P.head[r];
        const bool to_lower = st->dir * st->alpha > 0; // x_k was decreasing
        if (P.relaxed[k]) { // a phase-1 logical that reached feasibility gets its true bounds back
            const int64_t i = k - P.n;
            P.relaxed[k] = 0;
            P.lo[k] = 0.0;
            P.up[k] = P.row_lt[i] ? INFINITY : 0.0;
            P.cost[k] = 0.0;
            P.x[k] = 0.0;
            P.status[k] = ST_LOWER;
            st->n_relaxed -= 1;
        } else {
            P.status[k] = to_lower ? ST_LOWER : ST_UPPER;
            P.x[k] = to_lower ? P.lo[k] : P.up[k];
        }
        P.status[q] = ST_BASIC;
        P.head[r] = q;
        if (!(fabs(st->alpha) > PIV_TOL)) st->done = 3;
    }
    st->iters += 1;
}

// The bookkeeping stays a launch of its own: folding it into the last workgroup of k_spx_update_binv
// (ticket counter) was measured 3x slower overall -- thousands of atomics on one address cost far more
// than the ~4 us launch they save.
__global__ void k_spx_commit(Spx P) { spx_commit(P); }

// Binv -= dhat * rho^T with dhat_i = d_i/alpha (i != r), dhat_r = (alpha - 1)/alpha
__global__ __launch_bounds__(SX_WG) void k_spx_update_binv(Spx P) {
    const SpxState *st = P.st;
    if (st->done || st->flip) return;
    const int64_t m = P.m, r = st->r;
    const double alpha = st->alpha, inv = 1.0 / alpha;
    const int64_t total = m * m;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t k = e / m, i = e - k * m;
        const double dh = (i == r) ? (alpha - 1.0) * inv : P.d[i] * inv;
        P.Binv[e] = fma(-dh, P.rho[k], P.Binv[e]);
    }
}

// ------------------------------------------------------------------ set-up / refresh kernels
__global__ __launch_bounds__(SX_WG) void k_spx_identity(Spx P) {
    const int64_t m = P.m, total = m * m;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * SX_WG)
        P.Binv[e] = (e / m == e % m) ? 1.0 : 0.0;
}

// cold start of all variables: structurals non-basic at a finite bound (free ones at 0), logicals basic
__global__ __launch_bounds__(SX_WG) void k_spx_init(Spx P, const double *__restrict__ l, const double *__restrict__ u,
                                                    const double *__restrict__ c, int use_cost) {
    const int64_t N = P.n + P.m;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < N;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        P.relaxed[k] = 0;
        if (k < P.n) {
            const double lk = l[k], uk = u[k];
            P.lo[k] = lk;
            P.up[k] = uk;
            P.cost[k] = use_cost ? c[k] : 0.0;
            if (lk > -INFINITY) {
                P.status[k] = ST_LOWER;
                P.x[k] = lk;
            } else if (uk < INFINITY) {
                P.status[k] = ST_UPPER;
                P.x[k] = uk;
            } else {
                P.status[k] = ST_FREE;
                P.x[k] = 0.0;
            }
        } else {
            const int64_t i = k - P.n;
            P.lo[k] = 0.0;
            P.up[k] = P.row_lt[i] ? INFINITY : 0.0;
            P.cost[k] = 0.0;
            P.status[k] = ST_BASIC;
            P.x[k] = 0.0;
            P.head[i] = static_cast<int32_t>(k);
            P.y[i] = 0.0;
        }
    }
}

// non-basic structurals at the bound a warm basis asks for (codes: -1 lower, -2 upper, -3 free/zero)
__global__ __launch_bounds__(SX_WG) void k_spx_apply_vbasis(Spx P, const int8_t *__restrict__ vb) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < P.n;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int code = vb[k];
        if (code == ST_UPPER && P.up[k] < INFINITY) {
            P.status[k] = ST_UPPER;
            P.x[k] = P.up[k];
        } else if (code == ST_LOWER && P.lo[k] > -INFINITY) {
            P.status[k] = ST_LOWER;
            P.x[k] = P.lo[k];
        }
    }
}

// warm start: row for structural column q among rows still held by a logical (largest |d_i|)
__global__ __launch_bounds__(1024) void k_spx_crash_pick(Spx P, int q, const int8_t *__restrict__ cb) {
    SpxState *st = P.st;
    double best = 0.0;
    int br = -1;
    for (int i = threadIdx.x; i < P.m; i += 1024) {
        const int k = P.head[i];
        if (k < P.n) continue;                         // row already given to a structural
        const double a = fabs(P.d[i]);
        // prefer rows whose logical the warm basis marks non-basic; others only if nothing else works
        const double pref = (cb && cb[k - P.n] == 0) ? 1e-6 : 1.0;
        const double w = a * pref;
        if (a > PIV_TOL && (br < 0 || w > best)) {
            best = w;
            br = i;
        }
    }
    __shared__ double sb[16];
    __shared__ int sr[16];
    for (int o = 32; o > 0; o >>= 1) {
        const double b2 = __shfl_down(best, o, 64);
        const int r2 = __shfl_down(br, o, 64);
        if (r2 >= 0 && (br < 0 || b2 > best)) {
            best = b2;
            br = r2;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sb[threadIdx.x >> 6] = best;
        sr[threadIdx.x >> 6] = br;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sr[w] >= 0 && (br < 0 || sb[w] > best)) {
                best = sb[w];
                br = sr[w];
            }
        st->done = 0;
        st->q = q;
        st->r = br;
        st->flip = (br < 0) ? 1 : 0; // flip == 1 makes rho / update_binv skip this column
        st->alpha = (br >= 0) ? P.d[br] : 1.0;
        if (br >= 0) {
            const int k = P.head[br];
            P.status[k] = ST_LOWER;
            P.x[k] = 0.0;
            P.status[q] = ST_BASIC;
            P.head[br] = q;
        }
    }
}

// rhs = b - sum over non-basic structurals a_j x_j   (CSR walk with a masked operand)
struct StageNonbasic {
    const double *__restrict__ x;
    const int8_t *__restrict__ status;
    __device__ __forceinline__ void operator()(double v, int32_t j, double (&o)[1]) const {
        o[0] = (status[j] == ST_BASIC) ? 0.0 : v * x[j];
    }
};
__global__ __launch_bounds__(SX_WG) void k_spx_rhs(Spx P, const int64_t *__restrict__ tiles, int64_t ntiles,
                                                   const int64_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ colidx,
                                                   const double *__restrict__ val) {
    __shared__ sx_walk_lds<1, SPX_CHUNK> lds;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double acc[1];
        int64_t i;
        bool valid;
        sx_segwalk<1, SPX_CHUNK>(tiles, t, rowptr, colidx, val, StageNonbasic{P.x, P.status}, lds, i, valid, acc);
        if (valid) {
            const int64_t k = P.n + i; // a non-basic logical sits at 0 (or at its relaxed bound 0)
            P.rhs[i] = P.b[i] - acc[0] - ((P.status[k] == ST_BASIC) ? 0.0 : P.x[k]);
        }
    }
}

// x_B = Binv rhs  (row i of Binv dotted with rhs; column-major -> lanes walk i, loop over k)
// A workgroup of 16 waves owns 64 rows; wave w sums the columns k = w, w + 16, ... and the 16 partial sums
// are added in wave order (fixed order -> the same x_B on every run).
__global__ __launch_bounds__(1024) void k_spx_xb(Spx P) {
    __shared__ double part[16][64];
    const int64_t m = P.m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * 64; i0 < m; i0 += static_cast<int64_t>(gridDim.x) * 64) {
        const int64_t i = i0 + lane;
        double acc = 0.0;
        if (i < m)
            for (int64_t k = wave; k < m; k += 16) acc = fma(P.Binv[i + k * m], P.rhs[k], acc);
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && i < m) {
            double tot = part[0][lane];
#pragma unroll
            for (int w = 1; w < 16; ++w) tot += part[w][lane];
            P.x[P.head[i]] = tot;
        }
        __syncthreads();
    }
}

// y = Binv^T c_B  (column i of Binv dotted with c_B: contiguous -> one wave per column)
__global__ __launch_bounds__(SX_WG) void k_spx_btran(Spx P) {
    const int64_t m = P.m;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x) >> 6;
    const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * SX_WG) >> 6;
    for (int64_t i = wave; i < m; i += nwaves) {
        double acc = 0.0;
        for (int64_t k = lane; k < m; k += 64) acc = fma(P.Binv[k + i * m], P.cost[P.head[k]], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) P.y[i] = acc;
    }
}

// phase 1: relax basic logicals that violate their true bounds; phase-1 cost = distance to feasibility
__global__ void k_spx_phase1_setup(Spx P, double ftol) {
    long long cnt = 0;
    for (int64_t i = 0; i < P.m; ++i) {
        const int k = P.head[i];
        if (k < P.n) continue;
        const double v = P.x[k];
        if (v > P.up[k] + ftol) { // '=' row with positive residual
            P.lo[k] = 0.0;
            P.up[k] = INFINITY;
            P.cost[k] = 1.0;
            P.relaxed[k] = 1;
            ++cnt;
        } else if (v < P.lo[k] - ftol) {
            P.lo[k] = -INFINITY;
            P.up[k] = 0.0;
            P.cost[k] = -1.0;
            P.relaxed[k] = 1;
            ++cnt;
        }
    }
    P.st->n_relaxed = cnt;
}

// infeasibility of the current basic solution: max bound violation over basic variables, and the
// phase-1 objective (sum over relaxed logicals of |x|)
__global__ __launch_bounds__(1024) void k_spx_measure(Spx P, double *out /* [0]=max violation, [1]=phase-1 objective, [2]=objective */) {
    double viol = 0.0, p1 = 0.0;
    for (int i = threadIdx.x; i < P.m; i += 1024) {
        const int k = P.head[i];
        const double v = P.x[k];
        if (P.relaxed[k]) {
            p1 += fabs(v);
        } else {
            if (v < P.lo[k]) viol = fmax(viol, P.lo[k] - v);
            if (v > P.up[k]) viol = fmax(viol, v - P.up[k]);
        }
    }
    double obj = 0.0;
    for (int64_t k = threadIdx.x; k < P.n; k += 1024) obj = fma(P.c_true[k], P.x[k], obj);
    __shared__ double s0[16], s1[16], s2[16];
    for (int o = 32; o > 0; o >>= 1) {
        viol = fmax(viol, __shfl_down(viol, o, 64));
        p1 += __shfl_down(p1, o, 64);
        obj += __shfl_down(obj, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s0[threadIdx.x >> 6] = viol;
        s1[threadIdx.x >> 6] = p1;
        s2[threadIdx.x >> 6] = obj;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            viol = fmax(viol, s0[w]);
            p1 += s1[w];
            obj += s2[w];
        }
        out[0] = viol;
        out[1] = p1;
        out[2] = obj;
    }
}

// phase switch: true costs for structurals, true bounds (and zero cost) for every logical
__global__ __launch_bounds__(SX_WG) void k_spx_phase2_setup(Spx P) {
    const int64_t N = P.n + P.m;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < N;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        if (k < P.n) {
            P.cost[k] = P.c_true[k];
        } else {
            P.cost[k] = 0.0;
            if (P.relaxed[k]) {
                P.relaxed[k] = 0;
                P.lo[k] = 0.0;
                P.up[k] = P.row_lt[k - P.n] ? INFINITY : 0.0;
            }
        }
    }
}

// structural costs: the true ones (phase 2) or zero (phase 1)
__global__ __launch_bounds__(SX_WG) void k_spx_struct_cost(Spx P, int use_true) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < P.n;
         k += static_cast<int64_t>(gridDim.x) * SX_WG)
        P.cost[k] = use_true ? P.c_true[k] : 0.0;
}

__global__ void k_spx_reset_state(Spx P) {
    SpxState *st = P.st;
    st->done = 0;
    st->bland = 0;
    st->degenerate_run = 0;
    st->flip = 0;
    st->q = -1;
    st->r = -1;
}

// outputs in the reference's conventions
__global__ __launch_bounds__(SX_WG) void k_spx_export(Spx P, double *__restrict__ x_out, double *__restrict__ y_out,
                                                      int8_t *__restrict__ vb, int8_t *__restrict__ cb) {
    const int64_t N = P.n + P.m;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < N;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        if (k < P.n) {
            if (x_out) x_out[k] = P.x[k];
            if (vb) vb[k] = P.status[k];
        } else {
            const int64_t i = k - P.n;
            if (y_out) y_out[i] = P.y[i];
            if (cb) cb[i] = (P.status[k] == ST_BASIC) ? 0 : -1;
        }
    }
}

inline unsigned grid1d(int64_t n, int64_t cap = 4096) {
    int64_t g = (n + SX_WG - 1) / SX_WG;
    if (g > cap) g = cap;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

// per-solve device buffers: one arena allocation carved into 256-byte aligned pieces (a solve used to
// make ~20 hipMalloc / hipFree pairs, milliseconds that a short warm-started re-solve does not have);
// requests beyond the arena fall back to allocations of their own
struct DevBufs {
    std::vector<void *> p;
    char *arena = nullptr;
    size_t arena_bytes = 0, used = 0;
    ~DevBufs() {
        for (void *q : p)
            if (q) (void)hipFree(q);
    }
    int reserve(size_t bytes) {
        void *d = nullptr;
        SX_HIP(hipMalloc(&d, bytes));
        p.push_back(d);
        arena = static_cast<char *>(d);
        arena_bytes = bytes;
        used = 0;
        return SX_OK;
    }
    template <class T>
    int get(size_t count, T **out) {
        const size_t want = (sizeof(T) * (count ? count : 1) + 255) & ~static_cast<size_t>(255);
        if (arena && used + want <= arena_bytes) {
            *out = reinterpret_cast<T *>(arena + used);
            used += want;
            return SX_OK;
        }
        void *d = nullptr;
        SX_HIP(hipMalloc(&d, want));
        p.push_back(d);
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

// install a basis whose inverse is already in P.Binv: every logical non-basic first, then the head
__global__ __launch_bounds__(SX_WG) void k_spx_logicals_out(Spx P) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < P.m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int64_t k = P.n + i;
        P.status[k] = ST_LOWER;
        P.x[k] = 0.0;
    }
}

__global__ __launch_bounds__(SX_WG) void k_spx_install_head(Spx P, const int32_t *__restrict__ head_new) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < P.m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int32_t k = head_new[i];
        P.head[i] = k;
        P.status[k] = ST_BASIC;
    }
}

} // namespace

struct sx_simplex_session {
    sx_ctx *ctx = nullptr;
    int64_t m = 0;
    double *Binv = nullptr;        // device, m x m column-major, owned
    std::vector<int64_t> head_ids; // per basis position: structural id (>= 0) or -(row + 1) for a logical
    bool valid = false;
};

SX_API int sx_simplex_session_create(sx_ctx *ctx, sx_simplex_session **out) {
    SX_REQUIRE(ctx != nullptr && out != nullptr, "ctx or out is NULL");
    sx_simplex_session *s = new (std::nothrow) sx_simplex_session();
    if (!s) {
        sx_set_error("out of host memory");
        return SX_ERR_NOMEM;
    }
    s->ctx = ctx;
    *out = s;
    return SX_OK;
}

SX_API int sx_simplex_session_destroy(sx_simplex_session *session) {
    if (!session) return SX_OK;
    sx_device_guard guard(session->ctx->device);
    if (session->Binv) (void)hipFree(session->Binv);
    delete session;
    return SX_OK;
}

SX_API int sx_simplex_solve_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                                const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol,
                                double *x_out, double *y_out, int8_t *vbasis_out, int8_t *cbasis_out,
                                sx_simplex_result *result) {
    return sx_simplex_solve_session_dev(ctx, nullptr, A, b, c, l, u, row_is_lt, vbasis_in, cbasis_in, nullptr, max_iter,
                                        feas_tol, opt_tol, x_out, y_out, vbasis_out, cbasis_out, result);
}

SX_API int sx_simplex_solve_session_dev(sx_ctx *ctx, sx_simplex_session *session, const sx_matrix *A, const double *b,
                                        const double *c, const double *l, const double *u, const uint8_t *row_is_lt,
                                        const int8_t *vbasis_in, const int8_t *cbasis_in, const int64_t *col_ids,
                                        int64_t max_iter, double feas_tol, double opt_tol, double *x_out,
                                        double *y_out, int8_t *vbasis_out, int8_t *cbasis_out,
                                        sx_simplex_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(session == nullptr || session->ctx == ctx, "the session belongs to another context");
    SX_REQUIRE(A && b && c && l && u && row_is_lt && result, "NULL argument");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the simplex needs both layouts of A");
    SX_REQUIRE((vbasis_in == nullptr) == (cbasis_in == nullptr), "vbasis_in and cbasis_in go together");
    memset(result, 0, sizeof(*result));
    const int64_t m = A->m, n = A->n, N = n + m;
    if (m > 16384) {
        sx_set_error("sx_simplex_solve: m = %lld exceeds the dense-inverse limit of 16384 rows", (long long)m);
        return SX_ERR_UNSUPPORTED;
    }
    if (max_iter <= 0) max_iter = 50 * (m + n) + 1000;
    hipStream_t s = ctx->stream;

    DevBufs mem;
    {
        // everything below except a session-less inverse: 34 B per variable, 44 B per row, pricing partials
        const size_t um = static_cast<size_t>(m), uN = static_cast<size_t>(N);
        size_t bytes = 34 * uN + 48 * um + 24 * (static_cast<size_t>(SPX_GRID) + um / 64 + 64) + 32 * 256 + 4096;
        if (!session) bytes += sizeof(double) * um * um + 256;
        SX_TRY(mem.reserve(bytes));
    }
    Spx P;
    P.m = m;
    P.n = n;
    P.row_lt = row_is_lt;
    P.c_true = c;
    SX_TRY(mem.get(static_cast<size_t>(N), &P.status));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.relaxed));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.lo));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.up));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.cost));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.x));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.head));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.y));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.d));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.rho));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.rhs));
    if (session) { // the inverse outlives the call
        if (session->m != m || session->Binv == nullptr) {
            if (session->Binv) SX_HIP(hipFree(session->Binv));
            session->Binv = nullptr;
            session->valid = false;
            session->m = m;
            SX_HIP(hipMalloc(reinterpret_cast<void **>(&session->Binv),
                             sizeof(double) * (static_cast<size_t>(m) * static_cast<size_t>(m) + 1)));
        }
        P.Binv = session->Binv;
    } else {
        SX_TRY(mem.get(static_cast<size_t>(m) * static_cast<size_t>(m), &P.Binv));
    }
    SX_TRY(mem.get(1, &P.st));
    const int gP = static_cast<int>(A->n_csc_tiles < SPX_GRID ? (A->n_csc_tiles > 0 ? A->n_csc_tiles : 1) : SPX_GRID);
    const int gL = static_cast<int>(grid1d(m, 64));
    SX_TRY(mem.get(static_cast<size_t>(gP + gL), &P.p_score));
    SX_TRY(mem.get(static_cast<size_t>(gP + gL), &P.p_rc));
    SX_TRY(mem.get(static_cast<size_t>(gP + gL), &P.p_idx));
    double *meas = nullptr;
    SX_TRY(mem.get(3, &meas));
    P.b = const_cast<double *>(b);
    SX_HIP(hipMemsetAsync(P.st, 0, sizeof(SpxState), s));

    const unsigned gN = grid1d(N), gM = grid1d(m), gMM = grid1d(m * m, 8192);
    const int gR = static_cast<int>(A->n_csr_tiles < SPX_GRID ? (A->n_csr_tiles > 0 ? A->n_csr_tiles : 1) : SPX_GRID);

    auto refresh = [&](bool with_y) {
        hipLaunchKernelGGL(k_spx_rhs, dim3(gR), dim3(SX_WG), 0, s, P, A->csr_tiles, A->n_csr_tiles, A->csr_ptr,
                           A->csr_idx, A->csr_val);
        hipLaunchKernelGGL(k_spx_xb, dim3(grid1d(m * 4, 2048)), dim3(1024), 0, s, P); // one workgroup per 64 rows
        if (with_y) hipLaunchKernelGGL(k_spx_btran, dim3(grid1d(m * 64)), dim3(SX_WG), 0, s, P);
    };
    double host_meas[3] = {0, 0, 0};
    auto measure = [&]() -> int {
        hipLaunchKernelGGL(k_spx_measure, dim3(1), dim3(1024), 0, s, P, meas);
        SX_HIP(hipMemcpyAsync(host_meas, meas, sizeof(host_meas), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        return SX_OK;
    };
    auto cold_start = [&](int use_cost) {
        hipLaunchKernelGGL(k_spx_init, dim3(gN), dim3(SX_WG), 0, s, P, l, u, c, use_cost);
        hipLaunchKernelGGL(k_spx_identity, dim3(gMM), dim3(SX_WG), 0, s, P);
    };

    // ---- starting basis
    bool warm = false, reused = false;
    std::vector<int8_t> vb;
    if (vbasis_in) {
        vb.resize(static_cast<size_t>(n));
        SX_HIP(hipMemcpyAsync(vb.data(), vbasis_in, static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
    }
    if (session && session->valid && vbasis_in && col_ids && static_cast<int64_t>(session->head_ids.size()) == m) {
        // the session's inverse serves when the warm basis names exactly the variables it belongs to
        std::vector<int8_t> cb(static_cast<size_t>(m));
        SX_HIP(hipMemcpyAsync(cb.data(), cbasis_in, static_cast<size_t>(m), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        std::unordered_map<int64_t, int32_t> basic_of_id;
        int64_t n_basic = 0;
        for (int64_t j = 0; j < n; ++j)
            if (vb[static_cast<size_t>(j)] == ST_BASIC) {
                basic_of_id.emplace(col_ids[j], static_cast<int32_t>(j));
                ++n_basic;
            }
        for (int64_t i = 0; i < m; ++i) n_basic += (cb[static_cast<size_t>(i)] == 0) ? 1 : 0;
        std::vector<int32_t> head_new(static_cast<size_t>(m));
        bool ok = n_basic == m;
        for (int64_t i = 0; ok && i < m; ++i) {
            const int64_t id = session->head_ids[static_cast<size_t>(i)];
            if (id >= 0) {
                auto it = basic_of_id.find(id);
                if (it == basic_of_id.end()) ok = false;
                else {
                    head_new[static_cast<size_t>(i)] = it->second;
                    basic_of_id.erase(it); // every basic column serves one position
                }
            } else {
                const int64_t r = -(id + 1);
                if (r >= m || cb[static_cast<size_t>(r)] != 0) ok = false;
                else head_new[static_cast<size_t>(i)] = static_cast<int32_t>(n + r);
            }
        }
        if (ok && basic_of_id.empty()) {
            int32_t *head_dev = nullptr;
            SX_TRY(mem.get(static_cast<size_t>(m), &head_dev));
            SX_HIP(hipMemcpyAsync(head_dev, head_new.data(), sizeof(int32_t) * static_cast<size_t>(m),
                                  hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_spx_init, dim3(gN), dim3(SX_WG), 0, s, P, l, u, c, 1);
            hipLaunchKernelGGL(k_spx_apply_vbasis, dim3(grid1d(n)), dim3(SX_WG), 0, s, P, vbasis_in);
            hipLaunchKernelGGL(k_spx_logicals_out, dim3(gM), dim3(SX_WG), 0, s, P);
            hipLaunchKernelGGL(k_spx_install_head, dim3(gM), dim3(SX_WG), 0, s, P, head_dev);
            hipLaunchKernelGGL(k_spx_reset_state, dim3(1), dim3(1), 0, s, P);
            refresh(true);
            SX_TRY(measure()); // synchronises: head_new may go out of scope afterwards
            reused = warm = host_meas[0] <= feas_tol * 10;
        }
    }
    if (session) session->valid = false; // until this solve has left a basis behind
    if (!reused) cold_start(1);
    if (vbasis_in && !reused) {
        hipLaunchKernelGGL(k_spx_apply_vbasis, dim3(grid1d(n)), dim3(SX_WG), 0, s, P, vbasis_in);
        for (int64_t j = 0; j < n; ++j) {
            if (vb[static_cast<size_t>(j)] != ST_BASIC) continue;
            hipLaunchKernelGGL(k_spx_ftran, dim3(gM), dim3(SX_WG), 0, s, P, A->csc_ptr, A->csc_idx, A->csc_val,
                               static_cast<int>(j), 0);
            hipLaunchKernelGGL(k_spx_crash_pick, dim3(1), dim3(1024), 0, s, P, static_cast<int>(j), cbasis_in);
            hipLaunchKernelGGL(k_spx_rho, dim3(gM), dim3(SX_WG), 0, s, P);
            hipLaunchKernelGGL(k_spx_update_binv, dim3(gMM), dim3(SX_WG), 0, s, P);
        }
        hipLaunchKernelGGL(k_spx_reset_state, dim3(1), dim3(1), 0, s, P);
        refresh(true);
        SX_TRY(measure());
        warm = host_meas[0] <= feas_tol * 10;
        if (!warm) cold_start(1); // singular or infeasible warm basis: start from the logicals
    }
    SX_HIP(hipGetLastError());

    // ---- phases
    SpxState host;
    memset(&host, 0, sizeof(host));
    auto enqueue_pivot = [&]() {
        hipLaunchKernelGGL(k_spx_price, dim3(gP + gL), dim3(SX_WG), 0, s, P, A->csc_tiles, A->n_csc_tiles, A->csc_ptr,
                           A->csc_idx, A->csc_val, opt_tol, gP);
        hipLaunchKernelGGL(k_spx_ftran, dim3(gM), dim3(SX_WG), 0, s, P, A->csc_ptr, A->csc_idx, A->csc_val, -1, gP + gL);
        hipLaunchKernelGGL(k_spx_ratio, dim3(1), dim3(1024), 0, s, P);
        hipLaunchKernelGGL(k_spx_rho_update, dim3(gM), dim3(SX_WG), 0, s, P);
        hipLaunchKernelGGL(k_spx_update_binv, dim3(gMM), dim3(SX_WG), 0, s, P);
        hipLaunchKernelGGL(k_spx_commit, dim3(1), dim3(1), 0, s, P);
    };
    // a batch of 32 pivots = 192 small launches with fixed arguments: captured into a hipGraph and
    // replayed (pivots are launch-bound for small m); direct launches are the fallback.  Capturing and
    // instantiating costs milliseconds, more than a short warm-started re-solve takes altogether, so the
    // graph is only built once a solve has gone through GRAPH_AFTER pivots by direct launches.
    const int batch = 32;
    const int64_t GRAPH_AFTER = 256;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool graph_tried = false;
    auto build_graph = [&]() {
        graph_tried = true;
        if (!ctx->opt_graph) return;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int k = 0; k < batch; ++k) enqueue_pivot();
            if (hipStreamEndCapture(s, &graph) != hipSuccess || graph == nullptr ||
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
                exec = nullptr;
        }
        (void)hipGetLastError();
    };
    struct GraphGuard {
        hipGraph_t &g;
        hipGraphExec_t &e;
        ~GraphGuard() {
            if (e) (void)hipGraphExecDestroy(e);
            if (g) (void)hipGraphDestroy(g);
        }
    } graph_guard{graph, exec};

    int64_t direct_pivots = 0; // over both phases
    auto run_phase = [&](int64_t budget) -> int {
        int64_t done_iters = 0;
        while (true) {
            if (!exec && !graph_tried && direct_pivots >= GRAPH_AFTER) build_graph();
            if (!exec) direct_pivots += batch;
            if (exec) {
                SX_HIP(hipGraphLaunch(exec, s));
            } else {
                for (int k = 0; k < batch; ++k) enqueue_pivot();
            }
            done_iters += batch;
            SX_HIP(hipGetLastError());
            SX_HIP(hipMemcpyAsync(&host, P.st, sizeof(host), hipMemcpyDeviceToHost, s));
            SX_HIP(hipStreamSynchronize(s));
            if (host.done || host.iters >= budget) return SX_OK;
            if ((done_iters & 63) == 0) refresh(true); // numerical hygiene: x_B and y from scratch
        }
    };

    // status codes of sx_simplex_result: 0 optimal, 1 infeasible, 2 unbounded, 3 iteration limit, 4 numerical
    int status = -1;
    int64_t phase1_iters = 0;
    refresh(true);
    SX_TRY(measure());
    if (host_meas[0] > feas_tol) {
        // ---- phase 1: minimise the infeasibility of the relaxed logicals
        hipLaunchKernelGGL(k_spx_struct_cost, dim3(grid1d(n)), dim3(SX_WG), 0, s, P, 0);
        hipLaunchKernelGGL(k_spx_phase1_setup, dim3(1), dim3(1), 0, s, P, feas_tol);
        hipLaunchKernelGGL(k_spx_btran, dim3(grid1d(m * 64)), dim3(SX_WG), 0, s, P);
        SX_TRY(run_phase(max_iter));
        phase1_iters = host.iters;
        if (host.done == 3) status = 4;
        else if (host.done == 0) status = 3;
        else {
            refresh(false);
            SX_TRY(measure());
            // sum of |relaxed logicals| that are still basic; scaled tolerance on the right-hand side
            if (host_meas[1] > feas_tol * (1.0 + static_cast<double>(m)) || host_meas[0] > 10 * feas_tol) status = 1;
        }
        if (status < 0) {
            hipLaunchKernelGGL(k_spx_phase2_setup, dim3(gN), dim3(SX_WG), 0, s, P);
            hipLaunchKernelGGL(k_spx_reset_state, dim3(1), dim3(1), 0, s, P);
            refresh(true);
        }
    }
    if (status < 0) {
        SX_TRY(run_phase(max_iter));
        if (host.done == 1) status = 0;
        else if (host.done == 2) status = 2;
        else if (host.done == 3) status = 4;
        else status = 3;
    }
    refresh(true);
    SX_TRY(measure());
    hipLaunchKernelGGL(k_spx_export, dim3(gN), dim3(SX_WG), 0, s, P, x_out, y_out, vbasis_out, cbasis_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipStreamSynchronize(s));
    result->status = status;
    result->iters = host.iters;
    result->phase1_iters = phase1_iters;
    result->obj = host_meas[2];
    result->max_violation = host_meas[0];
    result->warm_start_used = reused ? 2 : (warm ? 1 : 0);
    if (session) { // remember which variables the inverse left in P.Binv belongs to
        std::vector<int32_t> head_host(static_cast<size_t>(m));
        SX_HIP(hipMemcpy(head_host.data(), P.head, sizeof(int32_t) * static_cast<size_t>(m), hipMemcpyDeviceToHost));
        session->head_ids.resize(static_cast<size_t>(m));
        for (int64_t i = 0; i < m; ++i) {
            const int32_t k = head_host[static_cast<size_t>(i)];
            session->head_ids[static_cast<size_t>(i)] = (k < n) ? (col_ids ? col_ids[k] : 0) : -(static_cast<int64_t>(k - n) + 1);
        }
        session->valid = col_ids != nullptr && status != 4;
    }
    return SX_OK;
}
