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
//                as K1/K10 -- plus the logicals, in one launch; Devex reference weights, brought up to
//                date with the previous pivot in the same walk (second product per column: rho^T a_j),
//                or the Dantzig rule ("spx_pricing" 0); Bland's rule after 100 consecutive degenerate
//                pivots; one partial per workgroup
//   ftran+ratio  every workgroup reduces the partials to the entering column q (same deterministic
//                reduction everywhere, no extra launch), then d = Binv a_q: linear combination of the
//                few columns of Binv that a_q touches.  Each lane owns one basis row and evaluates its
//                entry of the bounded ratio test on the spot; one candidate per workgroup goes to memory
//   rho+update   every workgroup reduces the candidates under a strict total order (smaller step, then
//                larger pivot -- Bland: smaller variable index --, then row) and decides between pivot
//                and bound flip of the entering variable, all alike; then rho = row r of the inverse;
//                x_B -= t*dir*d;  y += (rc_q/d_r) * rho; the last workgroup to arrive (ticket, no fences)
//                does the basis bookkeeping from values it loaded beforehand
//   inverse      Binv -= dhat rho^T is the only O(m^2) step (16 m^2 bytes of HBM traffic).  By default it
//                is not done per pivot: the pairs (dhat_s, rho_s) of a batch of 64 pivots are kept
//                (product form: Binv_now = Binv - sum_s dhat_s rho_s^T, so FTRAN and the pivot row need
//                one extra pass over the pending columns, without any recurrence) and k_spx_fold applies
//                them in one pass, in pivot order -- the same fma sequence per entry as 32 rank-one
//                updates, 1/32 of their traffic ("spx_defer" 0 restores the update per pivot)
//
// A pivot is three launches, each a chain of dependent memory round trips of about a microsecond rather
// than arithmetic; loads that do not depend on the entering column / pivot row are issued first.
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

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

namespace {

constexpr int SPX_CHUNK = 4096;
constexpr int SPX_GRID = 1024;   // price workgroups (= partials) at most
constexpr double PIV_TOL = 1e-9; // smallest |pivot| accepted
constexpr int BLAND_AFTER = 100;
constexpr int SPX_DEFER = 64;    // pivots whose inverse updates are held back and folded in together: the fold moves
                                 // 16 m^2 bytes, 46 % of a pivot at m = 2e4 with a batch of 32
constexpr int SPX_REG = 32;      // of the pending columns, those a lane requests before the entering column is known

// ST_FREE: non-basic and not at a bound -- a free variable at 0, or a *superbasic* one at the interior value
// x[k] a starting point gave it (the reference's basis code -3).  Pricing lets it move either way; entering, it
// can travel as far as its own bound in that direction.  A start in which every variable keeps the value of an
// interior point and the simplex pushes the superbasic ones out one by one is the crossover proper.
enum : int { ST_BASIC = 0, ST_LOWER = -1, ST_UPPER = -2, ST_FREE = -3 };

struct SpxState {
    long long iters;
    int done;  // 0 running, 1 no candidate (optimal for the current costs), 2 unbounded, 3 breakdown
    int q, dir, r, flip, bland, degenerate_run;
    double rc_q, t, alpha;
    double range_q; // up[q] - lo[q] of the entering variable (inf unless it is boxed)
    double obj;
    long long n_relaxed; // phase 1: basic logicals still outside their true bounds
    // Devex: what the next pricing pass needs to bring the reference weights up to date with the last pivot
    int dvx_on, dvx_p;       // last step changed the basis; variable that left it
    double dvx_alpha, dvx_wq; // its pivot element; weight of the variable that entered
};

struct Spx {
    int64_t m, n;
    // per variable (n + m)
    int8_t *status;
    uint8_t *relaxed; // logical with phase-1 bounds
    double *lo, *up, *cost, *x;
    double *w; // Devex reference weights (nullptr: Dantzig pricing)
    // true data of the logicals / costs / structural bounds for the phase switch
    const uint8_t *row_lt;
    const double *c_true;
    const double *l_true, *u_true;
    // value a logical keeps when a structural takes its row while a basis is installed (nullptr: it goes to 0)
    double *s_keep;
    // per row
    int32_t *head;
    double *y, *d, *rho, *rhs, *b;
    double *Binv;
    // deferred updates (product form inside a batch): eta columns, pivot rows of the inverse, pivot row ids
    double *E, *R; // m x SPX_DEFER each, column s = pivot s of the batch
    int32_t *er;   // row of pivot s, or -1 when slot s holds no update (bound flip, stopped)
    // ratio test: one candidate per ftran workgroup (3 doubles + 2 ints each); arrival tickets of the
    // two launches that end with a single-workgroup epilogue (ftran -> ratio, rho_update -> commit)
    double *rt_val;
    int32_t *rt_idx;
    unsigned *tickets;
    SpxState *st;
    // pricing partials
    double *p_score, *p_rc;
    long long *p_idx;
};

struct StageDot {
    const double *__restrict__ vec;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const { o[0] = v * vec[i]; }
};

// a_j^T y and a_j^T rho in one walk (Devex: the pivot row entry alpha_rj = rho^T a_j of the last pivot)
struct StageDot2 {
    const double *__restrict__ y;
    const double *__restrict__ rho;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[2]) const {
        o[0] = v * y[i];
        o[1] = v * rho[i];
    }
};

// Devex reference weights (Forrest & Goldfarb 1992, primal form): after a pivot with entering q, leaving p and
// pivot element alpha, every non-basic j gets w_j = max(w_j, (alpha_rj / alpha)^2 w_q) and p gets
// max(w_q / alpha^2, 1); the entering variable is then chosen by rc_j^2 / w_j.  The update belongs to the
// pivot that has just been made but needs alpha_rj for every column -- a second product per column of the
// walk that prices the next pivot anyway -- so it is done there, from (alpha, w_q, p) left in the state.
struct DevexLast {
    int on, p;
    double inv_alpha2_wq; // w_q / alpha^2
};

__device__ __forceinline__ double devex_weight(const DevexLast &L, double w, long long j, double alpha_rj) {
    if (!L.on) return w;
    if (j == L.p) return fmax(L.inv_alpha2_wq, 1.0);
    const double cand = alpha_rj * alpha_rj * L.inv_alpha2_wq;
    // reference framework too old when a weight runs away: start the column over
    return (cand > 1e12) ? 1.0 : fmax(w, cand);
}

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

// ------------------------------------------------------------------ ratio test pieces
// candidate of the bounded ratio test: step length, |pivot|, pivot d_r, basis row, variable held by the row
struct RatioCand {
    double t, piv, dval;
    int row, hk;
};

// strict total order: smaller step; then the larger pivot (Bland: the smaller variable index); then the row
__device__ __forceinline__ bool cand_better(const RatioCand &a, const RatioCand &b, int bland) {
    if (a.row < 0) return false;
    if (b.row < 0) return true;
    if (a.t != b.t) return a.t < b.t;
    if (bland) return a.hk < b.hk;
    if (a.piv != b.piv) return a.piv > b.piv;
    return a.row < b.row;
}

// best candidate of the workgroup, valid in thread 0
__device__ __forceinline__ RatioCand block_best_cand(RatioCand c, int bland, RatioCand *lds /* [SX_WG / 64] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        RatioCand c2;
        c2.t = __shfl_down(c.t, o, 64);
        c2.piv = __shfl_down(c.piv, o, 64);
        c2.dval = __shfl_down(c.dval, o, 64);
        c2.row = __shfl_down(c.row, o, 64);
        c2.hk = __shfl_down(c.hk, o, 64);
        if (cand_better(c2, c, bland)) c = c2;
    }
    __syncthreads(); // lds may still be read from an earlier call
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < SX_WG / 64; ++w)
            if (cand_better(lds[w], c, bland)) c = lds[w];
    return c;
}

__device__ __forceinline__ void put_cand(const Spx &P, unsigned slot, const RatioCand &c) {
    P.rt_val[3 * slot] = c.t;
    P.rt_val[3 * slot + 1] = c.piv;
    P.rt_val[3 * slot + 2] = c.dval;
    P.rt_idx[2 * slot] = c.row;
    P.rt_idx[2 * slot + 1] = c.hk;
}

__device__ __forceinline__ RatioCand get_cand(const Spx &P, unsigned slot) {
    return RatioCand{P.rt_val[3 * slot], P.rt_val[3 * slot + 1], P.rt_val[3 * slot + 2], P.rt_idx[2 * slot],
                     P.rt_idx[2 * slot + 1]};
}

// Every workgroup calls this when it is through with its rows; true in the one that arrives last.  No
// fences: the epilogue that follows (the basis bookkeeping) reads nothing the other workgroups wrote and
// stores to no address they store to -- it only has to come after their loads, which the ticket gives.
// (With fences, ~1 us each, an epilogue could also consume what the others wrote; the ratio test was done
// that way at first and now rides at the head of the next launch instead.  A dozen to a few dozen
// workgroups take a ticket per launch -- unlike the m^2-sized update, where this was tried and lost.)
__device__ __forceinline__ bool last_block_arrives(unsigned *ticket) {
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        is_last = (t == gridDim.x - 1) ? 1 : 0;
        if (is_last) atomicExch(ticket, 0u); // ready for the next launch
    }
    __syncthreads();
    return is_last != 0;
}

// outcome of the ratio test for entering variable q: pivot row or bound flip, step
// `range` = up[q] - lo[q] (inf unless the entering variable is boxed)
struct RatioOutcome {
    int unbounded, flip, r;
    double t, alpha;
};

__device__ __forceinline__ RatioOutcome ratio_outcome(const RatioCand &best, double range) {
    RatioOutcome o;
    const double best_t = (best.row >= 0) ? best.t : INFINITY;
    o.unbounded = (!(best_t < INFINITY) && !(range < INFINITY)) ? 1 : 0; // ray
    o.flip = (range <= best_t) ? 1 : 0;
    o.t = o.flip ? range : best_t;
    o.r = o.flip ? -1 : best.row;
    o.alpha = o.flip ? 1.0 : best.dval;
    return o;
}

// The candidates the preceding k_spx_ftran launch left (one per workgroup, same grid), reduced to the
// winner in every lane of every workgroup: same inputs, strict total order -> same answer everywhere.
__device__ __forceinline__ RatioCand reduce_candidates(const Spx &P, int bland, RatioCand *lds /* [SX_WG / 64 + 1] */) {
    RatioCand best{INFINITY, 0.0, 0.0, -1, 0};
    for (unsigned p = threadIdx.x; p < gridDim.x; p += SX_WG) {
        const RatioCand c = get_cand(P, p);
        if (cand_better(c, best, bland)) best = c;
    }
    best = block_best_cand(best, bland, lds);
    if (threadIdx.x == 0) lds[SX_WG / 64] = best;
    __syncthreads();
    return lds[SX_WG / 64];
}

// ------------------------------------------------------------------ pricing
// one launch prices everything: workgroups [0, gP) walk the structural columns, [gP, gridDim) the logicals
template <bool DEVEX>
__global__ __launch_bounds__(SX_WG) void k_spx_price(Spx P, const int64_t *__restrict__ tiles, int64_t ntiles,
                                                     const int64_t *__restrict__ colptr,
                                                     const int32_t *__restrict__ rowidx,
                                                     const double *__restrict__ val, double tol, int gP) {
    const SpxState *st = P.st;
    if (st->done) return;
    __shared__ sx_walk_lds<DEVEX ? 2 : 1, SPX_CHUNK> lds;
    const int bland = st->bland;
    DevexLast L{0, -1, 0.0};
    if (DEVEX && st->dvx_on) {
        L.on = 1;
        L.p = st->dvx_p;
        L.inv_alpha2_wq = st->dvx_wq / (st->dvx_alpha * st->dvx_alpha);
    }
    double s = 0.0, rc = 0.0;
    long long j = -1;
    // attractiveness: |rc| (Dantzig), rc^2 / w (Devex), 1 for every candidate under Bland's rule
    auto consider = [&](long long k, double r, double alpha_rk) {
        const int stk = P.status[k];
        if (stk == ST_BASIC) return;
        double sc = score_of(stk, r, P.lo[k], P.up[k], tol, bland);
        if (DEVEX) {
            const double w_old = P.w[k], w_new = devex_weight(L, w_old, k, alpha_rk);
            if (w_new != w_old) P.w[k] = w_new;
            if (!bland) sc = sc * sc / w_new;
        }
        if (sc > 0.0) better(s, j, rc, sc, k, r);
    };
    if (static_cast<int>(blockIdx.x) >= gP) { // logical columns e_i: rc = cost - y_i, pivot row entry rho_i
        const int gL = gridDim.x - gP;
        for (int64_t i = static_cast<int64_t>(blockIdx.x - gP) * SX_WG + threadIdx.x; i < P.m;
             i += static_cast<int64_t>(gL) * SX_WG)
            consider(P.n + i, P.cost[P.n + i] - P.y[i], (DEVEX && L.on) ? P.rho[i] : 0.0);
        block_best(s, j, rc, P.p_score, P.p_idx, P.p_rc, blockIdx.x);
        return;
    }
    for (int64_t t = blockIdx.x; t < ntiles; t += gP) {
        double acc[DEVEX ? 2 : 1];
        int64_t col;
        bool valid;
        if constexpr (DEVEX)
            sx_segwalk<2, SPX_CHUNK>(tiles, t, colptr, rowidx, val, StageDot2{P.y, P.rho}, lds, col, valid, acc);
        else
            sx_segwalk<1, SPX_CHUNK>(tiles, t, colptr, rowidx, val, StageDot{P.y}, lds, col, valid, acc);
        if (valid) consider(col, P.cost[col] - acc[0], DEVEX ? acc[1] : 0.0);
    }
    block_best(s, j, rc, P.p_score, P.p_idx, P.p_rc, blockIdx.x);
}

// ------------------------------------------------------------------ ftran: d = Binv * a_q
// In the pivot loop (q_override < 0) every workgroup first repeats the selection of the entering column
// from the pricing partials -- a deterministic reduction, so all arrive at the same q without another
// launch -- and workgroup 0 records it in the state; the ratio test follows in the same launch.
// With q_override >= 0 (installing a warm basis) column q_override is transformed and the candidates are
// for the row, among those still held by a logical, where it has the largest entry (crash_cb: the warm
// basis' row codes, rows whose logical it keeps basic are taken only if nothing else works).
__global__ __launch_bounds__(SX_WG) void k_spx_ftran(Spx P, const int64_t *__restrict__ colptr,
                                                     const int32_t *__restrict__ rowidx,
                                                     const double *__restrict__ val, int q_override, int nslots,
                                                     int pending, const int8_t *__restrict__ crash_cb) {
    __shared__ double sh_s[SX_WG / 64], sh_rc[SX_WG / 64];
    __shared__ long long sh_j[SX_WG / 64];
    __shared__ double coef[SPX_DEFER];
    __shared__ RatioCand sh_c[SX_WG / 64];
    int64_t q = q_override;
    const bool fuse = q_override < 0; // pivot loop: the ratio test rides along
    int dirq = 1, bland = 0;
    RatioCand best{INFINITY, 0.0, 0.0, -1, 0};
    // A pivot is a chain of dependent memory round trips, about a microsecond each, and little else; so
    // everything that does not depend on the entering column is requested here, before it is known: this
    // lane's row of the pending eta columns, the variable its basis row holds, which slots are live.
    const int64_t m = P.m;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; // one row per lane (host: grid = m / SX_WG)
    const bool have = i < m;
    const int hk = have ? P.head[i] : 0;
    double ev[SPX_REG];
#pragma unroll
    for (int s = 0; s < SPX_REG; ++s) ev[s] = (s < pending && have) ? P.E[i + s * m] : 0.0;
    const bool live = threadIdx.x < pending && P.er[threadIdx.x] >= 0;
    if (threadIdx.x < SPX_DEFER) coef[threadIdx.x] = 0.0;
    double xi = 0.0, lok = 0.0, upk = 0.0, range = INFINITY;
    if (q_override < 0) {
        SpxState *st = P.st;
        if (st->done) return;
        bland = st->bland;
        double s = 0.0, rc = 0.0;
        long long j = -1;
        for (int k = threadIdx.x; k < nslots; k += SX_WG) better(s, j, rc, P.p_score[k], P.p_idx[k], P.p_rc[k]);
        if (have) { // second round trip, under way while the selection is reduced
            xi = P.x[hk];
            lok = P.lo[hk];
            upk = P.up[hk];
        }
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
        if (!none) {
            const int stq = P.status[j];
            dirq = (stq == ST_LOWER) ? 1 : (stq == ST_UPPER) ? -1 : (rc < 0 ? 1 : -1);
            // from a bound: the width of the box; from an interior value (superbasic): what is left of it
            range = (stq == ST_FREE) ? ((dirq > 0) ? P.up[j] - P.x[j] : P.x[j] - P.lo[j]) : P.up[j] - P.lo[j];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (none) {
                st->done = 1;
                st->q = -1;
            } else {
                st->q = static_cast<int>(j);
                st->rc_q = rc;
                st->dir = dirq;
                st->range_q = range;
            }
        }
        if (none) return;
        q = j;
    }
    // updates of this batch not yet folded into Binv: Binv_now = Binv - sum_s E_s R_s^T, hence
    // d = Binv a_q - sum_s (R_s . a_q) E_s ; the entries of a_q are walked once, for both sums (the column
    // index is uniform: scalar loads)
    const int qs = __builtin_amdgcn_readfirstlane(static_cast<int>(q));
    const int64_t sm = static_cast<int64_t>(threadIdx.x) * m; // lane s < pending: column s of R
    double acc = 0.0, cs = 0.0;
    if (qs >= P.n) {
        const int64_t k = qs - P.n;
        if (have) acc = P.Binv[i + k * m];
        if (live) cs = P.R[k + sm];
    } else {
        const int64_t e0 = colptr[qs], e1 = colptr[qs + 1];
#pragma unroll 4
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t ri = rowidx[e];
            const double v = val[e];
            if (have) acc = fma(v, P.Binv[i + ri * m], acc);
            if (live) cs = fma(v, P.R[ri + sm], cs);
        }
    }
    if (pending > 0) {
        if (live) coef[threadIdx.x] = cs;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPX_REG; ++s) acc = fma(-coef[s], ev[s], acc); // slots >= pending: 0 * 0
        if (have)
#pragma unroll 8
            for (int s = SPX_REG; s < pending; ++s) acc = fma(-coef[s], P.E[i + s * m], acc); // same order as above
    }
    if (have) {
        P.d[i] = acc;
        if (fuse) { // this row's entry in the ratio test
            const double delta = dirq * acc;
            double ratio = INFINITY;
            if (delta > PIV_TOL) {
                if (lok > -INFINITY) ratio = (xi - lok) / delta;
            } else if (delta < -PIV_TOL) {
                if (upk < INFINITY) ratio = (upk - xi) / (-delta);
            }
            if (ratio < 0.0) ratio = 0.0; // slightly infeasible basic: degenerate step
            if (ratio < INFINITY) best = RatioCand{ratio, fabs(delta), acc, static_cast<int>(i), hk};
        } else if (hk >= P.n && fabs(acc) > PIV_TOL) { // crash: row still held by a logical, usable pivot
            const bool kept = crash_cb && crash_cb[hk - P.n] == 0; // the target basis keeps this logical
            const double pref = kept ? 1e-6 : 1.0;
            // with a starting point the kept logicals are not displaced at all: columns that find no other row
            // stay superbasic at their value
            if (!(kept && P.s_keep))
                best = RatioCand{-fabs(acc) * pref, 0.0, acc, static_cast<int>(i), hk}; // smallest t = largest weight
        }
    }
    // One candidate per workgroup goes to memory; the launch that follows (k_spx_rho_update, k_spx_rho)
    // starts by reducing them -- every workgroup the same way, under a strict total order -- so the ratio
    // test needs neither a launch nor a device-wide rendezvous of its own.
    best = block_best_cand(best, fuse ? bland : 0, sh_c);
    if (threadIdx.x == 0) put_cand(P, blockIdx.x, best);
}

// rho = row r of Binv (strided gather), before Binv changes
//
// With `pending` updates of the batch not yet folded in, row r of the current inverse is
// Binv[r,:] - sum_s E_s[r] R_s[:]; with slot >= 0 the update of this pivot is not applied to Binv but
// recorded as the pair (E_slot = dhat, R_slot = rho) for k_spx_fold.
__device__ __forceinline__ void rho_weights(const Spx &P, int64_t r, int pending, double *w /* LDS */) {
    if (pending > 0) {
        if (threadIdx.x < pending) {
            const int64_t s = threadIdx.x;
            w[s] = (P.er[s] >= 0) ? P.E[r + s * P.m] : 0.0;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double rho_entry(const Spx &P, int64_t r, int64_t k, int pending, const double *w) {
    double v = P.Binv[r + k * P.m];
#pragma unroll 8
    for (int s = 0; s < pending; ++s) v = fma(-w[s], P.R[k + s * P.m], v);
    return v;
}

__device__ __forceinline__ void record_update(const Spx &P, int64_t r, int64_t i, int slot, double rho_i, double alpha) {
    const double inv = 1.0 / alpha;
    P.R[i + slot * P.m] = rho_i;
    P.E[i + slot * P.m] = (i == r) ? (alpha - 1.0) * inv : P.d[i] * inv;
}

// Installing warm-basis column q: the row it takes is the best candidate k_spx_ftran left (every workgroup
// reduces them the same way); workgroup 0 does the bookkeeping, the rest as above.  A column without a
// usable row is skipped (flip = 1 in the state, slot marked empty).
__global__ __launch_bounds__(SX_WG) void k_spx_rho(Spx P, int q, int pending, int slot) {
    __shared__ double w[SPX_DEFER];
    __shared__ RatioCand sh_c[SX_WG / 64 + 1];
    const RatioCand pick = reduce_candidates(P, 0, sh_c);
    const int br = pick.row;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SpxState *st = P.st;
        st->done = 0;
        st->q = q;
        st->r = br;
        st->flip = (br < 0) ? 1 : 0;
        st->alpha = (br >= 0) ? pick.dval : 1.0;
        if (slot >= 0) P.er[slot] = br;
        if (br >= 0) { // nobody reads head / status in this launch
            const int i_log = pick.hk - static_cast<int>(P.n);
            const double keep = (P.s_keep && P.row_lt[i_log]) ? P.s_keep[i_log] : 0.0;
            P.status[pick.hk] = (keep > 0.0) ? ST_FREE : ST_LOWER; // superbasic slack at its value, or at 0
            P.x[pick.hk] = (keep > 0.0) ? keep : 0.0;
            P.status[q] = ST_BASIC;
            P.head[br] = q;
        }
    }
    if (br < 0) return;
    const int64_t m = P.m, r = br;
    const double alpha = pick.dval;
    rho_weights(P, r, pending, w);
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < m;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double rho_k = rho_entry(P, r, k, pending, w);
        P.rho[k] = rho_k;
        if (slot >= 0) record_update(P, r, k, slot, rho_k, alpha);
    }
}

// what the basis bookkeeping of a pivot needs to know, read before the rows are updated so that the
// bookkeeping itself -- run by the workgroup that finishes last -- only stores
struct CommitView {
    int q, dir, flip, r, k;    // entering variable, its direction, bound flip?, pivot row, leaving variable
    double t, alpha, xq, loq, upq, lok, upk, wq;
    double lok_true, upk_true; // bounds the leaving variable gets back when it was relaxed for phase 1
    long long iters, n_relaxed;
    int relaxed_k, row_lt_k, degenerate_run;
};

__device__ __forceinline__ CommitView commit_view(const Spx &P, int q, int dir, const RatioOutcome &o) {
    const SpxState *st = P.st;
    CommitView v;
    v.q = q;
    v.dir = dir;
    v.flip = o.flip;
    v.r = o.r;
    v.t = o.t;
    v.alpha = o.alpha;
    v.degenerate_run = st->degenerate_run;
    v.iters = st->iters;
    v.n_relaxed = st->n_relaxed;
    v.xq = P.x[v.q];
    v.loq = P.lo[v.q];
    v.upq = P.up[v.q];
    v.wq = P.w ? P.w[v.q] : 1.0;
    v.k = v.flip ? 0 : P.head[v.r];
    v.lok = v.upk = v.lok_true = v.upk_true = 0.0;
    v.relaxed_k = v.row_lt_k = 0;
    if (!v.flip) {
        v.lok = P.lo[v.k];
        v.upk = P.up[v.k];
        v.relaxed_k = P.relaxed[v.k];
        if (v.k >= P.n) {
            v.row_lt_k = P.row_lt[v.k - P.n];
            v.upk_true = v.row_lt_k ? INFINITY : 0.0;
        } else if (v.relaxed_k) {
            v.lok_true = P.l_true[v.k];
            v.upk_true = P.u_true[v.k];
        }
    }
    return v;
}

// basis bookkeeping of one pivot (single lane, after every row has been updated)
__device__ __forceinline__ void spx_commit(const Spx &P, const CommitView &v, int slot) {
    SpxState *st = P.st;
    if (slot >= 0) P.er[slot] = v.flip ? -1 : v.r; // does the slot hold an update for k_spx_fold?
    // outcome of the ratio test, for the host, the update per pivot ("spx_defer" 0) and the next launches
    st->flip = v.flip;
    st->t = v.t;
    st->r = v.r;
    st->alpha = v.alpha;
    if (v.t <= 1e-12) {
        st->degenerate_run = v.degenerate_run + 1;
        if (v.degenerate_run + 1 > BLAND_AFTER) st->bland = 1;
    } else {
        st->degenerate_run = 0;
        st->bland = 0;
    }
    const int q = v.q;
    if (v.flip) {
        P.status[q] = (v.dir > 0) ? ST_UPPER : ST_LOWER;
        P.x[q] = (v.dir > 0) ? v.upq : v.loq;
    } else {
        P.x[q] = v.xq + v.dir * v.t;
        const int k = v.k;
        const bool to_lower = v.dir * v.alpha > 0; // x_k was decreasing
        if (v.relaxed_k) {
            // a variable relaxed for phase 1 has reached the bound it violated: it gets its true bounds back
            // and stays there.  Relaxed above its upper bound U it lived in [U, inf) and leaves decreasing,
            // at U; relaxed below its lower bound L it lived in (-inf, L] and leaves increasing, at L.
            P.relaxed[k] = 0;
            P.lo[k] = v.lok_true;
            P.up[k] = v.upk_true;
            P.cost[k] = 0.0; // phase 1: only relaxed variables carry a cost
            P.x[k] = to_lower ? v.upk_true : v.lok_true;
            P.status[k] = to_lower ? ST_UPPER : ST_LOWER;
            st->n_relaxed = v.n_relaxed - 1;
        } else {
            P.status[k] = to_lower ? ST_LOWER : ST_UPPER;
            P.x[k] = to_lower ? v.lok : v.upk;
        }
        P.status[q] = ST_BASIC;
        P.head[v.r] = q;
        if (!(fabs(v.alpha) > PIV_TOL)) st->done = 3;
        st->dvx_p = k;
        st->dvx_alpha = v.alpha;
        st->dvx_wq = v.wq;
    }
    st->dvx_on = v.flip ? 0 : 1;
    st->iters = v.iters + 1;
}

// pivot loop: rho as above and, in the same launch, x_B -= t*dir*d ;  y += (rc_q / alpha) * rho
// (row i needs only its own rho[i]); the workgroup that finishes last does the basis bookkeeping.
// One row per lane (host: grid = m / SX_WG), loads ordered by what they depend on, as in k_spx_ftran.
__global__ __launch_bounds__(SX_WG) void k_spx_rho_update(Spx P, int pending, int slot) {
    __shared__ double w[SPX_DEFER];
    __shared__ RatioCand sh_c[SX_WG / 64 + 1];
    const SpxState *st = P.st;
    const int64_t m = P.m;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    const bool have = i < m;
    // independent of the pivot row
    const int hk = have ? P.head[i] : 0;
    const double di = have ? P.d[i] : 0.0, yi = have ? P.y[i] : 0.0;
    double rv[SPX_REG];
#pragma unroll
    for (int s = 0; s < SPX_REG; ++s) rv[s] = (s < pending && have) ? P.R[i + s * m] : 0.0;
    const bool live = threadIdx.x < pending && P.er[threadIdx.x] >= 0;
    if (threadIdx.x < SPX_DEFER) w[threadIdx.x] = 0.0;
    if (st->done) {
        if (slot >= 0 && blockIdx.x == 0 && threadIdx.x == 0) P.er[slot] = -1; // nothing recorded in this slot
        return;
    }
    // ratio test: the candidates of k_spx_ftran's workgroups, reduced here by every workgroup alike.  The
    // state is not written before the bookkeeping at the end (a workgroup that starts late must read the
    // same `bland` and `done` as the others).
    const int q = st->q, dir = st->dir;
    const double rc_q = st->rc_q;
    const RatioOutcome out = ratio_outcome(reduce_candidates(P, st->bland, sh_c), st->range_q);
    if (out.unbounded) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            P.st->done = 2; // a late workgroup then leaves through the check above: same effect
            if (slot >= 0) P.er[slot] = -1;
        }
        return;
    }
    const double step = out.t * dir;
    const bool pivot = !out.flip;
    const double alpha = out.alpha;
    const double mult = pivot ? rc_q / alpha : 0.0;
    const int64_t r = out.r;
    // second round trip: needs the pivot row / this row's variable
    CommitView view;
    if (threadIdx.x == 0) view = commit_view(P, q, dir, out);
    const double xk = have ? P.x[hk] : 0.0;
    double rho_i = (pivot && have) ? P.Binv[r + i * m] : 0.0;
    if (pivot && pending > 0) {
        if (live) w[threadIdx.x] = P.E[r + threadIdx.x * m];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPX_REG; ++s) rho_i = fma(-w[s], rv[s], rho_i); // slots >= pending: 0 * 0
        if (have)
#pragma unroll 8
            for (int s = SPX_REG; s < pending; ++s) rho_i = fma(-w[s], P.R[i + s * m], rho_i);
    }
    if (have) {
        // the leaving variable is set to its bound by the bookkeeping, which may run in another workgroup:
        // row r does not store to it, so no address is written from two workgroups
        if (!(pivot && i == r)) P.x[hk] = xk - step * di;
        if (pivot) {
            P.rho[i] = rho_i;
            P.y[i] = yi + mult * rho_i;
            if (slot >= 0) {
                const double inv = 1.0 / alpha;
                P.R[i + slot * m] = rho_i;
                P.E[i + slot * m] = (i == r) ? (alpha - 1.0) * inv : di * inv;
            }
        }
    }
    if (last_block_arrives(P.tickets) && threadIdx.x == 0) spx_commit(P, view, slot);
}

// Deferred form of the update below: the `cnt` recorded pairs of a batch go into the inverse in one pass,
// Binv[i,k] -= sum_s E_s[i] R_s[k] (in pivot order, one fma per pair: the same arithmetic as `cnt` rank-one
// updates, with 1/cnt of their HBM traffic).  A workgroup owns 256 rows x FOLD_COLS columns: each lane keeps
// its row of E in registers, the R tile sits in LDS and is read as a broadcast.
constexpr int FOLD_COLS = 64;
__global__ __launch_bounds__(SX_WG) void k_spx_fold(Spx P, int cnt, int col_tiles) {
    __shared__ double rt[SPX_DEFER][FOLD_COLS];
    __shared__ int any_live;
    const int64_t m = P.m;
    if (threadIdx.x == 0) any_live = 0;
    __syncthreads();
    if (threadIdx.x < cnt && P.er[threadIdx.x] >= 0) any_live = 1; // benign race: all writers store 1
    __syncthreads();
    if (!any_live) return;
    const int64_t rb = blockIdx.x / col_tiles, cb = blockIdx.x - rb * col_tiles;
    const int64_t i = rb * SX_WG + threadIdx.x, k0 = cb * FOLD_COLS;
    const int ncol = static_cast<int>((m - k0 < FOLD_COLS) ? (m - k0) : FOLD_COLS);
    for (int e = threadIdx.x; e < SPX_DEFER * FOLD_COLS; e += SX_WG) {
        const int s = e / FOLD_COLS, kk = e - s * FOLD_COLS;
        rt[s][kk] = (s < cnt && kk < ncol && P.er[s] >= 0) ? P.R[k0 + kk + s * m] : 0.0;
    }
    double ev[SPX_DEFER];
#pragma unroll
    for (int s = 0; s < SPX_DEFER; ++s) ev[s] = (s < cnt && i < m) ? P.E[i + s * m] : 0.0;
    __syncthreads();
    if (i >= m) return;
    // four columns at a time: four loads in flight and four independent fma chains per lane (the lane's row of E
    // takes 2 * SPX_DEFER registers, so few waves share a SIMD and each has to keep more bytes in flight)
    double *col = P.Binv + i + k0 * m;
    int kk = 0;
#pragma unroll 1
    for (; kk + 4 <= ncol; kk += 4, col += 4 * m) {
        double v0 = col[0], v1 = col[m], v2 = col[2 * m], v3 = col[3 * m];
#pragma unroll
        for (int s = 0; s < SPX_DEFER; ++s) {
            v0 = fma(-ev[s], rt[s][kk], v0);
            v1 = fma(-ev[s], rt[s][kk + 1], v1);
            v2 = fma(-ev[s], rt[s][kk + 2], v2);
            v3 = fma(-ev[s], rt[s][kk + 3], v3);
        }
        col[0] = v0;
        col[m] = v1;
        col[2 * m] = v2;
        col[3 * m] = v3;
    }
#pragma unroll 1
    for (; kk < ncol; ++kk, col += m) {
        double v0 = col[0];
#pragma unroll
        for (int s = 0; s < SPX_DEFER; ++s) v0 = fma(-ev[s], rt[s][kk], v0);
        col[0] = v0;
    }
}

// A batch's fold as ONE library DGEMM, Binv -= E R^T (m x m, K = SPX_DEFER), on the fp64 matrix cores: the kernel
// above reads an LDS operand per fma and is issue-bound beyond 32 pending updates (2.2 ms for 64 at m = 2e4, the
// traffic alone would be 1.3 ms); rocBLAS does the rank-64 update in 1.46 ms (profiles/r02/spx_fold.md).  Slots in
// which no pivot was recorded (bound flips, the tail of a finished batch) still hold an older batch's columns:
// they are zeroed first.
__global__ __launch_bounds__(SX_WG) void k_spx_fold_mask(Spx P, int cnt) {
    const int sl = blockIdx.y;
    if (sl < cnt && P.er[sl] >= 0) return;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (i < P.m) {
        P.E[i + sl * P.m] = 0.0;
        P.R[i + sl * P.m] = 0.0;
    }
}

// The rank-64 update itself on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), hand-written: Binv is column major,
// so the tile is computed TRANSPOSED -- D[kk][ii] = sum_s R[k0 + kk][s] E[i0 + ii][s] -- which puts the index that is
// contiguous in memory (the row i of Binv) on the MFMA's lane-minor axis: every load and store of the accumulator
// is four whole 128-byte lines (C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg).  A wave owns 16 columns
// x 64 rows of Binv (four tiles that share the R fragment, 16 k-steps each): 16 KB of Binv in and out against
// 10 KB of E / R fragments, which come from L2 -- workgroups are numbered so that the ~1000 that run together
// cover 64 column tiles x 16 row chunks, i.e. 0.5 MB of R and 2 MB of E per XCD's L2.  The update is memory bound
// (16 m^2 bytes) with the matrix pipe at about half load (2 m^2 64 flop at 78 TFLOP/s); results differ from the
// scalar kernel's by the order of the 64 products inside a tile row (the MFMA's own), within the tests' tolerance.
// NOT the default: see the measurement where it is selected.
typedef double spx_v4d __attribute__((ext_vector_type(4)));
constexpr int FM_KT = 64, FM_IC = 16; // column tiles x row chunks of one super-block of workgroups

typedef double spx_v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(SX_WG) void k_spx_fold_mfma(Spx P, int64_t nkt, int64_t nic) {
    const int64_t m = P.m;
    // workgroup -> (column tile kt, row chunk ic): super-blocks of FM_KT x FM_IC, column tiles fastest
    const int64_t per_sb = static_cast<int64_t>(FM_KT) * FM_IC;
    const int64_t sb_per_row = (nkt + FM_KT - 1) / FM_KT;
    const int64_t sb = blockIdx.x / per_sb, w = blockIdx.x - sb * per_sb;
    const int64_t kt = (sb % sb_per_row) * FM_KT + (w % FM_KT);
    const int64_t ic = (sb / sb_per_row) * FM_IC + (w / FM_KT);
    if (kt >= nkt || ic >= nic) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t k0 = kt * 16, i0 = ic * 256 + wave * 64;
    if (i0 >= m) return;
    // A operand: row = lane & 15 -> column k0 + l15 of Binv, k-index = 4 step + (lane >> 4)
    const int64_t kA = (k0 + l15 < m) ? k0 + l15 : m - 1;
    double ra[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) ra[st] = P.R[kA + static_cast<int64_t>(4 * st + l4) * m];
    // Row tiles in PAIRS with interleaved rows: MFMA column c of tile u (u = 0, 1) stands for row ip + 2 c + u of
    // Binv, so a lane's two accumulator entries for one column of Binv are neighbours in memory and every access
    // of Binv and of E is a 16-byte one (32 consecutive rows = 256 bytes per group of 16 lanes).
    const bool even = (m & 1) == 0; // 16-byte alignment of every column needs an even m
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        const int64_t ip = i0 + 32 * t;
        if (ip >= m) break;
        const int64_t r0 = ip + 2 * l15; // this lane's two rows: r0, r0 + 1
        const bool in0 = r0 < m, in1 = r0 + 1 < m;
        const bool fast = even && in1; // (an odd m breaks the 16-byte alignment of the columns; the last lane may hold one row)
        spx_v2d eb[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const double *e = P.E + static_cast<int64_t>(4 * st + l4) * m;
            if (fast) eb[st] = *reinterpret_cast<const spx_v2d *>(e + r0);
            else eb[st] = spx_v2d{in0 ? e[r0] : 0.0, in1 ? e[r0 + 1] : 0.0};
        }
        spx_v4d acc0, acc1;
        double *cp[4];
        bool okc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t kk = k0 + l4 + 4 * q;
            okc[q] = kk < m;
            cp[q] = P.Binv + (okc[q] ? kk : 0) * m;
            spx_v2d c2;
            if (fast) c2 = *reinterpret_cast<const spx_v2d *>(cp[q] + r0);
            else c2 = spx_v2d{in0 ? cp[q][r0] : 0.0, in1 ? cp[q][r0 + 1] : 0.0};
            acc0[q] = -c2.x; // accumulate  -Binv + R E^T  and flip the sign at the store
            acc1[q] = -c2.y;
        }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[st], eb[st].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[st], eb[st].y, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!okc[q]) continue;
            if (fast) {
                *reinterpret_cast<spx_v2d *>(cp[q] + r0) = spx_v2d{-acc0[q], -acc1[q]};
            } else {
                if (in0) cp[q][r0] = -acc0[q];
                if (in1) cp[q][r0 + 1] = -acc1[q];
            }
        }
    }
}

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
        if (P.w) P.w[k] = 1.0;
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

// phase 1: relax every basic variable that violates a bound -- above its upper bound U it gets [U, inf) and
// cost +1, below its lower bound L it gets (-inf, L] and cost -1 -- so that the phase-1 objective is the sum
// of the bound violations; a relaxed variable that reaches the violated bound leaves the basis there and
// gets its true bounds back (spx_commit).  Works from any basis, not only the all-logical one.
__global__ __launch_bounds__(SX_WG) void k_spx_phase1_setup(Spx P, double ftol) {
    long long cnt = 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < P.m;
         i += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int k = P.head[i];
        const double v = P.x[k], lo = P.lo[k], up = P.up[k];
        if (v > up + ftol) {
            P.lo[k] = up;
            P.up[k] = INFINITY;
            P.cost[k] = 1.0;
            P.relaxed[k] = 1;
            ++cnt;
        } else if (v < lo - ftol) {
            P.up[k] = lo;
            P.lo[k] = -INFINITY;
            P.cost[k] = -1.0;
            P.relaxed[k] = 1;
            ++cnt;
        }
    }
    cnt = sx_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long *>(&P.st->n_relaxed),
                                                  static_cast<unsigned long long>(cnt));
}

// infeasibility of the current basic solution: max bound violation over basic variables, and the
// phase-1 objective (sum over relaxed logicals of |x|)
__global__ __launch_bounds__(1024) void k_spx_measure(Spx P, double *out /* [0]=max violation, [1]=phase-1 objective, [2]=objective */) {
    double viol = 0.0, p1 = 0.0;
    for (int i = threadIdx.x; i < P.m; i += 1024) {
        const int k = P.head[i];
        const double v = P.x[k];
        if (P.relaxed[k]) { // distance to the bound it violates (the finite end of its relaxed range)
            p1 += (P.lo[k] > -INFINITY) ? fmax(v - P.lo[k], 0.0) : fmax(P.up[k] - v, 0.0);
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
        if (P.w) P.w[k] = 1.0; // new objective: new reference framework
        if (k < P.n) {
            P.cost[k] = P.c_true[k];
            if (P.relaxed[k]) {
                P.relaxed[k] = 0;
                P.lo[k] = P.l_true[k];
                P.up[k] = P.u_true[k];
            }
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
    st->dvx_on = 0;
    st->n_relaxed = 0;
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

// true residual of the current point, all variables: r_i = b_i - sum_j a_ij x_j - s_i; out[0] = max |r_i|,
// out[1] = max |b_i| (both as the bit pattern of a non-negative double, which orders like an integer)
__global__ __launch_bounds__(SX_WG) void k_spx_residual(Spx P, const int64_t *__restrict__ tiles, int64_t ntiles,
                                                        const int64_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ colidx,
                                                        const double *__restrict__ val,
                                                        unsigned long long *__restrict__ out) {
    __shared__ sx_walk_lds<1, SPX_CHUNK> lds;
    double worst = 0.0, bmax = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double acc[1];
        int64_t i;
        bool valid;
        sx_segwalk<1, SPX_CHUNK>(tiles, t, rowptr, colidx, val, StageDot{P.x}, lds, i, valid, acc);
        if (valid) {
            const double r = fabs(P.b[i] - acc[0] - P.x[P.n + i]);
            worst = (r > worst || r != r) ? (r != r ? INFINITY : r) : worst; // NaN counts as the worst residual
            bmax = fmax(bmax, fabs(P.b[i]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        worst = fmax(worst, __shfl_down(worst, o, 64));
        bmax = fmax(bmax, __shfl_down(bmax, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], static_cast<unsigned long long>(__double_as_longlong(worst)));
        atomicMax(&out[1], static_cast<unsigned long long>(__double_as_longlong(bmax)));
    }
}

// back to the all-logical basis without moving anything: every basic structural becomes superbasic at its
// current value (it is about to be pivoted in again; if no row takes it, it stays where it is), every logical
// basic, its current value remembered in s_keep for the moment a structural takes its row
__global__ __launch_bounds__(SX_WG) void k_spx_logical_basis(Spx P) {
    const int64_t N = P.n + P.m;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < N;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        if (k < P.n) {
            if (P.status[k] == ST_BASIC) P.status[k] = ST_FREE;
        } else {
            P.s_keep[k - P.n] = P.x[k];
            P.status[k] = ST_BASIC;
            P.head[k - P.n] = static_cast<int32_t>(k);
        }
    }
}

// starting point: structurals the warm basis wants basic (code 0) or superbasic (-3) take the point's value,
// clipped to their bounds, as superbasic variables -- pivoting them in will not move them -- and the slack
// b - A x of every row is remembered for the logical that loses its row
__global__ __launch_bounds__(SX_WG) void k_spx_apply_start(Spx P, const int8_t *__restrict__ vb,
                                                           const double *__restrict__ x_start) {
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; k < P.n;
         k += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const int code = vb[k];
        if (code == ST_BASIC || code == ST_FREE) {
            const double v = x_start[k];
            P.status[k] = ST_FREE;
            P.x[k] = (v == v) ? fmin(fmax(v, P.lo[k]), P.up[k]) : ((P.lo[k] > -INFINITY) ? P.lo[k] : 0.0);
        }
    }
}
__global__ __launch_bounds__(SX_WG) void k_spx_start_slacks(Spx P, const int64_t *__restrict__ tiles, int64_t ntiles,
                                                            const int64_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ colidx,
                                                            const double *__restrict__ val) {
    __shared__ sx_walk_lds<1, SPX_CHUNK> lds;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        double acc[1];
        int64_t i;
        bool valid;
        sx_segwalk<1, SPX_CHUNK>(tiles, t, rowptr, colidx, val, StageDot{P.x}, lds, i, valid, acc);
        if (valid) P.s_keep[i] = P.b[i] - acc[0];
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
            if (q) (void)sx_dfree(q);
    }
    int reserve(size_t bytes) {
        void *d = nullptr;
        SX_HIP(sx_dmalloc(&d, bytes));
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
        SX_HIP(sx_dmalloc(&d, want));
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
    if (session->Binv) { // parked in the context for the next session of the same size (one spare at most)
        sx_ctx *ctx = session->ctx;
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->spare_binv) (void)sx_dfree(ctx->spare_binv);
        ctx->spare_binv = session->Binv;
        ctx->spare_binv_m = session->m;
    }
    delete session;
    return SX_OK;
}

static int spx_solve(sx_ctx *ctx, sx_simplex_session *session, const sx_matrix *A, const double *b, const double *c,
                     const double *l, const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                     const int8_t *cbasis_in, const int64_t *col_ids, const double *x_start, int64_t max_iter,
                     double feas_tol, double opt_tol, double *x_out, double *y_out, int8_t *vbasis_out,
                     int8_t *cbasis_out, sx_simplex_result *result);

SX_API int sx_simplex_solve_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                                const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol,
                                double *x_out, double *y_out, int8_t *vbasis_out, int8_t *cbasis_out,
                                sx_simplex_result *result) {
    return spx_solve(ctx, nullptr, A, b, c, l, u, row_is_lt, vbasis_in, cbasis_in, nullptr, nullptr, max_iter, feas_tol,
                     opt_tol, x_out, y_out, vbasis_out, cbasis_out, result);
}

SX_API int sx_simplex_solve_session_dev(sx_ctx *ctx, sx_simplex_session *session, const sx_matrix *A, const double *b,
                                        const double *c, const double *l, const double *u, const uint8_t *row_is_lt,
                                        const int8_t *vbasis_in, const int8_t *cbasis_in, const int64_t *col_ids,
                                        int64_t max_iter, double feas_tol, double opt_tol, double *x_out,
                                        double *y_out, int8_t *vbasis_out, int8_t *cbasis_out,
                                        sx_simplex_result *result) {
    return spx_solve(ctx, session, A, b, c, l, u, row_is_lt, vbasis_in, cbasis_in, col_ids, nullptr, max_iter, feas_tol,
                     opt_tol, x_out, y_out, vbasis_out, cbasis_out, result);
}

SX_API int sx_simplex_crossover_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                    const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                                    const int8_t *cbasis_in, const double *x_start, int64_t max_iter, double feas_tol,
                                    double opt_tol, double *x_out, double *y_out, int8_t *vbasis_out,
                                    int8_t *cbasis_out, sx_simplex_result *result) {
    SX_REQUIRE(x_start != nullptr && vbasis_in != nullptr && cbasis_in != nullptr,
               "sx_simplex_crossover_dev needs a starting point and a basis guess");
    return spx_solve(ctx, nullptr, A, b, c, l, u, row_is_lt, vbasis_in, cbasis_in, nullptr, x_start, max_iter, feas_tol,
                     opt_tol, x_out, y_out, vbasis_out, cbasis_out, result);
}

static int spx_solve(sx_ctx *ctx, sx_simplex_session *session, const sx_matrix *A, const double *b, const double *c,
                     const double *l, const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                     const int8_t *cbasis_in, const int64_t *col_ids, const double *x_start, int64_t max_iter,
                     double feas_tol, double opt_tol, double *x_out, double *y_out, int8_t *vbasis_out,
                     int8_t *cbasis_out, sx_simplex_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(session == nullptr || session->ctx == ctx, "the session belongs to another context");
    SX_REQUIRE(A && b && c && l && u && row_is_lt && result, "NULL argument");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the simplex needs both layouts of A");
    SX_REQUIRE((vbasis_in == nullptr) == (cbasis_in == nullptr), "vbasis_in and cbasis_in go together");
    memset(result, 0, sizeof(*result));
    const int64_t m = A->m, n = A->n, N = n + m;
    {   // the basis inverse is an explicit dense m x m matrix: it has to fit the HBM that is free now
        size_t free_b = 0, total_b = 0;
        SX_HIP(hipMemGetInfo(&free_b, &total_b));
        const double need = 8.0 * static_cast<double>(m) * static_cast<double>(m) + 600.0 * static_cast<double>(m) +
                            40.0 * static_cast<double>(N);
        const bool have_inverse = session && session->m == m && session->Binv != nullptr;
        if (!have_inverse && need > 0.9 * static_cast<double>(free_b)) {
            sx_set_error("sx_simplex_solve: the dense basis inverse of %lld rows needs %.1f GB, %.1f GB of HBM are free",
                         (long long)m, need / 1e9, static_cast<double>(free_b) / 1e9);
            return SX_ERR_UNSUPPORTED;
        }
    }
    if (max_iter <= 0) max_iter = 50 * (m + n) + 1000;
    hipStream_t s = ctx->stream;
    const auto t_enter = std::chrono::steady_clock::now();
    // measured (profiles/r01/spx_bench.txt): holding the updates back is as fast as the rank-one update per
    // pivot at 200-1000 rows and 1.3x / 2.2x faster at 2000 / 4000, so "auto" means on
    const bool defer = ctx->opt_spx_defer != 0;
    const bool devex = ctx->opt_spx_pricing != 0;

    DevBufs mem;
    {
        // everything below except a session-less inverse: 34 B per variable, 44 B per row, pricing partials
        const size_t um = static_cast<size_t>(m), uN = static_cast<size_t>(N);
        size_t bytes = 34 * uN + 56 * um + 24 * (static_cast<size_t>(SPX_GRID) + um / 64 + 64) + 32 * 256 + 4096;
        if (!session) bytes += sizeof(double) * um * um + 256;
        if (defer) bytes += 2 * sizeof(double) * um * SPX_DEFER + 1024;
        bytes += 32 * (um / SX_WG + 2) + 1024; // ratio candidates, tickets
        if (devex) bytes += sizeof(double) * uN + 256;
        SX_TRY(mem.reserve(bytes));
    }
    Spx P;
    P.m = m;
    P.n = n;
    P.row_lt = row_is_lt;
    P.c_true = c;
    P.l_true = l;
    P.u_true = u;
    SX_TRY(mem.get(static_cast<size_t>(N), &P.status));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.relaxed));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.lo));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.up));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.cost));
    SX_TRY(mem.get(static_cast<size_t>(N), &P.x));
    P.w = nullptr;
    if (devex) SX_TRY(mem.get(static_cast<size_t>(N), &P.w));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.head));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.y));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.d));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.rho));
    SX_TRY(mem.get(static_cast<size_t>(m), &P.rhs));
    double *s_keep_buf = nullptr;
    SX_TRY(mem.get(static_cast<size_t>(m), &s_keep_buf));
    P.s_keep = nullptr;
    if (session) { // the inverse outlives the call
        if (session->m != m || session->Binv == nullptr) {
            if (session->Binv) SX_HIP(sx_dfree(session->Binv));
            session->Binv = nullptr;
            session->valid = false;
            session->m = m;
            if (ctx->spare_binv && ctx->spare_binv_m == m) {
                session->Binv = ctx->spare_binv;
                ctx->spare_binv = nullptr;
                ctx->spare_binv_m = 0;
            } else {
                SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&session->Binv),
                                 sizeof(double) * (static_cast<size_t>(m) * static_cast<size_t>(m) + 1)));
            }
        }
        P.Binv = session->Binv;
    } else {
        SX_TRY(mem.get(static_cast<size_t>(m) * static_cast<size_t>(m), &P.Binv));
    }
    P.E = P.R = nullptr;
    P.er = nullptr;
    if (defer) {
        SX_TRY(mem.get(static_cast<size_t>(m) * SPX_DEFER, &P.E));
        SX_TRY(mem.get(static_cast<size_t>(m) * SPX_DEFER, &P.R));
        SX_TRY(mem.get(static_cast<size_t>(SPX_DEFER), &P.er));
        // unused slots take part in the fold with a zero factor: they must hold finite numbers
        SX_HIP(hipMemsetAsync(P.E, 0, sizeof(double) * static_cast<size_t>(m) * SPX_DEFER, s));
        SX_HIP(hipMemsetAsync(P.R, 0, sizeof(double) * static_cast<size_t>(m) * SPX_DEFER, s));
        SX_HIP(hipMemsetAsync(P.er, 0xff, sizeof(int32_t) * SPX_DEFER, s));
    }
    {
        const size_t ratio_slots = static_cast<size_t>(grid1d(m));
        SX_TRY(mem.get(3 * ratio_slots, &P.rt_val));
        SX_TRY(mem.get(2 * ratio_slots, &P.rt_idx));
        SX_TRY(mem.get(2, &P.tickets));
        SX_HIP(hipMemsetAsync(P.tickets, 0, 2 * sizeof(unsigned), s));
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
    const int fold_col_tiles = static_cast<int>((m + FOLD_COLS - 1) / FOLD_COLS);
    const unsigned fold_grid = static_cast<unsigned>(((m + SX_WG - 1) / SX_WG) * fold_col_tiles);
    // large inverses fold on the matrix cores (k_spx_fold_mfma), small ones by the scalar kernel in pivot order
    static const char *blas_env = getenv("SX_SPX_BLAS"); // "0": the scalar fold kernel at every size (A/B runs)
    // measured at m = 2e4 (tools/fold_bench.py, profiles/r03/spx_fold.md): scalar kernel 1.76 ms per fold, this
    // matrix-core kernel 2.2 ms, the vendor DGEMM it replaced 1.46 ms -- so the scalar kernel is the default and the
    // matrix-core one is kept for the record (and its tests) behind SX_SPX_FOLD_MFMA_MIN = rows from which it runs
    const char *fm_env = getenv("SX_SPX_FOLD_MFMA_MIN");
    const int64_t fm_min = fm_env ? atoll(fm_env) : INT64_MAX;
    const bool fold_mfma = defer && m >= fm_min && !(blas_env && blas_env[0] == '0');
    const int64_t fm_nkt = (m + 15) / 16, fm_nic = (m + 255) / 256;
    const unsigned fm_grid = static_cast<unsigned>(((fm_nkt + FM_KT - 1) / FM_KT) * ((fm_nic + FM_IC - 1) / FM_IC) * FM_KT * FM_IC);
    auto fold = [&](int cnt) {
        if (fold_mfma) {
            hipLaunchKernelGGL(k_spx_fold_mask, dim3(gM, SPX_DEFER), dim3(SX_WG), 0, s, P, cnt); // empty slots: zero columns
            hipLaunchKernelGGL(k_spx_fold_mfma, dim3(fm_grid), dim3(SX_WG), 0, s, P, fm_nkt, fm_nic);
            return;
        }
        hipLaunchKernelGGL(k_spx_fold, dim3(fold_grid), dim3(SX_WG), 0, s, P, cnt, fold_col_tiles);
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

    // ---- helpers of the starting basis and of numerical hygiene
    unsigned long long *resid_dev = nullptr;
    SX_TRY(mem.get(2, &resid_dev));
    int8_t *vb_now = nullptr, *cb_now = nullptr; // the current basis in the reference's codes (re-inversion)
    SX_TRY(mem.get(static_cast<size_t>(n), &vb_now));
    SX_TRY(mem.get(static_cast<size_t>(m), &cb_now));
    // max_i |b_i - (A x)_i - s_i| / (1 + max |b|): what the explicit inverse has drifted by
    auto residual = [&](double *rel) -> int {
        SX_HIP(hipMemsetAsync(resid_dev, 0, 2 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_spx_residual, dim3(gR), dim3(SX_WG), 0, s, P, A->csr_tiles, A->n_csr_tiles, A->csr_ptr,
                           A->csr_idx, A->csr_val, resid_dev);
        unsigned long long h[2] = {0, 0};
        SX_HIP(hipMemcpyAsync(h, resid_dev, sizeof(h), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        double worst, bmax;
        memcpy(&worst, &h[0], sizeof(double));
        memcpy(&bmax, &h[1], sizeof(double));
        *rel = worst / (1.0 + bmax);
        return SX_OK;
    };
    const double resid_tol = feas_tol;
    // pivot the structural columns flagged basic in vb_host (n codes) into the all-logical basis (identity
    // inverse): for each column the largest available pivot among the rows still held by a logical, rows whose
    // logical the target basis keeps (cb_dev code 0) only as a last resort; a column without a usable row stays
    // non-basic
    auto install = [&](const std::vector<int8_t> &vb_host, const int8_t *cb_dev) {
        int slot = 0;
        for (int64_t j = 0; j < n; ++j) {
            if (vb_host[static_cast<size_t>(j)] != ST_BASIC) continue;
            hipLaunchKernelGGL(k_spx_ftran, dim3(gM), dim3(SX_WG), 0, s, P, A->csc_ptr, A->csc_idx, A->csc_val,
                               static_cast<int>(j), 0, defer ? slot : 0, cb_dev);
            hipLaunchKernelGGL(k_spx_rho, dim3(gM), dim3(SX_WG), 0, s, P, static_cast<int>(j), defer ? slot : 0,
                               defer ? slot : -1);
            if (!defer) {
                hipLaunchKernelGGL(k_spx_update_binv, dim3(gMM), dim3(SX_WG), 0, s, P);
            } else if (++slot == SPX_DEFER) {
                fold(slot);
                slot = 0;
            }
        }
        if (defer && slot > 0) fold(slot);
        hipLaunchKernelGGL(k_spx_reset_state, dim3(1), dim3(1), 0, s, P);
    };
    // rebuild the inverse of the CURRENT basis from its columns (the running inverse has only ever seen
    // rank-one updates and drifts): all logicals back in, identity, the basic structurals pivoted in again
    int64_t n_reinvert = 0;
    std::vector<int8_t> vb_host;
    auto reinvert = [&]() -> int {
        hipLaunchKernelGGL(k_spx_export, dim3(gN), dim3(SX_WG), 0, s, P, static_cast<double *>(nullptr),
                           static_cast<double *>(nullptr), vb_now, cb_now);
        vb_host.resize(static_cast<size_t>(n));
        SX_HIP(hipMemcpyAsync(vb_host.data(), vb_now, static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        P.s_keep = s_keep_buf; // nothing moves: displaced logicals keep the value they have now
        hipLaunchKernelGGL(k_spx_logical_basis, dim3(gN), dim3(SX_WG), 0, s, P);
        hipLaunchKernelGGL(k_spx_identity, dim3(gMM), dim3(SX_WG), 0, s, P);
        if (defer) SX_HIP(hipMemsetAsync(P.er, 0xff, sizeof(int32_t) * SPX_DEFER, s));
        install(vb_host, cb_now);
        P.s_keep = nullptr;
        ++n_reinvert;
        SX_HIP(hipGetLastError());
        return SX_OK;
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
            // the kept inverse is accepted when it still reproduces the constraints: x_B = Binv (b - N x_N) must
            // satisfy A x + s = b (a basis that is merely infeasible for the new bounds goes through phase 1)
            double rel = 0.0;
            SX_TRY(residual(&rel)); // synchronises: head_new may go out of scope afterwards
            reused = warm = rel <= resid_tol;
        }
    }
    if (session) session->valid = false; // until this solve has left a basis behind
    const bool trace = getenv("SX_SPX_TRACE") != nullptr; // stderr: where the wall time of a solve goes
    auto since_enter = [&]() {
        (void)hipStreamSynchronize(s);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    };
    const double t_setup_ms = trace ? since_enter() : 0.0;
    if (!reused) cold_start(1);
    double t_crash_ms = 0.0;
    if (vbasis_in && !reused) {
        hipLaunchKernelGGL(k_spx_apply_vbasis, dim3(grid1d(n)), dim3(SX_WG), 0, s, P, vbasis_in);
        if (x_start) {
            // crossover start: every variable keeps the value of the point -- columns the basis guess wants
            // basic become superbasic at x_start and are then pivoted in where a row is free, the slack of a
            // row a structural takes stays at b - A x_start -- so the first basic solution IS the point
            P.s_keep = s_keep_buf;
            hipLaunchKernelGGL(k_spx_apply_start, dim3(grid1d(n)), dim3(SX_WG), 0, s, P, vbasis_in, x_start);
            hipLaunchKernelGGL(k_spx_start_slacks, dim3(gR), dim3(SX_WG), 0, s, P, A->csr_tiles, A->n_csr_tiles,
                               A->csr_ptr, A->csr_idx, A->csr_val);
        }
        install(vb, cbasis_in);
        P.s_keep = nullptr;
        if (trace) {
            const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
            fprintf(stderr, "[sx_simplex] warm basis: launches enqueued by %.2f ms\n", enq);
            t_crash_ms = since_enter();
        }
        warm = true; // whatever could be pivoted in stays; infeasibilities go through phase 1
    }
    SX_HIP(hipGetLastError());
    const double t_start_ms = trace ? since_enter() : 0.0;

    // ---- phases
    SpxState host;
    memset(&host, 0, sizeof(host));
    const int batch = SPX_DEFER;
    // pivot number k of a batch; with deferred updates it sees k pending pairs and records its own in slot k
    auto enqueue_pivot = [&](int k) {
        if (devex)
            hipLaunchKernelGGL(k_spx_price<true>, dim3(gP + gL), dim3(SX_WG), 0, s, P, A->csc_tiles, A->n_csc_tiles,
                               A->csc_ptr, A->csc_idx, A->csc_val, opt_tol, gP);
        else
            hipLaunchKernelGGL(k_spx_price<false>, dim3(gP + gL), dim3(SX_WG), 0, s, P, A->csc_tiles, A->n_csc_tiles,
                               A->csc_ptr, A->csc_idx, A->csc_val, opt_tol, gP);
        hipLaunchKernelGGL(k_spx_ftran, dim3(gM), dim3(SX_WG), 0, s, P, A->csc_ptr, A->csc_idx, A->csc_val, -1, gP + gL,
                           defer ? k : 0, static_cast<const int8_t *>(nullptr));
        hipLaunchKernelGGL(k_spx_rho_update, dim3(gM), dim3(SX_WG), 0, s, P, defer ? k : 0, defer ? k : -1);
        if (!defer) hipLaunchKernelGGL(k_spx_update_binv, dim3(gMM), dim3(SX_WG), 0, s, P);
    };
    auto enqueue_batch = [&](bool with_fold = true) {
        for (int k = 0; k < batch; ++k) enqueue_pivot(k);
        if (defer && with_fold) fold(batch);
    };
    // a batch of 64 pivots = 192 small launches (+ the fold) with fixed arguments: captured into a hipGraph and
    // replayed (pivots are launch-bound for small m); direct launches are the fallback.  Capturing and
    // instantiating costs milliseconds, more than a short warm-started re-solve takes altogether, so the
    // graph is only built once a solve has gone through GRAPH_AFTER pivots by direct launches.
    const int64_t GRAPH_AFTER = 4096; // with 3 launches per pivot replay only insures against a slow host
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool graph_tried = false, graph_has_fold = true;
    auto build_graph = [&]() {
        graph_tried = true;
        if (!ctx->opt_graph) return;
        // rocprofv3 --kernel-trace (ROCm 7.2) segfaults inside hipGraphLaunch after some thousands of
        // replays of this graph; with three launches per pivot direct launches are as fast
        // (profiles/r01/spx_bench.txt), so a profiled run simply does not replay
        if (getenv("ROCP_TOOL_LIBRARIES") != nullptr) return;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            graph_has_fold = true;
            enqueue_batch(graph_has_fold);
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

    // x_B and y from scratch cost two passes over the m x m inverse, a pivot O(m): the interval grows with m
    // so that the refresh stays a small share of the pivots between two of them (64 up to m = 2048, 2048 from
    // m = 11.6k on); the residual check every "spx_check" pivots is the safety net either way
    int64_t refresh_every = 64;
    {
        const double scale = static_cast<double>(m) / 2048.0;
        int64_t want = static_cast<int64_t>(64.0 * scale * scale);
        want = (want / 64) * 64;
        refresh_every = want < 64 ? 64 : (want > 2048 ? 2048 : want);
    }
    int64_t direct_pivots = 0; // over both phases
    auto run_phase = [&](int64_t budget) -> int {
        int64_t done_iters = 0;
        while (true) {
            if (!exec && !graph_tried && direct_pivots >= GRAPH_AFTER) build_graph();
            if (!exec) direct_pivots += batch;
            if (exec) {
                SX_HIP(hipGraphLaunch(exec, s));
                if (defer && !graph_has_fold) fold(batch);
            } else {
                enqueue_batch();
            }
            done_iters += batch;
            SX_HIP(hipGetLastError());
            SX_HIP(hipMemcpyAsync(&host, P.st, sizeof(host), hipMemcpyDeviceToHost, s));
            SX_HIP(hipStreamSynchronize(s));
            if (host.done || host.iters >= budget) return SX_OK;
            if (done_iters % refresh_every == 0) refresh(true); // numerical hygiene: x_B and y from scratch
            if (done_iters % ctx->opt_spx_check == 0) { // and the inverse itself, against the constraints
                double rel = 0.0;
                SX_TRY(residual(&rel));
                if (!(rel <= resid_tol) || ctx->opt_spx_force_reinvert) {
                    SX_TRY(reinvert());
                    refresh(true);
                }
            }
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
        hipLaunchKernelGGL(k_spx_reset_state, dim3(1), dim3(1), 0, s, P); // also zeroes the relaxed count
        hipLaunchKernelGGL(k_spx_phase1_setup, dim3(gM), dim3(SX_WG), 0, s, P, feas_tol);
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
        // phase 2, and the check the advisor asked for: "optimal" is only reported for a point that satisfies
        // A x + s = b; if the running inverse has drifted, it is rebuilt from the basis columns and the phase
        // resumed from the same basis (twice at most)
        for (int attempt = 0; attempt < 3 && status < 0; ++attempt) {
            SX_TRY(run_phase(max_iter));
            if (host.done == 2) status = 2;
            else if (host.done == 3) status = 4;
            else if (host.done == 0) status = 3;
            else {
                refresh(true);
                double rel = 0.0;
                SX_TRY(residual(&rel));
                if (rel <= resid_tol) status = 0;
                else if (attempt == 2) status = 4;
                else {
                    SX_TRY(reinvert());
                    refresh(true);
                }
            }
        }
    }
    refresh(true);
    SX_TRY(measure());
    hipLaunchKernelGGL(k_spx_export, dim3(gN), dim3(SX_WG), 0, s, P, x_out, y_out, vbasis_out, cbasis_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipStreamSynchronize(s));
    if (trace) {
        const double t_all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
        fprintf(stderr,
                "[sx_simplex] m=%lld n=%lld start=%s: set-up %.2f ms, basis %.2f ms (columns installed by %.2f), "
                "phases %.2f ms, %lld pivots (%lld in phase 1), %lld re-inversions, status %d\n",
                (long long)m, (long long)n, reused ? "kept inverse" : (vbasis_in ? (warm ? "crash" : "crash dropped") : "cold"),
                t_setup_ms, t_start_ms - t_setup_ms, t_crash_ms > 0 ? t_crash_ms - t_setup_ms : 0.0, t_all - t_start_ms,
                (long long)host.iters, (long long)phase1_iters, (long long)n_reinvert, status);
    }
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
