// Sparse crossover (kernel group K16s): from the point the first-order stage (sx_pdlp.hip) leaves -- or from a given
// basis -- to an optimal vertex and its basis WITHOUT a dense m x m inverse.  Stands where the reference's backends run
// their crossover behind the barrier (lp_methods/algorithms.py:50-54 -> solver_caller/gurobi.py:111-115) and their
// warm-started simplex (lp_methods/algorithms.py:69-74), all inside Gurobi.
//
// Rounds 1-2 kept B^-1 explicitly (8 m^2 bytes); round 3 factored a band basis once, covered the dense (linking) rows by
// their logicals and paid for it with one pivot per linking row on a 66-99 GB tableau.  Now:
//   * the basis is factored in BORDERED form (sx_border.h): rows in their natural (or Cuthill-McKee) order, dense rows set
//     aside; every basic variable is matched to a band row by entry size -> B11, a band matrix (K16f); what finds no band
//     row -- the linking activities, logicals of dense rows -- forms the border together with the dense rows, and the
//     Schur complement of the border is factored by the dense LU (K16g).  EVERY basic variable is in the factors, so a
//     fresh factorisation of the current basis (full eta file, numerical trouble, final check) is a true
//     refactorisation, and a given basis (vbasis_in / cbasis_in) is taken as it is.  Columns the two LUs find
//     dependent leave the basis (superbasic), the logical of the row in question takes their place;
//   * the simplex works on an explicit TABLEAU of the few columns that can still move -- the superbasic ones plus what
//     pricing adds: T = B^-1 A_J (positions x |J|, column major in HBM, grown on demand).  A pivot is a ratio test down
//     one column, a copy of one row, and a rank-one update of T deferred in product form (64 per fold); every entering
//     column is also kept as an eta vector, so duals (B^-T c_B = B0^-T E_1^T .. E_k^T c_B) and new tableau columns
//     (E_k .. E_1 B0^-1 a_j) need no second factorisation until the eta file is full;
//   * when no tracked column prices out, all other columns are priced with those duals (the K1 walk) and the
//     violators join the tableau -- column generation, as in the reference's network crossover;
//   * before a vertex is called optimal its row residuals (one K2 walk) and the reduced costs of its basic columns are
//     checked against A itself; beyond the tolerances the basis is factored afresh and the run goes on.
// Memory O(nnz + m (kl + ku) + nb^2 + m |J|).  Everything that decides a pivot runs on the device; the host replays
// batches of pivots and reads one status word per batch.
#include "sx_internal.h"
#include "sx_border.h"

#include <algorithm>
#include <cstring>
#include <cmath>
#include <numeric>
#include <thread>

namespace {

constexpr int TB_WG = 256;
constexpr int TB_SUP = 3, TB_LOW = 1, TB_UPP = 2; // status of a tracked non-basic column
constexpr double TB_PIV = 1e-7;                   // smallest |alpha| the ratio test accepts
constexpr double TB_TINY = 1e-60; // band solves of tableau columns treat windows below this as zeros
constexpr double TB_DROP = 1e-14;                 // tableau entries below this become exact zeros (keeps B^-1 A_J local)

struct TbState {
    long long iters, max_iter, n_eta, cap_eta, pivots, flips, degen;
    int status;  // 0 running, 1 no tracked column prices out, 2 unbounded, 3 iteration limit, 4 numerical trouble
    int phase;   // 1: some basic variable is outside its bounds
    int q, dir;  // entering slot and its direction
    int r;       // leaving position (-1: bound flip of the entering column)
    int hit;     // bound the leaving variable stops at: 1 lower, 2 upper
    int n_inf;
    int n_bl;       // blocks of 256 positions in which the entering column has an entry (their list: blist)
    int n_pend;     // basis changes whose rank-one updates are not applied to T yet (product form: TbPend)
    int folding;    // the batch's fold is under way
    double theta, alpha_r, dq, sum_inf, feas_tol, opt_tol, tmax;
};

// Pending updates of a batch (product form).  Pivot s of the batch has entering slot q_s, leaving position r_s,
// u_s = alpha_s - e_{r_s} (alpha_s is the eta vector of the pivot) and v_s = row r_s of the tableau at that time
// divided by the pivot element (v_s[q_s] = 1 / pivot).  The current tableau column of slot j is
//     base_j - sum_{s >= s0[j]} u_s v_s[j],   base_j = T[:, j]  or  e_{sbase[j]} once j has been an entering slot,
// and row r is the same expression read across -- k vectors of m resp. |J| entries per pivot instead of a pass over
// the whole m x |J| tableau; k_tb_fold applies a batch in ONE pass (at 1e6 rows x 8,600 columns a rank-one update
// is 130 GB of traffic, 16 ms; the batch of 48-64 costs the same once).
constexpr int TB_K = 64;
struct TbPend {
    const double *vbuf; // [TB_K][ldv]
    const int32_t *pr;  // [TB_K] leaving position of the pending pivots
    const int32_t *s0;  // [slots] first pending pivot that applies to the slot
    const int32_t *sbase; // [slots] -1: base column is T's; >= 0: base column is that unit vector
    int64_t ldv;
};
__device__ __forceinline__ double tb_current(const double *__restrict__ T, const double *__restrict__ eta, const TbState *st,
                                              const TbPend &P, int64_t m, int64_t p, int j) {
    const int sb = P.sbase[j];
    double a = sb < 0 ? T[static_cast<size_t>(j) * m + p] : (p == sb ? 1.0 : 0.0);
    const int np = st->n_pend;
    const long long e0 = st->n_eta - np;
    for (int s = P.s0[j]; s < np; ++s) {
        const double u = eta[static_cast<size_t>(e0 + s) * m + p] - (p == P.pr[s] ? 1.0 : 0.0);
        a -= u * P.vbuf[static_cast<size_t>(s) * P.ldv + j];
    }
    return a;
}

struct TbPart {
    double t, a;
    int p, hit;
};

// ---------------------------------------------------------------------------------------------- set-up kernels
__global__ __launch_bounds__(TB_WG) void k_tb_scatter_cols(int64_t nslots, const int32_t *__restrict__ var, int64_t n,
                                                           const int64_t *__restrict__ cptr, const int32_t *__restrict__ cidx,
                                                           const double *__restrict__ cval, const int32_t *__restrict__ eqidx,
                                                           double *__restrict__ T, int64_t m) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    if (s >= nslots) return;
    double *col = T + static_cast<size_t>(s) * m;
    const int64_t v = var[s];
    if (v >= n) {
        col[eqidx[v - n]] = 1.0;
        return;
    }
    for (int64_t k = cptr[v]; k < cptr[v + 1]; ++k) col[eqidx[cidx[k]]] = cval[k];
}

__device__ __forceinline__ double tb_block_sum(double v, double *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    __syncthreads();
    return r;
}

// The probe's short cut: out[0] = the tallest column of A over the rows that are not dense (largest - smallest row index of
// a column's entries in such rows), out[1] = the number of dense rows.  Whatever columns a basis takes and whatever rows they
// are matched to, an entry lies at most that far from its column's position in the natural order of the band rows (setting
// rows aside only shortens distances): kl, ku <= out[0].
__global__ __launch_bounds__(TB_WG) void k_tb_colspan(int64_t m, int64_t n, const int64_t *__restrict__ cptr, const int32_t *__restrict__ cidx,
                                                      const int64_t *__restrict__ rptr, int64_t dense_thr, int *__restrict__ out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    int span = 0, dense = 0;
    if (t < n) {
        int lo = INT32_MAX, hi = -1;
        for (int64_t k = cptr[t]; k < cptr[t + 1]; ++k) {
            const int i = cidx[k];
            if (rptr[i + 1] - rptr[i] > dense_thr) continue;
            lo = i < lo ? i : lo;
            hi = i > hi ? i : hi;
        }
        if (hi >= 0) span = hi - lo;
    }
    if (t < m && rptr[t + 1] - rptr[t] > dense_thr) dense = 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(span, o, 64);
        span = other > span ? other : span;
        dense += __shfl_xor(dense, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (span > 0) atomicMax(&out[0], span);
        if (dense > 0) atomicAdd(&out[1], dense);
    }
}

// dst[:, dest[s]] = src[:, s]  (rows x gridDim.y columns)
__global__ __launch_bounds__(TB_WG) void k_tb_copy_cols(int64_t rows, const double *__restrict__ src, int64_t lds_, const int32_t *__restrict__ dest,
                                                        double *__restrict__ dst, int64_t ldd) {
    const int64_t sc = blockIdx.y;
    const double *a = src + static_cast<size_t>(sc) * lds_;
    double *o = dst + static_cast<size_t>(dest[sc]) * ldd;
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x; r < rows; r += static_cast<int64_t>(gridDim.x) * TB_WG) o[r] = a[r];
}

// entries below the drop tolerance -> exact zeros (what the pivots skip)
// (grid-stride: a launch carries fewer than 2^32 work-items, a tableau has more entries than that)
__global__ __launch_bounds__(TB_WG) void k_tb_drop(int64_t total, double *__restrict__ W, double tol) {
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x; t < total; t += static_cast<int64_t>(gridDim.x) * TB_WG)
        if (fabs(W[t]) < tol) W[t] = 0.0;
}

// out[s] = base[s] - sum_p w[p] T[p, s]        (one workgroup per tracked column)
__global__ __launch_bounds__(TB_WG) void k_tb_coldot(int64_t m, const double *__restrict__ T, const double *__restrict__ w,
                                                      const double *__restrict__ base, double *__restrict__ out) {
    __shared__ double sm[4];
    const int64_t s = blockIdx.x;
    const double *col = T + static_cast<size_t>(s) * m;
    double acc = 0.0;
    for (int64_t p = threadIdx.x; p < m; p += TB_WG) {
        const double t = col[p];
        if (t != 0.0) acc += w[p] * t;
    }
    const double tot = tb_block_sum(acc, sm);
    if (threadIdx.x == 0) out[s] = (base ? base[s] : 0.0) - tot;
}

// ---------------------------------------------------------------------------------------------- one pivot
// infeasibility signs of the basic variables; per-workgroup counts
__global__ __launch_bounds__(TB_WG) void k_tb_infeas(int64_t m, const double *__restrict__ xB, const double *__restrict__ lB,
                                                     const double *__restrict__ uB, const TbState *__restrict__ st,
                                                     double *__restrict__ g, double *__restrict__ part) {
    __shared__ double sm[4];
    if (st->status != 0) return;
    const int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    double gi = 0.0, inf = 0.0;
    if (p < m) {
        const double x = xB[p], tol = st->feas_tol;
        if (x > uB[p] + tol) {
            gi = 1.0;
            inf = x - uB[p];
        } else if (x < lB[p] - tol) {
            gi = -1.0;
            inf = lB[p] - x;
        }
        g[p] = gi;
    }
    const double cnt = tb_block_sum(gi != 0.0 ? 1.0 : 0.0, sm);
    const double sum = tb_block_sum(inf, sm);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = cnt;
        part[2 * blockIdx.x + 1] = sum;
    }
}

__global__ __launch_bounds__(TB_WG) void k_tb_phase(int nblk, const double *__restrict__ part, TbState *st, int32_t *__restrict__ infoff) {
    __shared__ double sm[4];
    __shared__ int sc[TB_WG];
    if (st->status != 0) return;
    // thread t owns the blocks [t c, (t + 1) c): their counts, then an exclusive scan -> infoff[block] = infeasible
    // positions before it (k_tb_inflist writes the list of positions in ascending order from these)
    const int chunk = (nblk + TB_WG - 1) / TB_WG;
    const int k0 = threadIdx.x * chunk, k1 = (k0 + chunk < nblk) ? k0 + chunk : nblk;
    double cnt = 0.0, sum = 0.0;
    for (int k = k0; k < k1; ++k) { // (counts are small integers: any order gives the same sum)
        cnt += part[2 * k];
        sum += part[2 * k + 1];
    }
    const int mine = static_cast<int>(cnt);
    sc[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < TB_WG; o <<= 1) {
        const int v = (threadIdx.x >= o) ? sc[threadIdx.x - o] : 0;
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sc[threadIdx.x] - mine;
    for (int k = k0; k < k1; ++k) {
        infoff[k] = run;
        run += static_cast<int>(part[2 * k]);
    }
    cnt = tb_block_sum(cnt, sm);
    sum = tb_block_sum(sum, sm);
    if (threadIdx.x == 0) {
        st->n_inf = static_cast<int>(cnt);
        st->sum_inf = sum;
        st->phase = cnt > 0.0 ? 1 : 2;
    }
}

// phase 1: the infeasible positions in ascending order (at most TB_INFLIST of them; more: the pricing kernels read
// the columns whole).  A workgroup per block of 256 positions that holds one: ordered compaction by ballots.
constexpr int TB_INFLIST = 4096;
__global__ __launch_bounds__(TB_WG) void k_tb_inflist(int64_t m, const double *__restrict__ g, const double *__restrict__ part,
                                                      const int32_t *__restrict__ infoff, const TbState *__restrict__ st,
                                                      int32_t *__restrict__ list) {
    __shared__ int wsum[TB_WG / 64];
    if (st->status != 0 || st->phase != 1 || st->n_inf > TB_INFLIST) return;
    if (part[2 * blockIdx.x] <= 0.0) return;
    const int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    const bool hit = p < m && g[p] != 0.0;
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (hit) list[infoff[blockIdx.x] + before + __popcll(bal & ((1ull << lane) - 1ull))] = static_cast<int32_t>(p);
}

// phase 1 only: d1[j] = - g^T (current column j) = -( g^T base_j - sum_{s >= s0[j]} (g^T u_s) v_s[j] ).  Few basic
// variables are infeasible behind a first-order point, so g^T T[:, j] gathers the listed positions only (k_tb_inflist:
// ~200 of 1e6 at config-5 size -- scanning the 256-position blocks that hold one cost 1.3 ms per iteration there);
// with more than TB_INFLIST of them it reads the column whole, coalesced.
__global__ __launch_bounds__(TB_WG) void k_tb_gu(int64_t m, int nblk, const double *__restrict__ eta, const double *__restrict__ g,
                                                 const int32_t *__restrict__ list, const TbState *__restrict__ st, TbPend P,
                                                 double *__restrict__ gu) {
    __shared__ double sm[4];
    if (st->status != 0 || st->phase != 1) return;
    const int sidx = blockIdx.x;
    if (sidx >= st->n_pend) return;
    const double *al = eta + static_cast<size_t>(st->n_eta - st->n_pend + sidx) * m;
    const int r = P.pr[sidx];
    double acc = 0.0;
    if (st->n_inf <= TB_INFLIST) {
        for (int i = threadIdx.x; i < st->n_inf; i += TB_WG) {
            const int p = list[i];
            acc += g[p] * (al[p] - (p == r ? 1.0 : 0.0));
        }
    } else {
        for (int64_t p = threadIdx.x; p < m; p += TB_WG) {
            const double gp = g[p];
            if (gp != 0.0) acc += gp * (al[p] - (p == r ? 1.0 : 0.0));
        }
    }
    const double tot = tb_block_sum(acc, sm);
    if (threadIdx.x == 0) gu[sidx] = tot;
}

__global__ __launch_bounds__(TB_WG) void k_tb_price1(int64_t m, int nblk, const double *__restrict__ T, const double *__restrict__ g,
                                                     const int32_t *__restrict__ list, const TbState *__restrict__ st, TbPend P,
                                                     const double *__restrict__ gu, double *__restrict__ d1) {
    __shared__ double sm[4];
    if (st->status != 0 || st->phase != 1) return;
    const int64_t s = blockIdx.x;
    const int sb = P.sbase[s];
    double acc = 0.0;
    if (sb >= 0) {
        if (threadIdx.x == 0) acc = g[sb];
    } else {
        const double *col = T + static_cast<size_t>(s) * m;
        if (st->n_inf <= TB_INFLIST) {
            for (int i = threadIdx.x; i < st->n_inf; i += TB_WG) {
                const int p = list[i];
                acc += g[p] * col[p];
            }
        } else {
            for (int64_t p = threadIdx.x; p < m; p += TB_WG) {
                const double gp = g[p];
                if (gp != 0.0) acc += gp * col[p];
            }
        }
    }
    double tot = tb_block_sum(acc, sm);
    if (threadIdx.x == 0) {
        for (int k = P.s0[s]; k < st->n_pend; ++k) tot -= gu[k] * P.vbuf[static_cast<size_t>(k) * P.ldv + s];
        d1[s] = -tot;
    }
}

// entering column: superbasic columns first (they have to leave their interior value), then the largest
// reduced cost of the right sign; one workgroup
__global__ __launch_bounds__(TB_WG) void k_tb_select(int64_t nJ, const double *__restrict__ dJ, const double *__restrict__ d1,
                                                     const int32_t *__restrict__ statJ, const double *__restrict__ xJ,
                                                     const double *__restrict__ lJ, const double *__restrict__ uJ, TbState *st) {
    __shared__ double bv[TB_WG];
    __shared__ int bi[TB_WG], bd[TB_WG];
    if (st->status != 0) return;
    const bool ph1 = st->phase == 1;
    const double tol = st->opt_tol;
    double best = -1.0;
    int bs = -1, bdir = 0;
    for (int64_t s = threadIdx.x; s < nJ; s += TB_WG) {
        const int stt = statJ[s];
        if (stt == 0) continue; // empty slot
        const double d = ph1 ? d1[s] : dJ[s];
        double score = -1.0;
        int dir = 0;
        const bool fixed = lJ[s] == uJ[s];
        if (fixed && stt != TB_SUP) continue;
        if (stt == TB_SUP) {
            if (d < -tol) dir = 1;
            else if (d > tol) dir = -1;
            if (dir != 0) score = fabs(d) + 1e30; // a superbasic column that prices out: first of all
            else if (!ph1) {
                // prices at zero: still has to reach a bound (vertex); towards the nearer one
                const double dl = xJ[s] - lJ[s], du = uJ[s] - xJ[s];
                if (!(isinf(dl) && isinf(du))) {
                    dir = (dl <= du) ? -1 : 1;
                    score = 1e29;
                }
            }
        } else if (stt == TB_LOW) {
            if (d < -tol) {
                dir = 1;
                score = -d;
            }
        } else if (stt == TB_UPP) {
            if (d > tol) {
                dir = -1;
                score = d;
            }
        }
        if (score > best) {
            best = score;
            bs = static_cast<int>(s);
            bdir = dir;
        }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = bs;
    bd[threadIdx.x] = bdir;
    __syncthreads();
    for (int o = TB_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const double ob = bv[threadIdx.x + o];
            const int oi = bi[threadIdx.x + o];
            if (ob > bv[threadIdx.x] || (ob == bv[threadIdx.x] && oi >= 0 && (bi[threadIdx.x] < 0 || oi < bi[threadIdx.x]))) {
                bv[threadIdx.x] = ob;
                bi[threadIdx.x] = oi;
                bd[threadIdx.x] = bd[threadIdx.x + o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (bi[0] < 0) {
            st->status = 1; // nothing tracked prices out (in phase 1: with respect to the infeasibility)
            st->q = -1;
        } else {
            st->q = bi[0];
            st->dir = bd[0];
            st->dq = dJ[bi[0]];
            if (st->iters >= st->max_iter) st->status = 3;
            else if (st->n_eta >= st->cap_eta) st->status = 3;
        }
    }
}

// Ratio test down the entering column (Harris, two passes): pass 1 finds the longest step that keeps every basic
// variable within feas_tol of its bounds; pass 2 takes, among the rows that block no later than that, the one with
// the largest pivot.  The column is kept as the eta vector of this pivot.
__device__ __forceinline__ void tb_row_limit(double a, double dir, double x, double lo, double up, double tol, double &dist, double &rate,
                                             int &hit) {
    // distance of x_B[p] to the bound it moves towards (inf: none) and |rate| of the approach
    rate = fabs(a);
    hit = 0;
    dist = INFINITY;
    const double r = -dir * a;
    if (r < 0.0) {
        if (x > up + tol) { // starts above its upper bound: becomes feasible there
            dist = x - up;
            hit = 2;
        } else if (!isinf(lo) && x >= lo - tol) {
            dist = fmax(x - lo, 0.0);
            hit = 1;
        }
    } else {
        if (x < lo - tol) {
            dist = lo - x;
            hit = 1;
        } else if (!isinf(up) && x <= up + tol) {
            dist = fmax(up - x, 0.0);
            hit = 2;
        }
    }
}

__global__ __launch_bounds__(TB_WG) void k_tb_ratio1(int64_t m, const double *__restrict__ T, const double *__restrict__ xB,
                                                     const double *__restrict__ lB, const double *__restrict__ uB,
                                                     const TbState *__restrict__ st, double *__restrict__ eta,
                                                     double *__restrict__ part, TbPend P) {
    __shared__ double sm[TB_WG];
    __shared__ int any_nz;
    if (st->status != 0) return;
    if (threadIdx.x == 0) any_nz = 0;
    __syncthreads();
    const int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    const double dir = static_cast<double>(st->dir), tol = st->feas_tol;
    double t = INFINITY;
    if (p < m) {
        double a = tb_current(T, eta, st, P, m, p, st->q);
        if (fabs(a) < TB_DROP) a = 0.0;
        eta[static_cast<size_t>(st->n_eta) * m + p] = a;
        if (a != 0.0) any_nz = 1; // (benign race: every writer stores 1)
        if (fabs(a) > TB_PIV) {
            double dist, rate;
            int hit;
            tb_row_limit(a, dir, xB[p], lB[p], uB[p], tol, dist, rate, hit);
            if (hit) t = (dist + tol) / rate;
        }
    }
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int o = TB_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = sm[0];
        part[gridDim.x + blockIdx.x] = any_nz ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(TB_WG) void k_tb_tmax(int nblk, const double *__restrict__ part, TbState *st, int32_t *__restrict__ blist) {
    __shared__ double sm[TB_WG];
    __shared__ int cnt[TB_WG];
    if (st->status != 0) return;
    // every lane a contiguous run of blocks: the list comes out in ascending order whatever the run lengths
    const int per = (nblk + TB_WG - 1) / TB_WG;
    const int k0 = threadIdx.x * per, k1 = (k0 + per < nblk) ? k0 + per : nblk;
    double t = INFINITY;
    int mine = 0;
    for (int k = k0; k < k1; ++k) {
        t = fmin(t, part[k]);
        if (part[nblk + k] != 0.0) ++mine;
    }
    sm[threadIdx.x] = t;
    cnt[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tm = INFINITY;
        int run = 0;
        for (int w = 0; w < TB_WG; ++w) {
            tm = fmin(tm, sm[w]);
            const int c = cnt[w];
            cnt[w] = run;
            run += c;
        }
        st->tmax = tm;
        st->n_bl = run;
    }
    __syncthreads();
    int o = cnt[threadIdx.x];
    for (int k = k0; k < k1; ++k)
        if (part[nblk + k] != 0.0) blist[o++] = k;
}

__global__ __launch_bounds__(TB_WG) void k_tb_ratio(int64_t m, const double *__restrict__ xB, const double *__restrict__ lB,
                                                    const double *__restrict__ uB, const TbState *__restrict__ st,
                                                    const double *__restrict__ eta, TbPart *__restrict__ part) {
    __shared__ double st_[TB_WG], sa_[TB_WG];
    __shared__ int sp_[TB_WG], sh_[TB_WG];
    if (st->status != 0) return;
    const int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    const double dir = static_cast<double>(st->dir), tol = st->feas_tol, tmax = st->tmax;
    double t = INFINITY, aa = -1.0;
    int hit = 0;
    if (p < m) {
        const double a = eta[static_cast<size_t>(st->n_eta) * m + p];
        if (fabs(a) > TB_PIV) {
            double dist, rate;
            int h;
            tb_row_limit(a, dir, xB[p], lB[p], uB[p], tol, dist, rate, h);
            if (h && dist / rate <= tmax) {
                t = dist / rate;
                aa = rate;
                hit = h;
            }
        }
    }
    st_[threadIdx.x] = t;
    sa_[threadIdx.x] = aa;
    sp_[threadIdx.x] = static_cast<int>(p);
    sh_[threadIdx.x] = hit;
    __syncthreads();
    for (int o = TB_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            if (sa_[threadIdx.x + o] > sa_[threadIdx.x]) { // larger pivot; ties to the smaller position (kept)
                st_[threadIdx.x] = st_[threadIdx.x + o];
                sa_[threadIdx.x] = sa_[threadIdx.x + o];
                sp_[threadIdx.x] = sp_[threadIdx.x + o];
                sh_[threadIdx.x] = sh_[threadIdx.x + o];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        TbPart r;
        r.t = st_[0];
        r.a = sa_[0];
        r.p = sp_[0];
        r.hit = sh_[0];
        part[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(TB_WG) void k_tb_decide(int nblk, const TbPart *__restrict__ part, const double *__restrict__ xJ,
                                                     const double *__restrict__ lJ, const double *__restrict__ uJ,
                                                     const int32_t *__restrict__ statJ, TbState *st) {
    __shared__ double sa_[TB_WG], st_[TB_WG];
    __shared__ int sp_[TB_WG], sh_[TB_WG];
    if (st->status != 0) return;
    // largest pivot among the rows pass 2 admitted; ties to the smaller position (blocks are in position order)
    TbPart b;
    b.a = -1.0;
    b.t = INFINITY;
    b.p = 0x7fffffff;
    b.hit = 0;
    for (int k = threadIdx.x; k < nblk; k += TB_WG) {
        const TbPart o = part[k];
        if (o.a > b.a || (o.a == b.a && o.p < b.p)) b = o;
    }
    sa_[threadIdx.x] = b.a;
    st_[threadIdx.x] = b.t;
    sp_[threadIdx.x] = b.p;
    sh_[threadIdx.x] = b.hit;
    __syncthreads();
    for (int o = TB_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const int j = threadIdx.x + o;
            if (sa_[j] > sa_[threadIdx.x] || (sa_[j] == sa_[threadIdx.x] && sp_[j] < sp_[threadIdx.x])) {
                sa_[threadIdx.x] = sa_[j];
                st_[threadIdx.x] = st_[j];
                sp_[threadIdx.x] = sp_[j];
                sh_[threadIdx.x] = sh_[j];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    b.a = sa_[0];
    b.t = st_[0];
    b.p = sp_[0];
    b.hit = sh_[0];
    if (b.a < 0.0) b.t = INFINITY; // no row blocks
    const int q = st->q;
    // the entering column's own way to its other bound
    double own = INFINITY;
    if (st->dir > 0) own = uJ[q] - xJ[q];
    else own = xJ[q] - lJ[q];
    if (own < 0.0) own = 0.0;
    if (isinf(b.t) && isinf(own)) {
        st->status = st->phase == 1 ? 4 : 2;
        return;
    }
    if (own <= b.t) {
        st->r = -1;
        st->theta = own;
        st->hit = st->dir > 0 ? 2 : 1;
    } else {
        st->r = b.p;
        st->theta = b.t;
        st->hit = b.hit;
    }
}

// row r of the current tableau -> rowbuf; alpha_r
__global__ __launch_bounds__(TB_WG) void k_tb_rowcopy(int64_t m, int64_t nJ, const double *__restrict__ T, const double *__restrict__ eta,
                                                      TbState *st, TbPend P, double *__restrict__ rowbuf) {
    if (st->status != 0 || st->r < 0) return;
    const int64_t s = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    if (s >= nJ) return;
    double v = tb_current(T, eta, st, P, m, st->r, static_cast<int>(s));
    if (fabs(v) < TB_DROP) v = 0.0;
    if (s == st->q) {
        v = eta[static_cast<size_t>(st->n_eta) * m + st->r]; // (the very number the ratio test saw)
        st->alpha_r = v;
    }
    rowbuf[s] = v;
}

// x_B moves along the entering column (only the blocks of positions in which that column has an entry); the
// tableau itself is not touched: the pivot joins the pending batch (k_tb_post) and k_tb_fold applies the batch
__global__ __launch_bounds__(TB_WG) void k_tb_update(int64_t m, double *__restrict__ xB, const double *__restrict__ eta,
                                                     const TbState *__restrict__ st, const int32_t *__restrict__ blist) {
    if (st->status != 0) return;
    const int r = st->r;
    for (int e = blockIdx.x; e < st->n_bl; e += gridDim.x) {
        const int64_t p = static_cast<int64_t>(blist[e]) * TB_WG + threadIdx.x;
        if (p >= m || p == r) continue;
        const double a = eta[static_cast<size_t>(st->n_eta) * m + p];
        if (a != 0.0) xB[p] = xB[p] - static_cast<double>(st->dir) * st->theta * a;
    }
}

// One pass over the tableau for a whole batch: T[p, j] = base - sum_{s >= s0[j]} u_s[p] v_s[j].  A lane owns a
// position (its u_s in registers, 64 at most), a workgroup TB_FJ slots (their v_s in LDS).
__global__ void k_tb_fold_begin(TbState *st, int slack) {
    st->folding = (st->n_pend > 0 && (st->status != 0 || st->n_pend + slack > TB_K)) ? 1 : 0;
}
constexpr int TB_FJ = 64; // slots per workgroup of the fold: the batch's u_s[p] are read once per TB_FJ columns
__global__ __launch_bounds__(TB_WG) void k_tb_fold(int64_t m, int64_t nJ, double *__restrict__ T, const double *__restrict__ eta,
                                                   const TbState *__restrict__ st, TbPend P) {
    __shared__ double vs[TB_K][TB_FJ + 1];
    __shared__ int ss0[TB_FJ], ssb[TB_FJ], snz[TB_FJ];
    if (!st->folding) return;
    const int np = st->n_pend;
    const long long e0 = st->n_eta - np;
    const int64_t j0 = static_cast<int64_t>(blockIdx.y) * TB_FJ;
    const int nj = static_cast<int>((nJ - j0 < TB_FJ) ? nJ - j0 : TB_FJ);
    if (threadIdx.x < TB_FJ) {
        ss0[threadIdx.x] = threadIdx.x < nj ? P.s0[j0 + threadIdx.x] : np;
        ssb[threadIdx.x] = threadIdx.x < nj ? P.sbase[j0 + threadIdx.x] : -1;
        snz[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TB_K * TB_FJ; e += TB_WG) {
        const int sidx = e / TB_FJ, jj = e % TB_FJ;
        // zero where the pivot does not apply to the slot (before its s0, beyond the batch): the inner loop below is 64
        // multiply-adds without a test -- with `if (sidx >= first && sidx < np)` around each, the uniform branches cost five
        // times the arithmetic (105 ms per pass at config-5 size)
        const double v = (sidx < np && jj < nj && sidx >= ss0[jj]) ? P.vbuf[static_cast<size_t>(sidx) * P.ldv + j0 + jj] : 0.0;
        vs[sidx][jj] = v;
        if (v != 0.0) snz[jj] = 1; // (every writer stores 1)
    }
    __syncthreads();
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x; p < m; p += static_cast<int64_t>(gridDim.x) * TB_WG) {
        double u[TB_K];
        bool any = false;
#pragma unroll
        for (int sidx = 0; sidx < TB_K; ++sidx) {
            double x = 0.0;
            if (sidx < np) x = eta[static_cast<size_t>(e0 + sidx) * m + p] - (p == P.pr[sidx] ? 1.0 : 0.0);
            u[sidx] = x;
            any = any || x != 0.0;
        }
        for (int jj = 0; jj < nj; ++jj) {
            const int sb = ssb[jj];
            // the row saw none of the batch's pivots, or the column none of its pivot rows: the slot keeps its entry
            if ((!any || !snz[jj]) && sb < 0) continue;
            double *t = T + static_cast<size_t>(j0 + jj) * m + p;
            double acc = sb < 0 ? *t : (p == sb ? 1.0 : 0.0);
#pragma unroll
            for (int sidx = 0; sidx < TB_K; ++sidx) acc = __builtin_fma(-u[sidx], vs[sidx][jj], acc); // (u is zero beyond the batch)
            *t = fabs(acc) < TB_DROP ? 0.0 : acc;
        }
    }
}
__global__ __launch_bounds__(TB_WG) void k_tb_fold_end(int64_t nJ, TbState *st, int32_t *__restrict__ s0, int32_t *__restrict__ sbase) {
    if (!st->folding) return;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    if (j < nJ) {
        s0[j] = 0;
        sbase[j] = -1;
    }
}
__global__ void k_tb_fold_done(TbState *st) {
    if (st->folding) {
        st->n_pend = 0;
        st->folding = 0;
    }
}

// reduced costs of the tracked columns and the bookkeeping of the pivot; one workgroup
__global__ __launch_bounds__(TB_WG) void k_tb_post(int64_t nJ, double *__restrict__ dJ, const double *__restrict__ rowbuf,
                                                   int32_t *__restrict__ head, double *__restrict__ xB, double *__restrict__ lB,
                                                   double *__restrict__ uB, double *__restrict__ cB, int32_t *__restrict__ varJ,
                                                   double *__restrict__ xJ, double *__restrict__ lJ, double *__restrict__ uJ,
                                                   double *__restrict__ cJ, int32_t *__restrict__ statJ, int32_t *__restrict__ eta_r,
                                                   TbState *st, double *__restrict__ vbuf, int64_t ldv, int32_t *__restrict__ pr,
                                                   int32_t *__restrict__ s0, int32_t *__restrict__ sbase) {
    if (st->status != 0) return;
    const int q = st->q, r = st->r;
    const double dq = st->dq, ar = st->alpha_r;
    if (r >= 0) {
        const double f = dq / ar;
        const bool good = fabs(ar) > TB_PIV;
        double *v = vbuf + static_cast<size_t>(st->n_pend) * ldv;
        for (int64_t s = threadIdx.x; s < nJ; s += TB_WG) {
            const double rb = rowbuf[s];
            if (s != q) {
                if (rb != 0.0) dJ[s] = dJ[s] - f * rb;
                if (good) v[s] = rb / ar;
            } else if (good) {
                v[s] = 1.0 / ar;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double xq = xJ[q] + static_cast<double>(st->dir) * st->theta;
    st->iters += 1;
    if (!(st->theta > 1e-12)) st->degen += 1;
    if (r < 0) { // the entering column went to its other bound
        xJ[q] = st->hit == 2 ? uJ[q] : lJ[q];
        statJ[q] = st->hit == 2 ? TB_UPP : TB_LOW;
        st->flips += 1;
        return;
    }
    if (!(fabs(ar) > TB_PIV)) {
        st->status = 4;
        return;
    }
    // leaving variable -> slot q at the bound it hit; entering variable -> position r
    const int32_t vout = head[r];
    const double lo = lB[r], up = uB[r], co = cB[r];
    head[r] = varJ[q];
    xB[r] = xq;
    lB[r] = lJ[q];
    uB[r] = uJ[q];
    cB[r] = cJ[q];
    varJ[q] = vout;
    lJ[q] = lo;
    uJ[q] = up;
    cJ[q] = co;
    xJ[q] = st->hit == 2 ? up : lo;
    statJ[q] = st->hit == 2 ? TB_UPP : TB_LOW;
    dJ[q] = -dq / ar;
    eta_r[st->n_eta] = r;
    pr[st->n_pend] = r;
    s0[q] = st->n_pend; // the slot now holds the leaving variable: its column is e_r - u v[q] from this pivot on
    sbase[q] = r;
    st->n_pend += 1;
    st->n_eta += 1;
    st->pivots += 1;
}

// ---------------------------------------------------------------------------------------------- eta file
// W <- E_{k0+K-1} ... E_{k0} W for ncols columns (ldw = m), K <= TB_EB etas at a time.  One eta: w_r <- w_r / alpha_r,
// w_p <- w_p - alpha_p w_r (p != r), i.e. w <- w - u sigma with u = alpha - e_r and sigma = w_r / alpha_r.  K of them:
// sigma_i = (w0[r_i] - sum_{j<i} u_j[r_i] sigma_j) / alpha_i[r_i] (a K x K triangle per column), then W <- W - U Sigma in
// ONE pass over W instead of K (applying 8,214 etas to 373 columns one eta at a time moved 9 GB per eta: 16.5 s).
constexpr int TB_EB = 64; // etas per block
constexpr int TB_EC = 64; // columns per workgroup of the rank-K pass (16: the eta block was re-read once per 16 columns -- 12 GB of a 18 GB pass at 1e6 rows x 373 columns)
__global__ __launch_bounds__(TB_WG) void k_tb_etab_gather(int64_t m, int64_t ncols, const double *__restrict__ W, const double *__restrict__ eta,
                                                          const int32_t *__restrict__ eta_r, int64_t k0, int K, double *__restrict__ G,
                                                          double *__restrict__ M) {
    // G[i][c] = W[r_i, c];  M[i][j] = u_j[r_i] (j < i), M[i][i] = alpha_i[r_i]
    const int64_t t = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x;
    if (t < static_cast<int64_t>(TB_EB) * ncols) {
        const int i = static_cast<int>(t / ncols);
        const int64_t c = t - static_cast<int64_t>(i) * ncols;
        G[t] = (i < K) ? W[static_cast<size_t>(c) * m + eta_r[k0 + i]] : 0.0;
    }
    if (t < TB_EB * TB_EB) {
        const int i = static_cast<int>(t) / TB_EB, j = static_cast<int>(t) % TB_EB;
        double v = (i == j) ? 1.0 : 0.0;
        if (i < K && j <= i) {
            const int ri = eta_r[k0 + i];
            v = eta[static_cast<size_t>(k0 + j) * m + ri];
            if (j < i && eta_r[k0 + j] == ri) v -= 1.0;
        }
        M[t] = v;
    }
}
__global__ __launch_bounds__(64) void k_tb_etab_solve(int64_t ncols, const double *__restrict__ G, const double *__restrict__ M,
                                                      double *__restrict__ S, int32_t *__restrict__ colflag) {
    __shared__ double sM[TB_EB * TB_EB];
    for (int e = threadIdx.x; e < TB_EB * TB_EB; e += 64) sM[e] = M[e];
    __syncthreads();
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
    if (c >= ncols) return;
    double sg[TB_EB];
    int any = 0;
#pragma unroll
    for (int i = 0; i < TB_EB; ++i) {
        double x = G[static_cast<size_t>(i) * ncols + c];
#pragma unroll
        for (int j = 0; j < i; ++j) x -= sM[i * TB_EB + j] * sg[j];
        sg[i] = x / sM[i * TB_EB + i];
        any |= sg[i] != 0.0;
    }
#pragma unroll
    for (int i = 0; i < TB_EB; ++i) S[static_cast<size_t>(i) * ncols + c] = sg[i];
    colflag[c] = any;
}
__global__ __launch_bounds__(TB_WG) void k_tb_etab_apply(int64_t m, int64_t ncols, double *__restrict__ W, const double *__restrict__ eta,
                                                         const int32_t *__restrict__ eta_r, int64_t k0, int K, const double *__restrict__ S,
                                                         const int32_t *__restrict__ colflag) {
    __shared__ double sS[TB_EB * TB_EC];
    __shared__ int sflag[TB_EC], sr[TB_EB], sany;
    const int64_t c0 = static_cast<int64_t>(blockIdx.y) * TB_EC;
    const int nc = static_cast<int>((ncols - c0 < TB_EC) ? ncols - c0 : TB_EC);
    if (threadIdx.x == 0) sany = 0;
    __syncthreads();
    if (threadIdx.x < TB_EC) {
        const int fl = (threadIdx.x < nc) ? colflag[c0 + threadIdx.x] : 0;
        sflag[threadIdx.x] = fl;
        if (fl) sany = 1;
    }
    if (threadIdx.x < TB_EB) sr[threadIdx.x] = (threadIdx.x < K) ? eta_r[k0 + threadIdx.x] : -1;
    for (int e = threadIdx.x; e < TB_EB * TB_EC; e += TB_WG) {
        const int i = e / TB_EC, cc = e % TB_EC;
        sS[e] = (cc < nc) ? S[static_cast<size_t>(i) * ncols + c0 + cc] : 0.0;
    }
    __syncthreads();
    if (!sany) return; // none of the tile's columns meets a pivot row of the block
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x; p < m; p += static_cast<int64_t>(gridDim.x) * TB_WG) {
        double u[TB_EB];
        int nz = 0;
#pragma unroll
        for (int i = 0; i < TB_EB; ++i) {
            double a = (i < K) ? eta[static_cast<size_t>(k0 + i) * m + p] : 0.0;
            if (sr[i] == p) a -= 1.0;
            u[i] = a;
            nz |= a != 0.0;
        }
        if (!nz) continue;
        for (int cc = 0; cc < nc; ++cc) {
            if (!sflag[cc]) continue;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < TB_EB; ++i) acc += u[i] * sS[i * TB_EC + cc];
            double *w = W + static_cast<size_t>(c0 + cc) * m + p;
            *w = *w - acc;
        }
    }
}
// v <- E_{k0}^T ... E_{k0+K-1}^T v (the duals' way through the eta file: the youngest eta first), K <= TB_EB etas per pass.
// One eta: v[r] <- (v[r] - sum_{p != r} alpha[p] v[p]) / alpha[r]: a dot over all positions that changes ONE entry, so the K dots
// of a block are taken against the vector the block found (k_tb_etat_dots: one pass over the K eta columns, all workgroups busy)
// and corrected by what the younger etas of the block changed -- a K x K triangle in one wave (k_tb_etat_solve).  One eta at a
// time this was 2 launches and a 64-workgroup dot per eta: 80 ms per pricing round with 921 etas over 1e6 positions.
constexpr int TB_ETS = 32; // slices of the positions per eta column (partial dots, summed in order)
__global__ __launch_bounds__(TB_WG) void k_tb_etat_dots(int64_t m, const double *__restrict__ v, const double *__restrict__ eta,
                                                        const int32_t *__restrict__ eta_r, int64_t k0, double *__restrict__ part) {
    __shared__ double sm[4];
    const int64_t k = k0 + blockIdx.y;
    const int r = eta_r[k];
    const double *alpha = eta + static_cast<size_t>(k) * m;
    double acc = 0.0;
    for (int64_t p = static_cast<int64_t>(blockIdx.x) * TB_WG + threadIdx.x; p < m; p += static_cast<int64_t>(gridDim.x) * TB_WG)
        if (p != r) {
            const double a = alpha[p];
            if (a != 0.0) acc += a * v[p];
        }
    const double tot = tb_block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.y * TB_ETS + blockIdx.x] = tot;
}
__global__ __launch_bounds__(64) void k_tb_etat_solve(int64_t m, int nslice, int K, double *__restrict__ v, const double *__restrict__ eta,
                                                      const int32_t *__restrict__ eta_r, int64_t k0, const double *__restrict__ part) {
    __shared__ double sM[TB_EB * TB_EB]; // sM[i][j] = alpha_i[r_j] (j > i, r_j != r_i): what step j's change adds to step i's dot
    __shared__ int sr[TB_EB];
    const int j = threadIdx.x;
    sr[j] = (j < K) ? eta_r[k0 + j] : -1;
    __syncthreads();
    for (int e = j; e < TB_EB * TB_EB; e += 64) {
        const int i = e / TB_EB, jj = e % TB_EB;
        double x = 0.0;
        if (i < K && jj < K && jj > i && sr[jj] != sr[i]) x = eta[static_cast<size_t>(k0 + i) * m + sr[jj]];
        sM[e] = x;
    }
    double dot = 0.0, cur = 0.0, diag = 1.0, delta = 0.0;
    if (j < K) {
        for (int g = 0; g < nslice; ++g) dot += part[j * TB_ETS + g];
        cur = v[sr[j]];
        diag = eta[static_cast<size_t>(k0 + j) * m + sr[j]];
    }
    __syncthreads();
    for (int i = K - 1; i >= 0; --i) { // lane jj holds delta of step jj (0 until it ran) and the current value at r_jj
        double t = (j > i && j < K) ? sM[i * TB_EB + j] * delta : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        const double s_i = __shfl(dot, i, 64) + t;
        const double cur_i = __shfl(cur, i, 64);
        const double nw = (cur_i - s_i) / __shfl(diag, i, 64);
        if (j == i) delta = nw - cur_i;
        if (j <= i && sr[j] == sr[i]) cur = nw; // (older etas of the block on the same position see the new value)
    }
    // the value a position ends with is that of the OLDEST eta of the block on it
    if (j < K) {
        bool last = true;
        for (int q = 0; q < j; ++q) last = last && (sr[q] != sr[j]);
        if (last) v[sr[j]] = cur;
    }
}

struct DevBufs {
    std::vector<void *> p;
    ~DevBufs() {
        for (void *q : p) (void)sx_dfree(q);
    }
    template <class T>
    int get(size_t count, T **out) {
        void *d = nullptr;
        if (sx_dmalloc(&d, sizeof(T) * (count ? count : 1)) != hipSuccess) {
            sx_set_error("hipMalloc of %zu bytes failed in the sparse crossover", sizeof(T) * count);
            return SX_ERR_NOMEM;
        }
        p.push_back(d);
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

template <class T>
int up(hipStream_t s, T *dst, const std::vector<T> &src) {
    if (!src.empty()) SX_HIP(hipMemcpyAsync(dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, s));
    return SX_OK;
}
template <class T>
int down(hipStream_t s, std::vector<T> &dst, const T *src, size_t count) {
    dst.resize(count);
    if (count) SX_HIP(hipMemcpyAsync(dst.data(), src, sizeof(T) * count, hipMemcpyDeviceToHost, s));
    return SX_OK;
}

inline unsigned gridof(int64_t n) { return static_cast<unsigned>(n > 0 ? (n + TB_WG - 1) / TB_WG : 1); }
inline unsigned gridcap(int64_t n) { return static_cast<unsigned>(std::min<int64_t>(gridof(n), 1 << 20)); } // for grid-stride kernels


// The two big blocks of an epoch -- eta file and tableau, tens of GB at 1e6 rows -- out of ONE block that a helper thread
// allocates while the host matches columns to rows: hipMalloc of a block that size takes as long as the matching itself
// (0.5-1.4 s for the 28 GB of config-5 size, box by box; sx_internal.h: the context keeps the block between calls and a
// backend can ask for it before its first-order stage)
struct BigArena {
    sx_ctx *ctx = nullptr;
    void *base = nullptr;
    size_t bytes = 0, used = 0;
    bool started = false;
    // the context's block when it is there (a backend that asked for it ahead of the call, or the last call's), else a
    // prefetch of our own, taken when it is first needed
    void start(sx_ctx *c, size_t want) {
        ctx = c;
        started = true;
        if (sx_ctx_take_block(ctx, want - want / 3, &base, &bytes)) return; // (a block a third short of the estimate will do)
        bytes = want;
        (void)sx_ctx_prefetch_block(ctx, want);
    }
    void join() {
        if (started && !base) {
            size_t got = 0;
            if (sx_ctx_take_block(ctx, bytes - bytes / 3, &base, &got)) bytes = got;
            else bytes = 0;
            started = false;
        }
    }
    void *take(size_t n) { // nullptr: no room (the caller allocates for itself)
        join();
        n = (n + 255) & ~static_cast<size_t>(255);
        if (!base || used + n > bytes) return nullptr;
        void *q = static_cast<char *>(base) + used;
        used += n;
        return q;
    }
    void reset() { used = 0; }
    ~BigArena() {
        join();
        if (ctx && base) sx_ctx_give_block(ctx, base, bytes); // (kept for the next call of the process)
    }
};

// per-slot arrays of the tableau (one slot per tracked column) and the tableau itself; grown when pricing brings in more
// columns than there is room for (the explicit tableau of round 3 was sized once, at 90 % of the free memory)
struct TbSlots {
    int64_t mp = 0, cap = 0;
    bool T_own = true; // false: T is a piece of the call's arena
    double *T = nullptr, *xJ = nullptr, *lJ = nullptr, *uJ = nullptr, *cJ = nullptr, *dJ = nullptr, *d1 = nullptr, *rowbuf = nullptr,
           *vbuf = nullptr, *ebG = nullptr, *ebS = nullptr;
    int32_t *varJ = nullptr, *statJ = nullptr, *s0 = nullptr, *sbase = nullptr, *elist = nullptr;
    TbSlots() = default;
    TbSlots(const TbSlots &) = delete;
    TbSlots &operator=(const TbSlots &) = delete;
    ~TbSlots() {
        for (void *q : {(void *)(T_own ? T : nullptr), (void *)xJ, (void *)lJ, (void *)uJ, (void *)cJ, (void *)dJ, (void *)d1, (void *)rowbuf, (void *)vbuf, (void *)ebG,
                        (void *)ebS, (void *)varJ, (void *)statJ, (void *)s0, (void *)sbase, (void *)elist})
            (void)sx_dfree(q);
    }
    void swap_with(TbSlots &o) {
        std::swap(mp, o.mp); std::swap(cap, o.cap); std::swap(T_own, o.T_own); std::swap(T, o.T); std::swap(xJ, o.xJ); std::swap(lJ, o.lJ); std::swap(uJ, o.uJ);
        std::swap(cJ, o.cJ); std::swap(dJ, o.dJ); std::swap(d1, o.d1); std::swap(rowbuf, o.rowbuf); std::swap(vbuf, o.vbuf); std::swap(ebG, o.ebG);
        std::swap(ebS, o.ebS); std::swap(varJ, o.varJ); std::swap(statJ, o.statJ); std::swap(s0, o.s0); std::swap(sbase, o.sbase); std::swap(elist, o.elist);
    }
    static size_t bytes_for(int64_t mp_, int64_t cap_) {
        return sizeof(double) * (static_cast<size_t>(mp_) * cap_ + static_cast<size_t>(7 + TB_K + 2 * TB_EB) * cap_) + sizeof(int32_t) * 5 * static_cast<size_t>(cap_);
    }
    // capacity newcap, the first `keep` slots (and tableau columns) carried over
    int reserve(hipStream_t s, int64_t mp_, int64_t newcap, int64_t keep, BigArena *arena = nullptr) {
        if (newcap <= cap && mp_ == mp) return SX_OK;
        TbSlots nw;
        nw.mp = mp_;
        nw.cap = newcap;
        const size_t c = static_cast<size_t>(newcap);
        if (arena) {
            nw.T = static_cast<double *>(arena->take(sizeof(double) * static_cast<size_t>(mp_) * c));
            nw.T_own = nw.T == nullptr;
        }
#define TB_GET(field, count)                                                                                                       \
    if (sx_dmalloc(reinterpret_cast<void **>(&nw.field), sizeof(*nw.field) * (count)) != hipSuccess) {                               \
        sx_set_error("hipMalloc of %zu bytes failed in the sparse crossover (tableau of %lld columns over %lld positions)",        \
                     sizeof(*nw.field) * (count), (long long)newcap, (long long)mp_);                                              \
        return SX_ERR_NOMEM;                                                                                                       \
    }
        if (!nw.T) TB_GET(T, static_cast<size_t>(mp_) * c)
        TB_GET(xJ, c) TB_GET(lJ, c) TB_GET(uJ, c) TB_GET(cJ, c) TB_GET(dJ, c) TB_GET(d1, c) TB_GET(rowbuf, c)
        TB_GET(vbuf, static_cast<size_t>(TB_K) * c) TB_GET(ebG, static_cast<size_t>(TB_EB) * c) TB_GET(ebS, static_cast<size_t>(TB_EB) * c)
        TB_GET(varJ, c) TB_GET(statJ, c) TB_GET(s0, c) TB_GET(sbase, c) TB_GET(elist, c)
#undef TB_GET
        if (keep > 0) {
            SX_REQUIRE(mp_ == mp && keep <= cap, "internal: tableau columns cannot be carried over");
            const size_t k = static_cast<size_t>(keep);
            SX_HIP(hipMemcpyAsync(nw.T, T, sizeof(double) * static_cast<size_t>(mp_) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.xJ, xJ, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.lJ, lJ, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.uJ, uJ, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.cJ, cJ, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.dJ, dJ, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.varJ, varJ, sizeof(int32_t) * k, hipMemcpyDeviceToDevice, s));
            SX_HIP(hipMemcpyAsync(nw.statJ, statJ, sizeof(int32_t) * k, hipMemcpyDeviceToDevice, s));
        }
        SX_HIP(hipMemsetAsync(nw.s0, 0, sizeof(int32_t) * c, s));       // no pending update anywhere
        SX_HIP(hipMemsetAsync(nw.sbase, 0xFF, sizeof(int32_t) * c, s)); // (-1: every slot's base column is T's)
        SX_HIP(hipStreamSynchronize(s));
        swap_with(nw); // (nw's destructor frees the old arrays)
        return SX_OK;
    }
};

// rows of a sparse matrix from (row, index, value) triplets (counting sort, stable); `list` != nullptr: only the rows
// that hold an entry, listed in ascending order
struct HostRows {
    std::vector<int64_t> ptr;
    std::vector<int32_t> idx, list;
    std::vector<double> val;
};
void rows_from_triplets(int64_t nrows, const std::vector<int32_t> &row, const std::vector<int32_t> &idx, const std::vector<double> &val, bool compact,
                        HostRows &out) {
    std::vector<int64_t> cnt(static_cast<size_t>(nrows) + 1, 0);
    for (int32_t r : row) ++cnt[static_cast<size_t>(r) + 1];
    for (int64_t r = 0; r < nrows; ++r) cnt[r + 1] += cnt[r];
    out.idx.assign(row.size(), 0);
    out.val.assign(row.size(), 0.0);
    std::vector<int64_t> at(cnt.begin(), cnt.end() - 1);
    for (size_t e = 0; e < row.size(); ++e) {
        const int64_t o = at[row[e]]++;
        out.idx[o] = idx[e];
        out.val[o] = val[e];
    }
    out.ptr.clear();
    out.list.clear();
    if (!compact) {
        out.ptr = cnt;
        return;
    }
    out.ptr.push_back(0);
    for (int64_t r = 0; r < nrows; ++r)
        if (cnt[r + 1] > cnt[r]) {
            out.list.push_back(static_cast<int32_t>(r));
            out.ptr.push_back(cnt[r + 1]);
        }
}

} // namespace

// One entry for the three ways the sparse crossover is started: from the first-order point alone (vbasis_in == NULL: the
// basis is guessed from the margins), or from a given basis (the reference's warm-started final solve,
// lp_methods/algorithms.py:69-74; sx_crossover_band_dev passes NULL).
namespace {
int crossover_band_impl(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                        const double *u, const uint8_t *row_is_lt, const double *x_start, const int8_t *vbasis_in,
                        const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol, double *x_out,
                        double *y_out, int8_t *vbasis_out, int8_t *cbasis_out, sx_simplex_result *result, bool probe_only) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && b && c && l && u && x_start && result, "NULL argument");
    SX_REQUIRE((vbasis_in == nullptr) == (cbasis_in == nullptr), "vbasis_in and cbasis_in come together");
    SX_REQUIRE(A->csr_ptr && A->csc_ptr, "the sparse crossover needs both layouts of A");
    const int64_t m = A->m, n = A->n;
    SX_REQUIRE(m > 0 && n > 0 && m + n < 2000000000LL, "problem size");
    const bool trace = getenv("SX_SPX_TRACE") != nullptr;
    hipStream_t s = ctx->stream;
    auto now = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    const double t_begin = now();
    *result = sx_simplex_result{};
    result->status = 4;
    const int64_t NV = n + m;
    if (max_iter <= 0) max_iter = 50 * (m + n);
    if (!(feas_tol > 0)) feas_tol = 1e-7;
    if (!(opt_tol > 0)) opt_tol = 1e-7;
    DevBufs dev; // lives as long as the call
    if (probe_only && !vbasis_in) {
        // the question "would the band LU take the matched basis?" has a sufficient answer in the matrix alone: no column of
        // A is taller (over the rows that are not dense) than a band the LU takes, so no basis is.  One kernel instead of the
        // host copies, the row order and the matching (18 ms at 1e5 rows); a matrix that fails it gets the full answer below
        const double avg_row_q = static_cast<double>(A->nnz) / static_cast<double>(m);
        const int64_t dense_thr_q = std::max<int64_t>(24, static_cast<int64_t>(6.0 * avg_row_q));
        int *d_q = nullptr;
        SX_TRY(dev.get(2, &d_q));
        SX_HIP(hipMemsetAsync(d_q, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(k_tb_colspan, dim3(gridof(std::max(m, n))), dim3(TB_WG), 0, s, m, n, A->csc_ptr, A->csc_idx, A->csr_ptr, dense_thr_q, d_q);
        int hq[2] = {0, 0};
        SX_HIP(hipMemcpyAsync(hq, d_q, sizeof(hq), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        if (trace) fprintf(stderr, "[sx_crossover_band] probe: tallest column %d rows outside the %d dense rows\n", hq[0], hq[1]);
        if (hq[1] <= 16384 / 2 && sx_bandlu_supports(hq[0], hq[0])) {
            result->status = 0;
            return SX_OK;
        }
    }
    // ------------------------------------------------------------------ host copies (once)
    std::vector<int64_t> cptr, rptr;
    std::vector<int32_t> cidx;
    std::vector<double> cval, hb, hc, hl, hu, hx, hslack;
    std::vector<uint8_t> hlt(static_cast<size_t>(m), 0);
    std::vector<int8_t> vb_in, cb_in;
    SX_TRY(down(s, cptr, A->csc_ptr, static_cast<size_t>(n) + 1));
    SX_TRY(down(s, rptr, A->csr_ptr, static_cast<size_t>(m) + 1));
    SX_TRY(down(s, cidx, A->csc_idx, static_cast<size_t>(A->nnz)));
    SX_TRY(down(s, cval, A->csc_val, static_cast<size_t>(A->nnz)));
    SX_TRY(down(s, hb, b, static_cast<size_t>(m)));
    SX_TRY(down(s, hc, c, static_cast<size_t>(n)));
    SX_TRY(down(s, hl, l, static_cast<size_t>(n)));
    SX_TRY(down(s, hu, u, static_cast<size_t>(n)));
    SX_TRY(down(s, hx, x_start, static_cast<size_t>(n)));
    if (row_is_lt) SX_HIP(hipMemcpyAsync(hlt.data(), row_is_lt, static_cast<size_t>(m), hipMemcpyDeviceToHost, s));
    if (vbasis_in) {
        SX_TRY(down(s, vb_in, vbasis_in, static_cast<size_t>(n)));
        SX_TRY(down(s, cb_in, cbasis_in, static_cast<size_t>(m)));
    }
    double *d_tmpm = nullptr, *d_tmpn = nullptr, *d_rc = nullptr;
    SX_TRY(dev.get(static_cast<size_t>(m), &d_tmpm));
    SX_TRY(dev.get(static_cast<size_t>(n), &d_tmpn));
    SX_TRY(dev.get(static_cast<size_t>(n), &d_rc));
    SX_HIP(hipStreamSynchronize(s));
    for (int64_t j = 0; j < n; ++j) hx[j] = std::min(std::max(hx[j], hl[j]), hu[j]);
    auto var_lo = [&](int64_t v) { return v < n ? hl[v] : 0.0; };
    auto var_up = [&](int64_t v) { return v < n ? hu[v] : (hlt[v - n] ? INFINITY : 0.0); };
    auto var_cost = [&](int64_t v) { return v < n ? hc[v] : 0.0; };
    // ------------------------------------------------------------------ dense rows, band rows (once)
    const double avg_row = static_cast<double>(A->nnz) / static_cast<double>(m);
    const int64_t dense_thr = std::max<int64_t>(24, static_cast<int64_t>(6.0 * avg_row));
    std::vector<int32_t> eqidx(static_cast<size_t>(m)), rows_band, rows_dense;
    for (int64_t i = 0; i < m; ++i) {
        if (rptr[i + 1] - rptr[i] > dense_thr) rows_dense.push_back(static_cast<int32_t>(i));
        else rows_band.push_back(static_cast<int32_t>(i));
    }
    const int64_t m1 = static_cast<int64_t>(rows_band.size()), ndr = static_cast<int64_t>(rows_dense.size());
    constexpr int64_t MAX_NB = 16384; // rows of the Schur complement the dense LU takes
    if (ndr > MAX_NB) {
        sx_set_error("%lld rows with more than %lld entries: the border of the basis would pass the dense LU's %lld rows", (long long)ndr,
                     (long long)dense_thr, (long long)MAX_NB);
        return SX_ERR_UNSUPPORTED;
    }
    std::vector<int32_t> rowb(static_cast<size_t>(m), -1);
    // ---- the order of the band rows.  Natural order first (staged models are written stage after stage); when
    // that leaves some column taller than a band can be, a Cuthill-McKee order of the rows (breadth first over
    // "shares a column with", from a far end found by one extra sweep) -- rows of a model that was shuffled, or
    // written variable by variable, fall back into stages.
    {
        auto tallest = [&](const std::vector<int32_t> &order) {
            std::vector<int32_t> where(static_cast<size_t>(m), -1);
            for (size_t k = 0; k < order.size(); ++k) where[order[k]] = static_cast<int32_t>(k);
            int32_t worst = 0;
            for (int64_t j = 0; j < n; ++j) {
                int32_t lo = INT32_MAX, hi = -1;
                for (int64_t k = cptr[j]; k < cptr[j + 1]; ++k) {
                    const int32_t ib = where[cidx[k]];
                    if (ib >= 0) {
                        lo = std::min(lo, ib);
                        hi = std::max(hi, ib);
                    }
                }
                if (hi >= 0) worst = std::max(worst, hi - lo);
            }
            return worst;
        };
        const int32_t tall_nat = tallest(rows_band);
        if (tall_nat > 600 && m1 > 1) {
            std::vector<int32_t> ridx;
            SX_TRY(down(s, ridx, A->csr_idx, static_cast<size_t>(A->nnz)));
            SX_HIP(hipStreamSynchronize(s));
            std::vector<uint8_t> is_band(static_cast<size_t>(m), 0);
            for (int32_t r : rows_band) is_band[r] = 1;
            const int64_t col_cap = std::max<int64_t>(64, static_cast<int64_t>(16.0 * static_cast<double>(A->nnz) / static_cast<double>(n)));
            auto sweep = [&](int32_t start, std::vector<int32_t> &order) { // breadth first over all components, `start` first
                order.clear();
                std::vector<uint8_t> seen(static_cast<size_t>(m), 0), used(static_cast<size_t>(n), 0);
                size_t next_seed = 0;
                int32_t seed = start;
                while (static_cast<int64_t>(order.size()) < m1) {
                    if (seed < 0) {
                        while (next_seed < rows_band.size() && seen[rows_band[next_seed]]) ++next_seed;
                        if (next_seed >= rows_band.size()) break;
                        seed = rows_band[next_seed];
                    }
                    size_t head_q = order.size();
                    seen[seed] = 1;
                    order.push_back(seed);
                    for (; head_q < order.size(); ++head_q) {
                        const int32_t i = order[head_q];
                        for (int64_t k = rptr[i]; k < rptr[i + 1]; ++k) {
                            const int32_t j = ridx[k];
                            if (used[j]) continue;
                            used[j] = 1;
                            if (cptr[j + 1] - cptr[j] > col_cap) continue; // a column that touches everything orders nothing
                            for (int64_t q = cptr[j]; q < cptr[j + 1]; ++q) {
                                const int32_t i2 = cidx[q];
                                if (is_band[i2] && !seen[i2]) {
                                    seen[i2] = 1;
                                    order.push_back(i2);
                                }
                            }
                        }
                    }
                    seed = -1;
                }
            };
            std::vector<int32_t> o1, o2;
            sweep(rows_band[0], o1);
            sweep(o1.empty() ? rows_band[0] : o1.back(), o2); // from the far end of the first sweep
            if (static_cast<int64_t>(o2.size()) == m1) {
                const int32_t tall_cm = tallest(o2);
                if (trace) fprintf(stderr, "[sx_crossover_band] tallest column: %d rows in natural order, %d in Cuthill-McKee order\n", tall_nat, tall_cm);
                if (tall_cm < tall_nat) rows_band = o2;
            }
        }
    }
    for (int64_t k = 0; k < m1; ++k) {
        rowb[rows_band[k]] = static_cast<int32_t>(k);
        eqidx[rows_band[k]] = static_cast<int32_t>(k);
    }
    for (int64_t k = 0; k < ndr; ++k) eqidx[rows_dense[k]] = static_cast<int32_t>(m1 + k);
    int32_t *d_eqidx = nullptr;
    SX_TRY(dev.get(eqidx.size(), &d_eqidx));
    SX_TRY(up(s, d_eqidx, eqidx));
    // ------------------------------------------------------------------ the first basic set
    const double MARGIN = 1e-7;
    std::vector<uint8_t> is_basic(static_cast<size_t>(NV), 0); // the variables the next factorisation has to hold
    std::vector<int32_t> tracked;                               // columns of the tableau that are not basic
    std::vector<int8_t> atup(static_cast<size_t>(NV), 0);
    std::vector<double> margin; // of the first guess (empty for a given basis): the border takes its columns in this order
    // smallest pivot the two LUs accept.  A GUESSED basis (the m largest margins of a first-order point) may hold columns
    // that are all but dependent -- its basic solution then lies far from the point and phase 1 pays for it pivot by pivot
    // -- so its LUs set such columns aside (they stay superbasic, a logical takes their place); a basis the simplex has
    // reached, or one that was given, is kept unless it is singular to working accuracy
    const double tol_band_guess = getenv("SX_BAND_PIVTOL_B") ? atof(getenv("SX_BAND_PIVTOL_B")) : 1e-3;
    const double tol_schur_guess = getenv("SX_BAND_PIVTOL_S") ? atof(getenv("SX_BAND_PIVTOL_S")) : 1e-1;
    const double tol_band_keep = 1e-7, tol_schur_keep = 1e-9;
    if (vbasis_in) { // a given basis: codes as in output.py (0 basic, -1 at lower, -2 at upper, -3 superbasic at x_start)
        for (int64_t j = 0; j < n; ++j) {
            const int code = vb_in[j];
            if (code == 0) is_basic[j] = 1;
            else if (code == -3) tracked.push_back(static_cast<int32_t>(j));
            else {
                double v = code == -2 ? hu[j] : hl[j];
                if (std::isinf(v)) v = code == -2 ? hl[j] : hu[j];
                if (std::isinf(v)) v = 0.0;
                hx[j] = v;
            }
        }
        for (int64_t i = 0; i < m; ++i)
            if (cb_in[i] == 0) is_basic[n + i] = 1;
    } else {
        SX_TRY(sx_score_rows_dev(ctx, A, x_start, b, nullptr, 0.0, d_tmpm, nullptr)); // slack = b - A x
        SX_TRY(down(s, hslack, d_tmpm, static_cast<size_t>(m)));
        SX_HIP(hipStreamSynchronize(s));
        margin.assign(static_cast<size_t>(NV), -1.0);
        std::vector<int64_t> cand;
        for (int64_t j = 0; j < n; ++j) {
            const double mg = std::min(hx[j] - hl[j], hu[j] - hx[j]);
            if (std::isinf(hl[j]) && std::isinf(hu[j])) margin[j] = 1e300;
            else if (mg > MARGIN * (1.0 + std::fabs(hx[j]))) margin[j] = mg;
            if (margin[j] > 0) cand.push_back(j);
        }
        for (int64_t i = 0; i < m; ++i)
            if (hlt[i] && hslack[i] > MARGIN * (1.0 + std::fabs(hb[i]))) {
                margin[n + i] = hslack[i];
                cand.push_back(n + i);
            }
        // the m candidates with the largest margins may sit in the basis, the others start superbasic
        if (static_cast<int64_t>(cand.size()) > m)
            std::nth_element(cand.begin(), cand.begin() + m, cand.end(),
                             [&](int64_t a, int64_t bb) { return margin[a] > margin[bb] || (margin[a] == margin[bb] && a < bb); });
        for (size_t k = 0; k < cand.size(); ++k) {
            if (static_cast<int64_t>(k) < m) is_basic[cand[k]] = 1;
            else tracked.push_back(static_cast<int32_t>(cand[k]));
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] m=%lld n=%lld: %zu interior candidates, band rows %lld, dense rows %lld\n", (long long)m, (long long)n, cand.size(), (long long)m1, (long long)ndr);
    }
    if (trace) fprintf(stderr, "[sx_crossover_band] host copies, row order and first basic set at %.1f ms\n", now() - t_begin);
    // ------------------------------------------------------------------ the big blocks, allocated beside the matching
    BigArena arena;
    {
        size_t free_b = 0, total_b = 0;
        SX_HIP(hipMemGetInfo(&free_b, &total_b));
        // (estimates before the matching: border = dense rows + up to 32 separators of ~128 rows; a twelfth of the border is
        //  set aside by the dense LU of a guess -- 1,032 of 13,771 at config-5 size, 158 of 2,299 at the headline size)
        const double mp_est = static_cast<double>(m) + static_cast<double>(ndr) + 33.0 * 400.0 + 2048.0;
        const double nb_est = static_cast<double>(ndr) + 32.0 * 128.0 + 1024.0;
        const double track_est = static_cast<double>(tracked.size()) + nb_est / 12.0;
        const double epoch_est = std::min(std::min(20000.0, std::min(16.0e9, 0.4 * static_cast<double>(free_b)) / (8.0 * mp_est)), 4.0 * (track_est + nb_est / 8.0) + 2048.0);
        const double want = 8.0 * mp_est * (epoch_est + 2.0 + 1.5 * track_est + 712.0);
        if (want > 2.0e9 && want < 0.6 * static_cast<double>(free_b) && !getenv("SX_BAND_NO_ARENA")) arena.start(ctx, static_cast<size_t>(want));
    }
    // ------------------------------------------------------------------ small per-call blocks
    TbState *d_st = nullptr;
    SX_TRY(dev.get(1, &d_st));
    int32_t *d_pr = nullptr;
    double *d_gu = nullptr, *d_ebM = nullptr;
    SX_TRY(dev.get(static_cast<size_t>(TB_K), &d_pr));
    SX_TRY(dev.get(static_cast<size_t>(TB_K), &d_gu));
    SX_TRY(dev.get(static_cast<size_t>(TB_EB) * TB_EB, &d_ebM));

    long long tot_iters = 0, tot_pivots = 0, tot_flips = 0, tot_degen = 0;
    int64_t added_total = 0;
    int epochs = 0, final_status = 4, bad_epochs = 0, check_epochs = 0;
    std::vector<double> hy(static_cast<size_t>(m), 0.0);
    std::vector<int8_t> vstat(static_cast<size_t>(NV), 0); // 0 non-basic at a bound, 1 basic, 2 tracked
    std::vector<double> xlog(static_cast<size_t>(m), 0.0);  // values of tracked logicals
    std::vector<int64_t> prev_pos;                          // band position -> the variable that sat there and never moved
    double viol_max = 0.0, resid_max = 0.0;
    size_t peak_bytes = 0;
    bool strict_next = false, strict_tried = false;

    for (;;) {
        ++epochs;
        // the basic set is a guess from the margins of the point -- or a given basis whose basic solution turned out far
        // outside the bounds (below): the LUs then set aside what is all but dependent
        const bool guessed = (epochs == 1 && !vbasis_in) || strict_next;
        strict_next = false;
        DevBufs edev; // this epoch's device arrays
        // ---------------------------------------------------------------- columns to rows: who sits on which band row
        // Every band row gets at most ONE variable, whose position is the row's: a variable that sat there in the last
        // factorisation and is still basic stays; a row whose own logical is basic keeps it; the other basic columns are
        // matched to the free rows greedily by entry size, largest first (large entries on the diagonal, and a band no
        // wider than a column is tall: a column only ever sits on a row it has an entry in).  A row no column takes holds
        // a PLACEHOLDER (unit vector; border row "its value = 0"), a column that finds no row goes to the BORDER.
        std::vector<int64_t> bvar(static_cast<size_t>(m1), -1);
        std::vector<uint8_t> placed(static_cast<size_t>(NV), 0);
        if (!prev_pos.empty())
            for (int64_t p = 0; p < m1; ++p) {
                const int64_t v = prev_pos[p];
                if (v >= 0 && is_basic[v] && !placed[v]) {
                    bvar[p] = v;
                    placed[v] = 1;
                }
            }
        for (int64_t p = 0; p < m1; ++p) {
            const int64_t v = n + rows_band[p];
            if (bvar[p] < 0 && is_basic[v] && !placed[v]) {
                bvar[p] = v;
                placed[v] = 1;
            }
        }
        {
            struct Ent {
                double a;
                int32_t ib;
                int64_t var;
            };
            // order: size descending, then variable, then band row -- a stable LSD radix sort on the inverted bits of the
            // (positive, finite or infinite) size, 16 bits a pass (8e5 candidates in ~10 ms against ~40 for std::sort with the
            // three-way comparison) -- then every candidate in that order takes its row if both are still free
            auto place_sorted = [&](std::vector<Ent> &ents) {
                std::vector<Ent> tmp(ents.size());
                std::vector<uint32_t> hist(65536);
                auto key = [](const Ent &e) {
                    uint64_t bits;
                    std::memcpy(&bits, &e.a, sizeof(bits));
                    return ~bits;
                };
                for (int pass = 0; pass < 4 && ents.size() > 1; ++pass) {
                    const int sh = 16 * pass;
                    std::fill(hist.begin(), hist.end(), 0u);
                    for (const Ent &e : ents) ++hist[(key(e) >> sh) & 0xFFFF];
                    uint32_t run = 0;
                    for (uint32_t &hh_ : hist) {
                        const uint32_t cc = hh_;
                        hh_ = run;
                        run += cc;
                    }
                    for (const Ent &e : ents) tmp[hist[(key(e) >> sh) & 0xFFFF]++] = e;
                    ents.swap(tmp);
                }
                for (const Ent &e : ents)
                    if (!placed[e.var] && bvar[e.ib] < 0) {
                        bvar[e.ib] = e.var;
                        placed[e.var] = 1;
                    }
            };
            // Two rounds (the one sort over every entry of every column was 0.16 of the 0.41 s this set-up took at 1e6 rows:
            // 7e6 candidates of 24 bytes, four passes): first every column's LARGEST entry in a free row competes -- one
            // candidate per column, and in a column-dominant basis the winner nearly everywhere --, then the columns that lost
            // theirs compete with all their entries for the rows that are left
            std::vector<Ent> ents;
            for (int64_t j = 0; j < n; ++j) {
                if (!is_basic[j] || placed[j]) continue;
                Ent best{0.0, -1, j};
                for (int64_t k = cptr[j]; k < cptr[j + 1]; ++k) {
                    const int32_t ib = rowb[cidx[k]];
                    const double a = std::fabs(cval[k]);
                    if (ib >= 0 && bvar[ib] < 0 && a > 1e-6 && (a > best.a || (a == best.a && ib < best.ib))) best = Ent{a, ib, j};
                }
                if (best.ib >= 0) ents.push_back(best);
            }
            place_sorted(ents);
            ents.clear();
            for (int64_t j = 0; j < n; ++j) {
                if (!is_basic[j] || placed[j]) continue;
                const size_t first = ents.size();
                for (int64_t k = cptr[j]; k < cptr[j + 1]; ++k) {
                    const int32_t ib = rowb[cidx[k]];
                    const double a = std::fabs(cval[k]);
                    if (ib >= 0 && bvar[ib] < 0 && a > 1e-6) ents.push_back(Ent{a, ib, j});
                }
                // a column's candidates by band row (a handful: insertion sort), so that the stable sort leaves equal sizes
                // in (variable, band row) order
                for (size_t q = first + 1; q < ents.size(); ++q) {
                    const Ent e = ents[q];
                    size_t r = q;
                    for (; r > first && ents[r - 1].ib > e.ib; --r) ents[r] = ents[r - 1];
                    ents[r] = e;
                }
            }
            place_sorted(ents);
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: columns matched to rows at %.1f ms\n", epochs, now() - t_begin);
        // ---------------------------------------------------------------- blocks: the band cut at SEPARATORS
        // A panel of the band LU is a chain of 32 dependent column steps on one workgroup (53 us at kl + ku = 230), a band of
        // 1e5 rows 3,100 of them in a row.  Cut every RL + W positions: W = max(kl, ku) positions (rows AND the columns matched
        // to them) go to the border as separators -- no column left of a separator reaches a row right of it, or the other way
        // round -- so the P blocks between them are independent band matrices, factored side by side
        // (sx_bandlu_factor_blocks_dev; identity padding of kl + ku + 32 positions keeps their memory apart).  The Schur
        // complement grows by (P - 1) W rows.
        const int64_t m1_all = m1;
        int64_t P = 1, RL = m1_all, Wsep = 0, PAD = 0;
        {
            int kl0 = 0, ku0 = 0;
            for (int64_t p = 0; p < m1_all; ++p) {
                const int64_t v = bvar[p];
                if (v < 0 || v >= n) continue;
                for (int64_t k = cptr[v]; k < cptr[v + 1]; ++k) {
                    const int32_t ib = rowb[cidx[k]];
                    if (ib >= 0) {
                        kl0 = std::max<int>(kl0, ib - static_cast<int>(p));
                        ku0 = std::max<int>(ku0, static_cast<int>(p) - ib);
                    }
                }
            }
            const int64_t w = std::max(kl0, ku0), pad = ((kl0 + ku0 + 32 + 31) / 32) * 32;
            int64_t want = getenv("SX_BAND_BLOCKS") ? atoll(getenv("SX_BAND_BLOCKS")) : std::min<int64_t>(32, m1_all / 8192);
            while (want > 1) { // (2,048 rows of the border are left to placeholders and columns without a row)
                const int64_t rl = ((m1_all - (want - 1) * w) / want) / 32 * 32;
                if (w > 0 && rl >= 4 * pad && ndr + (want - 1) * w + 2048 <= MAX_NB) break;
                --want;
            }
            if (want > 1) {
                P = want;
                Wsep = w;
                PAD = pad;
                RL = ((m1_all - (P - 1) * w) / P) / 32 * 32;
            }
        }
        const int64_t RL_last = m1_all - (P - 1) * (RL + Wsep), stride = RL + PAD;
        const int64_t m1e = (P - 1) * stride + RL_last, nsep = (P - 1) * Wsep;
        std::vector<int64_t> bvar_e(static_cast<size_t>(m1e), -2); // (-2: padding, -1: placeholder)
        std::vector<int32_t> posrow_e(static_cast<size_t>(m1e), -1), old_of_new(static_cast<size_t>(m1e), -1), rows_dense_e(rows_dense);
        std::vector<int32_t> rowb_e(static_cast<size_t>(m), -1), eqidx_e(static_cast<size_t>(m), 0);
        rows_dense_e.resize(static_cast<size_t>(ndr + nsep));
        for (int64_t p = 0; p < m1_all; ++p) {
            const int64_t b = std::min<int64_t>(p / (RL + Wsep), P - 1), off = p - b * (RL + Wsep);
            const int32_t row = rows_band[p];
            if (b < P - 1 && off >= RL) { // a separator: its row joins the dense rows, its column goes to the border
                rows_dense_e[static_cast<size_t>(ndr + b * Wsep + off - RL)] = row;
                if (bvar[p] >= 0) placed[bvar[p]] = 0;
                continue;
            }
            const int64_t np = b * stride + off;
            bvar_e[np] = bvar[p];
            posrow_e[np] = row;
            old_of_new[np] = static_cast<int32_t>(p);
            rowb_e[row] = static_cast<int32_t>(np);
            eqidx_e[row] = static_cast<int32_t>(np);
        }
        for (size_t k = 0; k < rows_dense_e.size(); ++k) eqidx_e[rows_dense_e[k]] = static_cast<int32_t>(m1e + static_cast<int64_t>(k));
        int32_t *d_eqidx_e = nullptr;
        SX_TRY(edev.get(eqidx_e.size(), &d_eqidx_e));
        SX_TRY(up(s, d_eqidx_e, eqidx_e));
        { // from here on the geometry is this epoch's: band positions with padding, dense rows with the separators' rows
        const int64_t m1 = m1e, ndr = static_cast<int64_t>(rows_dense_e.size());
        std::vector<int64_t> &bvar = bvar_e;
        const std::vector<int32_t> &rows_band = posrow_e, &rows_dense = rows_dense_e, &rowb = rowb_e, &eqidx = eqidx_e;
        const int32_t *d_eqidx = d_eqidx_e;
        // band triplets (a placeholder or a logical: the unit vector of the row), band widths
        std::vector<int32_t> trow, tcol;
        std::vector<double> tval;
        trow.reserve(static_cast<size_t>(A->nnz) / 2 + static_cast<size_t>(m1));
        tcol.reserve(trow.capacity());
        tval.reserve(trow.capacity());
        int kl = 0, ku = 0;
        for (int64_t p = 0; p < m1; ++p) {
            const int64_t v = bvar[p];
            if (v < 0 || v >= n) {
                trow.push_back(static_cast<int32_t>(p));
                tcol.push_back(static_cast<int32_t>(p));
                tval.push_back(1.0);
                continue;
            }
            for (int64_t k = cptr[v]; k < cptr[v + 1]; ++k) {
                const int32_t ib = rowb[cidx[k]];
                if (ib >= 0) {
                    trow.push_back(ib);
                    tcol.push_back(static_cast<int32_t>(p));
                    tval.push_back(cval[k]);
                    kl = std::max<int>(kl, ib - static_cast<int>(p));
                    ku = std::max<int>(ku, static_cast<int>(p) - ib);
                }
            }
        }
        if (!sx_bandlu_supports(kl, ku)) {
            sx_set_error("the basis is not a band matrix in the order of the rows (kl = %d, ku = %d after setting %lld dense rows aside)", kl, ku,
                         (long long)ndr);
            return SX_ERR_UNSUPPORTED;
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: matching and band assembly done at %.1f ms\n", epochs, now() - t_begin);
        if (probe_only) { // (sx_crossover_band_probe_dev: the matched basis is a band the LU takes -- that was the question)
            result->status = 0;
            return SX_OK;
        }
        // ---------------------------------------------------------------- factor B11
        sx_bandlu *lu = nullptr;
        sx_denselu *dl = nullptr;
        struct FactorGuard {
            sx_bandlu *&h;
            sx_denselu *&d;
            ~FactorGuard() {
                if (h) sx_bandlu_destroy(h);
                if (d) sx_denselu_destroy(d);
            }
        } fguard{lu, dl};
        std::vector<int32_t> ph_row(static_cast<size_t>(m1), -1); // the row whose unit vector a placeholder is
        int64_t nrep = 0;
        if (m1 > 0) {
            int32_t *d_trow = nullptr, *d_tcol = nullptr;
            double *d_tval = nullptr;
            DevBufs tdev;
            SX_TRY(tdev.get(trow.size(), &d_trow));
            SX_TRY(tdev.get(tcol.size(), &d_tcol));
            SX_TRY(tdev.get(tval.size(), &d_tval));
            SX_TRY(up(s, d_trow, trow));
            SX_TRY(up(s, d_tcol, tcol));
            SX_TRY(up(s, d_tval, tval));
            SX_TRY(sx_bandlu_create_dev(ctx, m1, kl, ku, static_cast<int64_t>(tval.size()), d_trow, d_tcol, d_tval, &lu));
            std::vector<int32_t> rep(static_cast<size_t>(m1)), piv(static_cast<size_t>(m1));
            SX_TRY(sx_bandlu_factor_blocks_dev(lu, guessed ? tol_band_guess : tol_band_keep, static_cast<int>(P), P > 1 ? stride : m1, P > 1 ? RL : m1, P > 1 ? RL_last : m1, &nrep,
                                                rep.data(), piv.data()));
            // a replaced column stands for the unit vector of the row that sat on its diagonal: a placeholder; the column
            // itself goes to the border
            std::vector<int32_t> rowof(static_cast<size_t>(m1));
            std::iota(rowof.begin(), rowof.end(), 0);
            for (int64_t j = 0; j < m1; ++j) {
                if (rep[j]) {
                    if (bvar[j] >= 0) placed[bvar[j]] = 0;
                    bvar[j] = -1;
                    ph_row[j] = rows_band[rowof[j]];
                } else {
                    if (piv[j] != j) std::swap(rowof[j], rowof[piv[j]]);
                    if (bvar[j] == -1) ph_row[j] = rows_band[j];
                }
            }
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: band LU done at %.1f ms (kl=%d ku=%d, %lld blocks of %lld positions with separators of %lld, %lld columns set aside)\n", epochs, now() - t_begin, kl, ku, (long long)P, (long long)RL, (long long)Wsep, (long long)nrep);
        // ---------------------------------------------------------------- the border: as many columns as rows
        auto untrack = [&](int64_t v) { tracked.erase(std::remove(tracked.begin(), tracked.end(), static_cast<int32_t>(v)), tracked.end()); };
        std::vector<int64_t> bcol;
        for (int64_t v = 0; v < NV; ++v)
            if (is_basic[v] && !placed[v]) bcol.push_back(v);
        int64_t h = 0;
        for (int64_t p = 0; p < m1; ++p) h += bvar[p] == -1;
        {
            // too few columns for the border rows (or a border beyond the dense LU): placeholders become their row's own
            // logical, then the dense rows' logicals fill in; too many: the surplus leaves the basis (superbasic)
            std::vector<int32_t> gone;
            for (int64_t p = 0; p < m1 && (static_cast<int64_t>(bcol.size()) < ndr + h || ndr + h > MAX_NB); ++p) {
                if (bvar[p] != -1) continue;
                const int64_t w = n + ph_row[p];
                if (is_basic[w]) continue;
                bvar[p] = w;
                is_basic[w] = 1;
                placed[w] = 1;
                gone.push_back(static_cast<int32_t>(w));
                --h;
            }
            for (int64_t k = 0; k < ndr && static_cast<int64_t>(bcol.size()) < ndr + h; ++k) {
                const int64_t w = n + rows_dense[k];
                if (is_basic[w]) continue;
                is_basic[w] = 1;
                bcol.push_back(w);
                gone.push_back(static_cast<int32_t>(w));
            }
            if (!gone.empty()) {
                std::vector<uint8_t> g(static_cast<size_t>(NV), 0);
                for (int32_t w : gone) g[w] = 1;
                tracked.erase(std::remove_if(tracked.begin(), tracked.end(), [&](int32_t v) { return g[v] != 0; }), tracked.end());
            }
            if (guessed && !margin.empty()) // the surest columns first: a column the dense LU sets aside is one of the doubtful ones
                std::sort(bcol.begin(), bcol.end(), [&](int64_t a, int64_t bb) { return margin[a] > margin[bb] || (margin[a] == margin[bb] && a < bb); });
            else std::sort(bcol.begin(), bcol.end());
            while (static_cast<int64_t>(bcol.size()) > ndr + h) {
                const int64_t v = bcol.back();
                bcol.pop_back();
                is_basic[v] = 0;
                tracked.push_back(static_cast<int32_t>(v));
            }
            if (static_cast<int64_t>(bcol.size()) != ndr + h || ndr + h > MAX_NB) {
                sx_set_error("the border of the basis has %lld rows and %zu columns (at most %lld rows)", (long long)(ndr + h), bcol.size(), (long long)MAX_NB);
                return SX_ERR_UNSUPPORTED;
            }
        }
        const int64_t nb = ndr + h, mp = m1 + nb;
        // the eta file of this epoch -- allocated now because its memory doubles as the work block of the Schur complement's
        // assembly (a block of 24 GB allocated and freed for that alone cost 0.3 s at 1e5 rows and 1.2 s at 1e6: hipFree of a
        // touched block).  Basis changes before the basis is factored afresh: a few times the columns that can move, within
        // 16 GB; a fresh factorisation keeps every basic variable (bordered form), so the file need not be long
        double *d_eta = nullptr;
        int32_t *d_eta_r = nullptr;
        int64_t EPOCH = 0;
        {
            size_t free_b = 0, total_b = 0;
            SX_HIP(hipMemGetInfo(&free_b, &total_b));
            const int64_t by_memory = std::max<int64_t>(64, static_cast<int64_t>(std::min(16.0e9, 0.4 * static_cast<double>(free_b)) / (8.0 * static_cast<double>(mp))) - 1);
            EPOCH = std::min<int64_t>(std::min<int64_t>(20000, by_memory), 4 * static_cast<int64_t>(tracked.size() + static_cast<size_t>(nb) / 8) + 2048);
            if (const char *e = getenv("SX_BAND_EPOCH")) EPOCH = std::max<int64_t>(16, atoll(e));
            arena.reset(); // (what the last epoch held there is dead)
            d_eta = static_cast<double *>(arena.take(sizeof(double) * static_cast<size_t>(mp) * (EPOCH + 1)));
            if (!d_eta) SX_TRY(edev.get(static_cast<size_t>(mp) * (EPOCH + 1), &d_eta));
            SX_TRY(edev.get(static_cast<size_t>(EPOCH) + 2, &d_eta_r));
        }
        std::vector<int32_t> ph_pos;
        for (int64_t p = 0; p < m1; ++p)
            if (bvar[p] == -1) ph_pos.push_back(static_cast<int32_t>(p));
        // ---- B21: the dense rows' entries in the band columns, one unit entry per placeholder row
        SxBorderOps ops;
        ops.ctx = ctx;
        ops.m1 = m1;
        ops.nb = nb;
        ops.mp = mp;
        ops.lu = lu;
        ops.tiny = TB_TINY;
        auto rows_to_dev = [&](const HostRows &R, int64_t nrows, bool with_list, SxRowsDev &out) -> int {
            int64_t *dp = nullptr;
            int32_t *di = nullptr, *dlist = nullptr;
            double *dv = nullptr;
            SX_TRY(edev.get(R.ptr.size(), &dp));
            SX_TRY(edev.get(R.idx.size(), &di));
            SX_TRY(edev.get(R.val.size(), &dv));
            SX_TRY(up(s, dp, R.ptr));
            SX_TRY(up(s, di, R.idx));
            SX_TRY(up(s, dv, R.val));
            if (with_list) {
                SX_TRY(edev.get(R.list.size(), &dlist));
                SX_TRY(up(s, dlist, R.list));
            }
            out.nrows = nrows;
            out.ptr = dp;
            out.idx = di;
            out.val = dv;
            out.list = dlist;
            return SX_OK;
        };
        HostRows b21r, b21c, b12r, b12c;
        if (nb > 0) {
            std::vector<int32_t> er, ep;
            std::vector<double> ev;
            for (int64_t p = 0; p < m1; ++p) {
                const int64_t v = bvar[p];
                if (v < 0 || v >= n) continue;
                for (int64_t k = cptr[v]; k < cptr[v + 1]; ++k)
                    if (rowb[cidx[k]] < 0) {
                        er.push_back(eqidx[cidx[k]] - static_cast<int32_t>(m1));
                        ep.push_back(static_cast<int32_t>(p));
                        ev.push_back(cval[k]);
                    }
            }
            for (size_t t = 0; t < ph_pos.size(); ++t) {
                er.push_back(static_cast<int32_t>(ndr + static_cast<int64_t>(t)));
                ep.push_back(ph_pos[t]);
                ev.push_back(1.0);
            }
            rows_from_triplets(nb, er, ep, ev, false, b21r);
            rows_from_triplets(m1, ep, er, ev, true, b21c);
            SX_TRY(rows_to_dev(b21r, nb, false, ops.b21_rows));
            SX_TRY(rows_to_dev(b21c, static_cast<int64_t>(b21c.list.size()), true, ops.b21_cols));
        }
        // ---- the Schur complement, a block of border columns at a time: S[:, j] = (a2 - B21 B11^-1 a1) of column j
        int64_t nrep2 = 0;
        if (nb > 0) {
            {
                size_t free_b = 0, total_b = 0;
                SX_HIP(hipMemGetInfo(&free_b, &total_b));
                ops.v_limit = static_cast<size_t>(std::min(4.0e9, 0.05 * static_cast<double>(free_b)) / 8.0);
                if (getenv("SX_BAND_NO_V")) ops.v_failed = 1;
            }
            SX_TRY(sx_denselu_create_dev(ctx, nb, &dl));
            double *Sa = nullptr;
            int64_t Sld = 0;
            SX_TRY(sx_denselu_matrix(dl, &Sa, &Sld));
            // blocks of up to 4,096 columns (in the memory of the still empty eta file: a sparse band solve is a chain of panel steps per
            // group of 8 columns whatever the number of groups, 250 columns a launch kept 31 of 256 CUs busy), taken in the
            // order of the VARIABLES -- neighbours in that order live in neighbouring rows, so the 8 columns of a group share
            // their panels -- and written to the column of S the border's own order gives them
            const int64_t CH = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nb, 4096), EPOCH + 1));
            DevBufs sdev;
            double *Wc = d_eta; // (the eta file is empty until the simplex starts)
            int32_t *d_bvars = nullptr, *d_dest = nullptr;
            SX_TRY(sdev.get(static_cast<size_t>(nb), &d_bvars));
            SX_TRY(sdev.get(static_cast<size_t>(nb), &d_dest));
            std::vector<int32_t> order(static_cast<size_t>(nb)), bv32(static_cast<size_t>(nb));
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int32_t a, int32_t bb) { return bcol[a] < bcol[bb]; });
            for (int64_t t = 0; t < nb; ++t) bv32[t] = static_cast<int32_t>(bcol[order[t]]);
            SX_TRY(up(s, d_bvars, bv32));
            SX_TRY(up(s, d_dest, order));
            peak_bytes = std::max(peak_bytes, sizeof(double) * (static_cast<size_t>(mp) * (EPOCH + 1) + static_cast<size_t>(Sld) * nb));
            for (int64_t c0 = 0; c0 < nb; c0 += CH) {
                const int64_t kc = std::min<int64_t>(CH, nb - c0);
                SX_HIP(hipMemsetAsync(Wc, 0, sizeof(double) * static_cast<size_t>(mp) * kc, s));
                hipLaunchKernelGGL(k_tb_scatter_cols, dim3(gridof(kc)), dim3(TB_WG), 0, s, kc, d_bvars + c0, n, A->csc_ptr, A->csc_idx, A->csc_val, d_eqidx, Wc, mp);
                SX_TRY(ops.ftran(Wc, kc, true, true));
                SX_TRY(ops.pack_v(Wc, kc, order.data() + c0)); // (the band parts B11^-1 B12 of these columns, by their windows)
                hipLaunchKernelGGL(k_tb_copy_cols, dim3(static_cast<unsigned>(std::min<int64_t>(gridof(nb), 64)), static_cast<unsigned>(kc)), dim3(TB_WG), 0, s, nb, Wc + m1, mp,
                                   d_dest + c0, Sa, Sld);
            }
            SX_HIP(hipStreamSynchronize(s));
            if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: Schur complement (%lld rows: %lld dense, %lld placeholders) assembled at %.1f ms\n", epochs, (long long)nb, (long long)ndr, (long long)h, now() - t_begin);
            std::vector<int32_t> rep2(static_cast<size_t>(nb)), perm2(static_cast<size_t>(nb));
            SX_TRY(sx_denselu_factor_dev(dl, guessed ? tol_schur_guess : tol_schur_keep, &nrep2, rep2.data(), perm2.data()));
            // a border column without a pivot leaves the basis (it stays in the tableau); what takes its place is the unit
            // vector of the border row on its diagonal: a dense row's logical -- or, for a placeholder's row, the
            // placeholder turns into its row's real logical (and the border column into its mirror image, a dummy)
            // (all of them leave first: the logical a replaced column stands for may itself be a later border column that
            //  the same elimination set aside)
            for (int64_t j = 0; j < nb; ++j)
                if (rep2[j]) {
                    is_basic[bcol[j]] = 0;
                    tracked.push_back(static_cast<int32_t>(bcol[j]));
                }
            for (int64_t j = 0; j < nb; ++j) {
                if (!rep2[j]) continue;
                const int64_t r = perm2[j];
                int64_t w;
                if (r < ndr) {
                    w = n + rows_dense[r];
                    bcol[j] = w;
                } else {
                    const int64_t p = ph_pos[static_cast<size_t>(r - ndr)];
                    w = n + ph_row[p];
                    bvar[p] = w;
                    bcol[j] = -1;
                }
                if (is_basic[w]) {
                    sx_set_error("internal: the repair of the Schur complement wants logical %lld twice", (long long)(w - n));
                    return SX_ERR_UNSUPPORTED;
                }
                is_basic[w] = 1;
                untrack(w);
            }
            SX_TRY(ops.finish_v(rep2)); // (a column that left the basis: its V column is zero, like the unit vector's that replaced it)
            if (trace) {
                std::vector<double> dg(static_cast<size_t>(nb));
                SX_HIP(hipMemcpy2D(dg.data(), sizeof(double), Sa, sizeof(double) * (Sld + 1), sizeof(double), static_cast<size_t>(nb), hipMemcpyDeviceToHost));
                double mn = INFINITY;
                int64_t c2 = 0, c4 = 0, c6 = 0;
                for (double d : dg) {
                    const double a = std::fabs(d);
                    mn = std::min(mn, a);
                    c2 += a < 1e-2;
                    c4 += a < 1e-4;
                    c6 += a < 1e-6;
                }
                fprintf(stderr, "[sx_crossover_band] epoch %d: Schur complement factored at %.1f ms (%lld columns set aside; smallest pivot %.2e, %lld / %lld / %lld below 1e-2 / 1e-4 / 1e-6); "
                                "B11^-1 B12 %s: %.1f rows per border column, %.3f GB\n",
                        epochs, now() - t_begin, (long long)nrep2, mn, (long long)c2, (long long)c4, (long long)c6, ops.v_ready ? "packed by windows" : "not kept (solve form)",
                        static_cast<double>(ops.v_used) / static_cast<double>(nb), static_cast<double>(ops.v_used) * 8e-9);
            }
        }
        ops.dl = dl;
        // ---- B12: the border columns' entries in the band rows
        if (nb > 0 && m1 > 0) {
            std::vector<int32_t> ep, ej;
            std::vector<double> ev;
            for (int64_t j = 0; j < nb; ++j) {
                const int64_t v = bcol[j];
                if (v < 0) continue;
                if (v >= n) {
                    if (rowb[v - n] >= 0) {
                        ep.push_back(rowb[v - n]);
                        ej.push_back(static_cast<int32_t>(j));
                        ev.push_back(1.0);
                    }
                    continue;
                }
                for (int64_t k = cptr[v]; k < cptr[v + 1]; ++k)
                    if (rowb[cidx[k]] >= 0) {
                        ep.push_back(rowb[cidx[k]]);
                        ej.push_back(static_cast<int32_t>(j));
                        ev.push_back(cval[k]);
                    }
            }
            rows_from_triplets(m1, ep, ej, ev, true, b12r);
            rows_from_triplets(nb, ej, ep, ev, false, b12c);
            SX_TRY(rows_to_dev(b12r, static_cast<int64_t>(b12r.list.size()), true, ops.b12_rows));
            SX_TRY(rows_to_dev(b12c, nb, false, ops.b12_cols));
            ops.work_cols = ops.v_ready ? 1 : std::max<int64_t>(1, std::min<int64_t>(128, static_cast<int64_t>(1.0e9 / (8.0 * static_cast<double>(m1)))));
            SX_TRY(edev.get(static_cast<size_t>(m1) * ops.work_cols, &ops.work));
        }
        if (trace) {
            SX_HIP(hipStreamSynchronize(s));
            fprintf(stderr, "[sx_crossover_band] epoch %d: border blocks in place at %.1f ms\n", epochs, now() - t_begin);
        }
        // ---- who is where
        std::vector<int64_t> head(static_cast<size_t>(mp), -1);
        for (int64_t p = 0; p < m1; ++p) head[p] = bvar[p];
        for (int64_t j = 0; j < nb; ++j) head[m1 + j] = bcol[j];
        std::fill(vstat.begin(), vstat.end(), 0);
        int64_t n_basic = 0, n_dummy = 0;
        for (int64_t p = 0; p < mp; ++p) {
            if (head[p] < 0) {
                ++n_dummy;
                continue;
            }
            if (vstat[head[p]] == 1) {
                sx_set_error("internal: variable %lld covers two positions of the basis", (long long)head[p]);
                return SX_ERR_UNSUPPORTED;
            }
            vstat[head[p]] = 1;
            ++n_basic;
        }
        if (n_basic != m) {
            sx_set_error("internal: %lld basic variables for %lld rows", (long long)n_basic, (long long)m);
            return SX_ERR_UNSUPPORTED;
        }
        for (int64_t v = 0; v < NV; ++v) is_basic[v] = vstat[v] == 1;
        std::vector<int32_t> varJ;
        for (int32_t v : tracked)
            if (vstat[v] == 0) {
                vstat[v] = 2;
                varJ.push_back(v);
            }
        std::sort(varJ.begin(), varJ.end()); // (neighbours in the order of the variables share the panels of a sparse band solve)
        int64_t nJ = static_cast<int64_t>(varJ.size());
        // ---------------------------------------------------------------- this epoch's blocks: positions, tableau, eta file
        const int nblk = static_cast<int>(gridof(mp));
        double *d_xB = nullptr, *d_lB = nullptr, *d_uB = nullptr, *d_cB = nullptr, *d_g = nullptr, *d_vec = nullptr, *d_part = nullptr, *d_infpart = nullptr, *d_etp = nullptr;
        int32_t *d_head = nullptr, *d_blist = nullptr, *d_infoff = nullptr, *d_inflist = nullptr;
        TbPart *d_rpart = nullptr;
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_xB));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_lB));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_uB));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_cB));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_g));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_vec));
        SX_TRY(edev.get(static_cast<size_t>(mp), &d_head));
        SX_TRY(edev.get(static_cast<size_t>(2 * nblk), &d_part));
        SX_TRY(edev.get(static_cast<size_t>(TB_EB) * TB_ETS, &d_etp));
        SX_TRY(edev.get(static_cast<size_t>(nblk), &d_blist));
        SX_TRY(edev.get(static_cast<size_t>(2 * nblk), &d_infpart));
        SX_TRY(edev.get(static_cast<size_t>(nblk), &d_infoff));
        SX_TRY(edev.get(static_cast<size_t>(TB_INFLIST), &d_inflist));
        SX_TRY(edev.get(static_cast<size_t>(nblk), &d_rpart));
        TbSlots sl;
        {
            size_t free_b = 0, total_b = 0;
            SX_HIP(hipMemGetInfo(&free_b, &total_b));
            int64_t cap0 = nJ + std::max<int64_t>(512, nJ / 2); // (growing it later is an allocation, a copy and a free of GBs)
            if (static_cast<double>(TbSlots::bytes_for(mp, cap0)) > 0.6 * static_cast<double>(free_b)) cap0 = nJ + 16;
            if (static_cast<double>(TbSlots::bytes_for(mp, cap0)) > 0.8 * static_cast<double>(free_b)) {
                sx_set_error("the tableau of %lld tracked columns over %lld positions does not fit the free device memory", (long long)cap0, (long long)mp);
                return SX_ERR_NOMEM;
            }
            SX_TRY(sl.reserve(s, mp, cap0, 0, &arena));
            peak_bytes = std::max(peak_bytes, TbSlots::bytes_for(mp, cap0) + sizeof(double) * static_cast<size_t>(mp) * (EPOCH + 1));
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: tableau and position blocks allocated at %.1f ms\n", epochs, now() - t_begin);
        // ---------------------------------------------------------------- values: non-basic and tracked columns, right-hand side
        // logical of row i: s_i = b_i - (A x)_i for the tracked ones (their current value), 0 for the non-basic ones
        SX_TRY(up(s, d_tmpn, hx));
        SX_TRY(sx_score_rows_dev(ctx, A, d_tmpn, b, nullptr, 0.0, d_tmpm, nullptr));
        SX_TRY(down(s, hslack, d_tmpm, static_cast<size_t>(m)));
        SX_HIP(hipStreamSynchronize(s));
        std::vector<double> xN(hx);
        for (int64_t j = 0; j < n; ++j) {
            if (vstat[j] == 1) xN[j] = 0.0;
            else if (vstat[j] == 0) {
                const bool upb = (hu[j] - hx[j]) < (hx[j] - hl[j]);
                double v = upb ? hu[j] : hl[j];
                if (std::isinf(v)) v = 0.0;
                xN[j] = v;
                hx[j] = v;
                atup[j] = upb;
            }
        }
        for (int64_t i = 0; i < m; ++i) {
            xlog[i] = 0.0;
            if (vstat[n + i] == 2) xlog[i] = hlt[i] ? std::max(hslack[i], 0.0) : 0.0;
        }
        SX_TRY(up(s, d_tmpn, xN));
        SX_TRY(sx_score_rows_dev(ctx, A, d_tmpn, b, nullptr, 0.0, d_tmpm, nullptr));
        std::vector<double> hr;
        SX_TRY(down(s, hr, d_tmpm, static_cast<size_t>(m)));
        SX_HIP(hipStreamSynchronize(s));
        std::vector<double> rp(static_cast<size_t>(mp), 0.0); // (a placeholder's border row: 0)
        for (int64_t i = 0; i < m; ++i) rp[eqidx[i]] = hr[i] - xlog[i];
        // solve helper: row-space columns through the bordered factors, then the eta file
        auto ftran_cols = [&](double *W, int64_t ncols, int64_t n_eta_now, bool lp_columns) -> int {
            if (ncols == 0) return SX_OK;
            SX_TRY(ops.ftran(W, ncols, lp_columns));
            if (n_eta_now > 0) { // (what the tableau would drop anyway goes first: the list of an eta holds real entries only)
                hipLaunchKernelGGL(k_tb_drop, dim3(gridcap(mp * ncols)), dim3(TB_WG), 0, s, mp * ncols, W, TB_DROP);
            }
            for (int64_t k0 = 0; k0 < n_eta_now; k0 += TB_EB) {
                const int K = static_cast<int>(std::min<int64_t>(TB_EB, n_eta_now - k0));
                hipLaunchKernelGGL(k_tb_etab_gather, dim3(gridof(std::max<int64_t>(TB_EB * ncols, TB_EB * TB_EB))), dim3(TB_WG), 0, s, mp, ncols, W, d_eta,
                                   d_eta_r, k0, K, sl.ebG, d_ebM);
                hipLaunchKernelGGL(k_tb_etab_solve, dim3(static_cast<unsigned>((ncols + 63) / 64)), dim3(64), 0, s, ncols, sl.ebG, d_ebM, sl.ebS, sl.elist);
                hipLaunchKernelGGL(k_tb_etab_apply, dim3(static_cast<unsigned>(std::min(nblk, 1024)), static_cast<unsigned>((ncols + TB_EC - 1) / TB_EC)),
                                   dim3(TB_WG), 0, s, mp, ncols, W, d_eta, d_eta_r, k0, K, sl.ebS, sl.elist);
            }
            SX_HIP(hipGetLastError());
            return SX_OK;
        };
        // ---- basic variables (a dummy -- placeholder or mirror column -- is free, costs nothing and never leaves)
        {
            std::vector<double> hlB(static_cast<size_t>(mp)), huB(static_cast<size_t>(mp)), hcB(static_cast<size_t>(mp));
            std::vector<int32_t> hhead(static_cast<size_t>(mp));
            for (int64_t p = 0; p < mp; ++p) {
                hhead[p] = static_cast<int32_t>(head[p]);
                if (head[p] < 0) {
                    hlB[p] = -INFINITY;
                    huB[p] = INFINITY;
                    hcB[p] = 0.0;
                    continue;
                }
                hlB[p] = var_lo(head[p]);
                huB[p] = var_up(head[p]);
                hcB[p] = var_cost(head[p]);
            }
            SX_TRY(up(s, d_head, hhead));
            SX_TRY(up(s, d_lB, hlB));
            SX_TRY(up(s, d_uB, huB));
            SX_TRY(up(s, d_cB, hcB));
            SX_TRY(up(s, d_xB, rp));
            SX_HIP(hipStreamSynchronize(s));
        }
        if (trace) fprintf(stderr, "[sx_crossover_band] epoch %d: right-hand side and basic variables' arrays at %.1f ms\n", epochs, now() - t_begin);
        SX_TRY(ftran_cols(d_xB, 1, 0, false));
        if (vbasis_in && epochs == 1 && !strict_tried) {
            // a GIVEN basis is kept as it is (tolerances of a basis the simplex has reached) -- unless its basic solution says it
            // is no basis to start from: members that are all but dependent put it far outside the bounds, and phase 1 would
            // pay for each of them pivot by pivot (15 wrong members of 1,200: 6,537 iterations against 402 from the point alone).
            // Then the same set is factored once more with the tolerances of a guess
            std::vector<double> t;
            SX_TRY(down(s, t, d_xB, static_cast<size_t>(mp)));
            SX_HIP(hipStreamSynchronize(s));
            double worst = 0.0;
            int64_t ninf = 0;
            for (int64_t p = 0; p < mp; ++p) {
                if (head[p] < 0) continue;
                const double vi = std::max(var_lo(head[p]) - t[p], t[p] - var_up(head[p]));
                if (vi > feas_tol) ++ninf;
                worst = std::max(worst, vi);
            }
            strict_tried = true;
            if (worst > 1.0 || ninf > std::max<int64_t>(16, m / 50)) {
                if (trace) fprintf(stderr, "[sx_crossover_band] the given basis puts %lld variables outside their bounds (worst %.3e): factored again with the tolerances of a guess\n", (long long)ninf, worst);
                strict_next = true;
                continue;
            }
        }
        if (trace) { // how far the basic solution is from the point handed over (conditioning of the guessed basis)
            std::vector<double> t;
            SX_TRY(down(s, t, d_xB, static_cast<size_t>(mp)));
            SX_HIP(hipStreamSynchronize(s));
            double devi = 0.0, worst = 0.0, dummy = 0.0;
            int64_t ninf = 0, inf_bs = 0, inf_bl = 0, inf_border = 0, border_struct = 0;
            for (int64_t j = 0; j < nb; ++j) border_struct += bcol[j] >= 0 && bcol[j] < n;
            for (int64_t p = 0; p < mp; ++p) {
                const int64_t v = head[p];
                if (v < 0) {
                    if (p < m1) dummy = std::max(dummy, std::fabs(t[p]));
                    continue;
                }
                const double ref = v < n ? hx[v] : (hlt[v - n] ? std::max(hslack[v - n], 0.0) : 0.0);
                devi = std::max(devi, std::fabs(t[p] - ref));
                const double vi = std::max(var_lo(v) - t[p], t[p] - var_up(v));
                if (vi > feas_tol) {
                    ++ninf;
                    if (p >= m1) ++inf_border;
                    else if (v < n) ++inf_bs;
                    else ++inf_bl;
                }
                worst = std::max(worst, vi);
            }
            fprintf(stderr, "[sx_crossover_band] epoch %d: outside their bounds: %lld band columns, %lld band logicals, %lld border variables (the border holds %lld columns, %lld logicals)\n",
                    epochs, (long long)inf_bs, (long long)inf_bl, (long long)inf_border, (long long)border_struct, (long long)(nb - border_struct));
            fprintf(stderr, "[sx_crossover_band] epoch %d: basic solution deviates from the point by at most %.3e; %lld basic variables outside their bounds (worst %.3e); "
                            "placeholders hold at most %.1e; %.1f ms\n", epochs, devi, (long long)ninf, worst, dummy, now() - t_begin);
        }
        // ---- tracked columns
        auto value_of = [&](int64_t v) { return v < n ? hx[v] : xlog[v - n]; };
        auto load_slots = [&](int64_t s0, const std::vector<int32_t> &vars) -> int {
            const size_t k = vars.size();
            if (k == 0) return SX_OK;
            std::vector<double> vx(k), vl(k), vu(k), vc(k);
            std::vector<int32_t> vs(k);
            for (size_t t = 0; t < k; ++t) {
                const int64_t v = vars[t];
                vl[t] = var_lo(v);
                vu[t] = var_up(v);
                vc[t] = var_cost(v);
                vx[t] = std::min(std::max(value_of(v), vl[t]), vu[t]);
                if (vx[t] > vl[t] && vx[t] < vu[t]) vs[t] = TB_SUP;
                else vs[t] = (vx[t] >= vu[t] && vu[t] > vl[t]) ? TB_UPP : TB_LOW;
            }
            SX_HIP(hipMemcpyAsync(sl.varJ + s0, vars.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipMemcpyAsync(sl.xJ + s0, vx.data(), sizeof(double) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipMemcpyAsync(sl.lJ + s0, vl.data(), sizeof(double) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipMemcpyAsync(sl.uJ + s0, vu.data(), sizeof(double) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipMemcpyAsync(sl.cJ + s0, vc.data(), sizeof(double) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipMemcpyAsync(sl.statJ + s0, vs.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, s));
            SX_HIP(hipStreamSynchronize(s)); // (the staging vectors go out of scope)
            return SX_OK;
        };
        auto build_cols = [&](int64_t s0, int64_t k, int64_t n_eta_now) -> int {
            if (k == 0) return SX_OK;
            double *W = sl.T + static_cast<size_t>(s0) * mp;
            SX_HIP(hipMemsetAsync(W, 0, sizeof(double) * static_cast<size_t>(mp) * k, s));
            hipLaunchKernelGGL(k_tb_scatter_cols, dim3(gridof(k)), dim3(TB_WG), 0, s, k, sl.varJ + s0, n, A->csc_ptr, A->csc_idx, A->csc_val, d_eqidx, W, mp);
            SX_TRY(ftran_cols(W, k, n_eta_now, true));
            hipLaunchKernelGGL(k_tb_drop, dim3(gridcap(mp * k)), dim3(TB_WG), 0, s, mp * k, W, 10.0 * TB_DROP);
            // reduced costs of the new columns under the current basis: d = c_J - c_B^T T
            hipLaunchKernelGGL(k_tb_coldot, dim3(static_cast<unsigned>(k)), dim3(TB_WG), 0, s, mp, W, d_cB, sl.cJ + s0, sl.dJ + s0);
            SX_HIP(hipGetLastError());
            return SX_OK;
        };
        SX_TRY(load_slots(0, varJ));
        SX_TRY(build_cols(0, nJ, 0));
        if (trace) {
            SX_HIP(hipStreamSynchronize(s));
            fprintf(stderr, "[sx_crossover_band] epoch %d: tableau columns (FTRAN of %lld) done at %.1f ms\n", epochs, (long long)nJ, now() - t_begin);
        }
        TbState hst{};
        hst.max_iter = max_iter - tot_iters;
        hst.cap_eta = EPOCH;
        hst.feas_tol = feas_tol;
        hst.opt_tol = opt_tol;
        SX_HIP(hipMemcpyAsync(d_st, &hst, sizeof(hst), hipMemcpyHostToDevice, s));
        SX_HIP(hipStreamSynchronize(s));
        if (trace)
            fprintf(stderr, "[sx_crossover_band] epoch %d: %lld positions (%lld band + %lld border, %lld dummies), %lld tracked columns (tableau capacity %lld = %.2f GB, "
                            "eta file %lld = %.2f GB); %.1f ms so far\n", epochs, (long long)mp, (long long)m1, (long long)nb, (long long)n_dummy, (long long)nJ,
                    (long long)sl.cap, static_cast<double>(mp) * sl.cap * 8e-9, (long long)EPOCH, static_cast<double>(mp) * (EPOCH + 1) * 8e-9, now() - t_begin);
        // ---------------------------------------------------------------- the simplex on the tracked columns
        auto one_pivot = [&]() {
            const TbPend pend{sl.vbuf, d_pr, sl.s0, sl.sbase, sl.cap};
            hipLaunchKernelGGL(k_tb_infeas, dim3(nblk), dim3(TB_WG), 0, s, mp, d_xB, d_lB, d_uB, d_st, d_g, d_infpart);
            hipLaunchKernelGGL(k_tb_phase, dim3(1), dim3(TB_WG), 0, s, nblk, d_infpart, d_st, d_infoff);
            hipLaunchKernelGGL(k_tb_inflist, dim3(nblk), dim3(TB_WG), 0, s, mp, d_g, d_infpart, d_infoff, d_st, d_inflist);
            hipLaunchKernelGGL(k_tb_gu, dim3(TB_K), dim3(TB_WG), 0, s, mp, nblk, d_eta, d_g, d_inflist, d_st, pend, d_gu);
            hipLaunchKernelGGL(k_tb_price1, dim3(static_cast<unsigned>(nJ)), dim3(TB_WG), 0, s, mp, nblk, sl.T, d_g, d_inflist, d_st, pend, d_gu, sl.d1);
            hipLaunchKernelGGL(k_tb_select, dim3(1), dim3(TB_WG), 0, s, nJ, sl.dJ, sl.d1, sl.statJ, sl.xJ, sl.lJ, sl.uJ, d_st);
            hipLaunchKernelGGL(k_tb_ratio1, dim3(nblk), dim3(TB_WG), 0, s, mp, sl.T, d_xB, d_lB, d_uB, d_st, d_eta, d_part, pend);
            hipLaunchKernelGGL(k_tb_tmax, dim3(1), dim3(TB_WG), 0, s, nblk, d_part, d_st, d_blist);
            hipLaunchKernelGGL(k_tb_ratio, dim3(nblk), dim3(TB_WG), 0, s, mp, d_xB, d_lB, d_uB, d_st, d_eta, d_rpart);
            hipLaunchKernelGGL(k_tb_decide, dim3(1), dim3(TB_WG), 0, s, nblk, d_rpart, sl.xJ, sl.lJ, sl.uJ, sl.statJ, d_st);
            hipLaunchKernelGGL(k_tb_rowcopy, dim3(gridof(nJ)), dim3(TB_WG), 0, s, mp, nJ, sl.T, d_eta, d_st, pend, sl.rowbuf);
            hipLaunchKernelGGL(k_tb_update, dim3(static_cast<unsigned>(std::min(nblk, 64))), dim3(TB_WG), 0, s, mp, d_xB, d_eta, d_st, d_blist);
            hipLaunchKernelGGL(k_tb_post, dim3(1), dim3(TB_WG), 0, s, nJ, sl.dJ, sl.rowbuf, d_head, d_xB, d_lB, d_uB, d_cB, sl.varJ, sl.xJ, sl.lJ, sl.uJ,
                               sl.cJ, sl.statJ, d_eta_r, d_st, sl.vbuf, sl.cap, d_pr, sl.s0, sl.sbase);
        };
        auto fold = [&]() { // applies the pending batch when it is nearly full, or when the run has stopped
            const TbPend pend{sl.vbuf, d_pr, sl.s0, sl.sbase, sl.cap};
            hipLaunchKernelGGL(k_tb_fold_begin, dim3(1), dim3(1), 0, s, d_st, 16);
            hipLaunchKernelGGL(k_tb_fold, dim3(static_cast<unsigned>(std::min(nblk, 128)), static_cast<unsigned>((nJ + TB_FJ - 1) / TB_FJ)), dim3(TB_WG), 0, s,
                               mp, nJ, sl.T, d_eta, d_st, pend);
            hipLaunchKernelGGL(k_tb_fold_end, dim3(gridof(nJ)), dim3(TB_WG), 0, s, nJ, d_st, sl.s0, sl.sbase);
            hipLaunchKernelGGL(k_tb_fold_done, dim3(1), dim3(1), 0, s, d_st);
        };
        int rounds = 0;
        bool restart = false;
        std::vector<double> hrc; // reduced costs of the structural columns under the last duals
        for (;;) {
            ++rounds;
            if (nJ > 0) {
                for (;;) {
                    for (int k = 0; k < 16; ++k) one_pivot();
                    fold();
                    SX_HIP(hipMemcpyAsync(&hst, d_st, sizeof(hst), hipMemcpyDeviceToHost, s));
                    SX_HIP(hipStreamSynchronize(s));
                    SX_HIP(hipGetLastError());
                    if (hst.status != 0) break;
                }
            } else { // nothing tracked: only the state of the basic variables decides the phase
                hipLaunchKernelGGL(k_tb_infeas, dim3(nblk), dim3(TB_WG), 0, s, mp, d_xB, d_lB, d_uB, d_st, d_g, d_infpart);
                hipLaunchKernelGGL(k_tb_phase, dim3(1), dim3(TB_WG), 0, s, nblk, d_infpart, d_st, d_infoff);
                SX_HIP(hipMemcpyAsync(&hst, d_st, sizeof(hst), hipMemcpyDeviceToHost, s));
                SX_HIP(hipStreamSynchronize(s));
                hst.status = 1;
            }
            if (trace)
                fprintf(stderr, "[sx_crossover_band]   round %d: status %d after %lld iterations (%lld pivots, %lld flips, %lld of no length), phase %d, %d "
                                "infeasible (sum %.3e), %.1f ms so far\n", rounds, hst.status, hst.iters, hst.pivots, hst.flips, hst.degen, hst.phase,
                        hst.n_inf, hst.sum_inf, now() - t_begin);
            if (hst.status == 3 && hst.n_eta >= hst.cap_eta && tot_iters + hst.iters < max_iter) {
                restart = true; // the eta file is full: factor the basis afresh
                break;
            }
            if (hst.status != 1) break;
            // ---- no tracked column prices out: duals, then every other column.  In phase 1 the cost is the
            //      infeasibility's (g on the basic variables, nothing elsewhere)
            const bool ph1 = hst.phase == 1;
            SX_HIP(hipMemcpyAsync(d_vec, ph1 ? d_g : d_cB, sizeof(double) * static_cast<size_t>(mp), hipMemcpyDeviceToDevice, s));
            const int ets = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(TB_ETS, mp / 8192)));
            for (int64_t hi = hst.n_eta; hi > 0; hi -= TB_EB) { // blocks of TB_EB etas, the youngest first
                const int64_t k0 = std::max<int64_t>(0, hi - TB_EB);
                const int K = static_cast<int>(hi - k0);
                hipLaunchKernelGGL(k_tb_etat_dots, dim3(static_cast<unsigned>(ets), static_cast<unsigned>(K)), dim3(TB_WG), 0, s, mp, d_vec, d_eta, d_eta_r, k0, d_etp);
                hipLaunchKernelGGL(k_tb_etat_solve, dim3(1), dim3(64), 0, s, mp, ets, K, d_vec, d_eta, d_eta_r, k0, d_etp);
            }
            SX_TRY(ops.btran(d_vec)); // B_aug^T y = v: position space in, row space out
            std::vector<double> yeq;
            SX_TRY(down(s, yeq, d_vec, static_cast<size_t>(mp)));
            SX_HIP(hipStreamSynchronize(s));
            if (trace) fprintf(stderr, "[sx_crossover_band]   round %d: duals (eta file of %lld, transposed solves) by %.1f ms\n", rounds, (long long)hst.n_eta, now() - t_begin);
            for (int64_t i = 0; i < m; ++i) hy[i] = yeq[eqidx[i]];
            // reduced costs of all structural columns with these duals (the K1 walk): rc = c' - A^T y
            SX_HIP(hipMemcpyAsync(d_tmpm, hy.data(), sizeof(double) * static_cast<size_t>(m), hipMemcpyHostToDevice, s));
            const double *cost = c;
            if (ph1) {
                SX_HIP(hipMemsetAsync(d_tmpn, 0, sizeof(double) * static_cast<size_t>(n), s));
                cost = d_tmpn;
            }
            SX_TRY(sx_score_columns_dev(ctx, A, d_tmpm, cost, nullptr, nullptr, nullptr, 0.0, d_rc, nullptr));
            SX_TRY(down(s, hrc, d_rc, static_cast<size_t>(n)));
            // who is where now
            std::vector<int32_t> hh, hvarJ, hstatJ;
            SX_TRY(down(s, hh, d_head, static_cast<size_t>(mp)));
            SX_TRY(down(s, hvarJ, sl.varJ, static_cast<size_t>(nJ)));
            SX_TRY(down(s, hstatJ, sl.statJ, static_cast<size_t>(nJ)));
            SX_HIP(hipStreamSynchronize(s));
            if (trace) fprintf(stderr, "[sx_crossover_band]   round %d: reduced costs by %.1f ms\n", rounds, now() - t_begin);
            std::fill(vstat.begin(), vstat.end(), 0);
            for (int64_t p = 0; p < mp; ++p)
                if (hh[p] >= 0) vstat[hh[p]] = 1;
            for (int64_t t = 0; t < nJ; ++t) vstat[hvarJ[t]] = 2;
            std::vector<std::pair<double, int32_t>> viol;
            for (int64_t j = 0; j < n; ++j) {
                if (vstat[j] != 0 || hl[j] == hu[j]) continue;
                const double d = hrc[j];
                if (atup[j] ? d > opt_tol : d < -opt_tol) viol.emplace_back(std::fabs(d), static_cast<int32_t>(j));
            }
            for (int64_t i = 0; i < m; ++i) { // non-basic logicals sit at 0: reduced cost -y_i, only '<' rows may move (up)
                if (vstat[n + i] != 0 || !hlt[i]) continue;
                const double d = -hy[i];
                if (d < -opt_tol) viol.emplace_back(std::fabs(d), static_cast<int32_t>(n + i));
            }
            if (trace) fprintf(stderr, "[sx_crossover_band]   round %d (%s): %zu columns outside the tableau price out\n", rounds, ph1 ? "phase 1" : "phase 2", viol.size());
            if (viol.empty()) {
                if (ph1) hst.status = 101; // infeasible: no column anywhere reduces the infeasibility
                break;
            }
            std::sort(viol.begin(), viol.end(), [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &bb) { return a.first > bb.first || (a.first == bb.first && a.second < bb.second); });
            const int64_t take = std::min<int64_t>(static_cast<int64_t>(viol.size()), 2048);
            if (nJ + take > sl.cap) { // (nothing is pending here: the run stopped, its batch was folded)
                const int rc_ = sl.reserve(s, mp, nJ + take + std::max<int64_t>(512, nJ / 2), nJ);
                if (rc_ != SX_OK) return rc_;
                peak_bytes = std::max(peak_bytes, 2 * TbSlots::bytes_for(mp, nJ) + sizeof(double) * static_cast<size_t>(mp) * (EPOCH + 1));
            }
            std::vector<int32_t> add(static_cast<size_t>(take));
            for (int64_t t = 0; t < take; ++t) add[t] = viol[t].second;
            std::sort(add.begin(), add.end());
            for (int32_t v : add)
                if (v >= n) xlog[v - n] = 0.0;
            SX_TRY(load_slots(nJ, add));
            SX_TRY(build_cols(nJ, take, hst.n_eta));
            nJ += take;
            added_total += take;
            hst.status = 0;
            SX_HIP(hipMemcpyAsync(d_st, &hst, sizeof(hst), hipMemcpyHostToDevice, s)); // (status only changed; counters as read)
            SX_HIP(hipStreamSynchronize(s));
            if (trace) fprintf(stderr, "[sx_crossover_band]   round %d: %lld columns brought into the tableau by %.1f ms\n", rounds, (long long)take, now() - t_begin);
        }
        // ---------------------------------------------------------------- the epoch's end state -> host
        std::vector<int32_t> hh, hvarJ, hstatJ;
        std::vector<double> hxb, hxJ;
        SX_TRY(down(s, hh, d_head, static_cast<size_t>(mp)));
        SX_TRY(down(s, hxb, d_xB, static_cast<size_t>(mp)));
        SX_TRY(down(s, hvarJ, sl.varJ, static_cast<size_t>(nJ)));
        SX_TRY(down(s, hstatJ, sl.statJ, static_cast<size_t>(nJ)));
        SX_TRY(down(s, hxJ, sl.xJ, static_cast<size_t>(nJ)));
        SX_HIP(hipStreamSynchronize(s));
        tot_iters += hst.iters;
        tot_pivots += hst.pivots;
        tot_flips += hst.flips;
        tot_degen += hst.degen;
        std::fill(is_basic.begin(), is_basic.end(), 0);
        std::fill(vstat.begin(), vstat.end(), 0);
        tracked.clear();
        viol_max = 0.0;
        prev_pos.assign(static_cast<size_t>(m1_all), -1);
        for (int64_t p = 0; p < mp; ++p) {
            const int64_t v = hh[p];
            if (v < 0) continue;
            is_basic[v] = 1;
            vstat[v] = 1;
            viol_max = std::max(viol_max, std::max(var_lo(v) - hxb[p], hxb[p] - var_up(v)));
            if (v < n) hx[v] = hxb[p];
            if (p < m1 && v == head[p] && old_of_new[p] >= 0) prev_pos[old_of_new[p]] = v;
        }
        for (int64_t t = 0; t < nJ; ++t) {
            const int64_t v = hvarJ[t];
            tracked.push_back(static_cast<int32_t>(v));
            vstat[v] = hstatJ[t] == TB_SUP ? 3 : 2;
            if (v < n) {
                hx[v] = hxJ[t];
                atup[v] = hstatJ[t] == TB_UPP;
            }
        }
        final_status = hst.status;
        if (restart) continue;
        if ((hst.status == 4 || hst.status == 2) && bad_epochs < 2 && tot_iters < max_iter) { // a tiny pivot / a ray that should not be: fresh factors first
            ++bad_epochs;
            continue;
        }
        if (hst.status == 1) {
            // ---- the vertex against A itself before it is called optimal: x_B was only ever updated, y came through the eta
            //      file -- row residuals of x (one K2 walk) and the reduced costs of the basic columns (they are in hrc).
            //      Beyond the tolerances: factor the current basis afresh and go on from there (twice at most)
            SX_TRY(up(s, d_tmpn, hx));
            SX_TRY(sx_score_rows_dev(ctx, A, d_tmpn, b, nullptr, 0.0, d_tmpm, nullptr));
            SX_TRY(down(s, hslack, d_tmpm, static_cast<size_t>(m)));
            SX_HIP(hipStreamSynchronize(s));
            auto row_residual = [&]() {
                double worst = 0.0;
                for (int64_t i = 0; i < m; ++i) {
                    const double sc = 1.0 + std::fabs(hb[i]);
                    const double r = hlt[i] ? std::max(-hslack[i], 0.0) : std::fabs(hslack[i]);
                    worst = std::max(worst, r / sc);
                }
                return worst;
            };
            resid_max = row_residual();
            if (resid_max > 1e-11) {
                // one step of iterative refinement with the factors at hand: B delta = b - A x - s (s: the logicals'
                // values), x_B += delta -- what thousands of updates x_B -= theta alpha left behind goes at once
                std::vector<double> slog(static_cast<size_t>(m), 0.0), rr(static_cast<size_t>(mp), 0.0), delta;
                for (int64_t p = 0; p < mp; ++p)
                    if (hh[p] >= n) slog[hh[p] - n] = hxb[p];
                for (int64_t t = 0; t < nJ; ++t)
                    if (hvarJ[t] >= n) slog[hvarJ[t] - n] = hxJ[t];
                for (int64_t i = 0; i < m; ++i) rr[eqidx[i]] = hslack[i] - slog[i];
                SX_TRY(up(s, d_vec, rr));
                SX_TRY(ftran_cols(d_vec, 1, hst.n_eta, false));
                SX_TRY(down(s, delta, d_vec, static_cast<size_t>(mp)));
                SX_HIP(hipStreamSynchronize(s));
                viol_max = 0.0;
                for (int64_t p = 0; p < mp; ++p) {
                    const int64_t v = hh[p];
                    if (v < 0) continue;
                    hxb[p] += delta[p];
                    viol_max = std::max(viol_max, std::max(var_lo(v) - hxb[p], hxb[p] - var_up(v)));
                    if (v < n) hx[v] = hxb[p];
                }
                SX_TRY(up(s, d_tmpn, hx));
                SX_TRY(sx_score_rows_dev(ctx, A, d_tmpn, b, nullptr, 0.0, d_tmpm, nullptr));
                SX_TRY(down(s, hslack, d_tmpm, static_cast<size_t>(m)));
                SX_HIP(hipStreamSynchronize(s));
                const double before = resid_max;
                resid_max = row_residual();
                if (trace) fprintf(stderr, "[sx_crossover_band] refinement of x_B: row residual %.2e -> %.2e (relative), bound violation %.2e\n", before, resid_max, viol_max);
            }
            double rc_basic = 0.0;
            if (static_cast<int64_t>(hrc.size()) == n)
                for (int64_t j = 0; j < n; ++j)
                    if (vstat[j] == 1) rc_basic = std::max(rc_basic, std::fabs(hrc[j]) / (1.0 + std::fabs(hc[j])));
            if (trace) fprintf(stderr, "[sx_crossover_band] check of the vertex: row residual %.2e (relative), reduced costs of basic columns %.2e\n", resid_max, rc_basic);
            if (resid_max > 10.0 * feas_tol || rc_basic > 10.0 * opt_tol || viol_max > 10.0 * feas_tol) {
                if (check_epochs < 2 && tot_iters < max_iter) {
                    ++check_epochs;
                    continue;
                }
                final_status = 4;
            }
        }
        break;
        } // (this epoch's geometry)
    }
    // ------------------------------------------------------------------ outputs
    std::vector<int8_t> vb(static_cast<size_t>(n)), cb(static_cast<size_t>(m), -1);
    for (int64_t j = 0; j < n; ++j) vb[j] = vstat[j] == 1 ? 0 : (vstat[j] == 3 ? -3 : (atup[j] ? -2 : -1));
    for (int64_t i = 0; i < m; ++i)
        if (vstat[n + i] == 1) cb[i] = 0;
    double obj = 0.0;
    for (int64_t j = 0; j < n; ++j) obj += hc[j] * hx[j];
    if (x_out) SX_HIP(hipMemcpyAsync(x_out, hx.data(), sizeof(double) * static_cast<size_t>(n), hipMemcpyHostToDevice, s));
    if (y_out) SX_HIP(hipMemcpyAsync(y_out, hy.data(), sizeof(double) * static_cast<size_t>(m), hipMemcpyHostToDevice, s));
    if (vbasis_out) SX_HIP(hipMemcpyAsync(vbasis_out, vb.data(), static_cast<size_t>(n), hipMemcpyHostToDevice, s));
    if (cbasis_out) SX_HIP(hipMemcpyAsync(cbasis_out, cb.data(), static_cast<size_t>(m), hipMemcpyHostToDevice, s));
    SX_HIP(hipStreamSynchronize(s));
    result->status = final_status == 1 ? 0 : (final_status == 101 ? 1 : final_status);
    result->iters = tot_iters;
    result->phase1_iters = added_total;
    result->warm_start_used = vbasis_in ? 1 : 0;
    result->obj = obj;
    result->max_violation = std::max(viol_max > 0 ? viol_max : 0.0, resid_max);
    if (trace)
        fprintf(stderr, "[sx_crossover_band] done: status %lld, %lld iterations (%lld pivots, %lld flips, %lld of no length) in %d epochs, %lld columns added by "
                        "pricing, objective %.12e, max violation %.2e, largest blocks %.2f GB, %.1f ms\n", (long long)result->status, tot_iters, tot_pivots, tot_flips,
                tot_degen, epochs, (long long)added_total, obj, result->max_violation, static_cast<double>(peak_bytes) * 1e-9, now() - t_begin);
    return SX_OK;
}

} // namespace

SX_API int sx_crossover_band_basis_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                       const double *u, const uint8_t *row_is_lt, const double *x_start, const int8_t *vbasis_in,
                                       const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol, double *x_out,
                                       double *y_out, int8_t *vbasis_out, int8_t *cbasis_out, sx_simplex_result *result) {
    return crossover_band_impl(ctx, A, b, c, l, u, row_is_lt, x_start, vbasis_in, cbasis_in, max_iter, feas_tol, opt_tol, x_out, y_out, vbasis_out,
                               cbasis_out, result, false);
}

// Would the sparse crossover take this LP from this point?  Runs its set-up up to the band-width check -- host copies, row
// order, the basic set guessed from x_start, matching -- and nothing else: SX_OK / SX_ERR_UNSUPPORTED.  A backend asks before
// it sizes its first-order stage (the dense crossover needs four times the iterations in front of it).
SX_API int sx_crossover_band_probe_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                       const double *u, const uint8_t *row_is_lt, const double *x_start) {
    sx_simplex_result r{};
    return crossover_band_impl(ctx, A, b, c, l, u, row_is_lt, x_start, nullptr, nullptr, 0, 0.0, 0.0, nullptr, nullptr, nullptr, nullptr, &r, true);
}

SX_API int sx_crossover_band_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                 const double *u, const uint8_t *row_is_lt, const double *x_start, int64_t max_iter,
                                 double feas_tol, double opt_tol, double *x_out, double *y_out, int8_t *vbasis_out,
                                 int8_t *cbasis_out, sx_simplex_result *result) {
    return sx_crossover_band_basis_dev(ctx, A, b, c, l, u, row_is_lt, x_start, nullptr, nullptr, max_iter, feas_tol, opt_tol, x_out, y_out,
                                       vbasis_out, cbasis_out, result);
}
