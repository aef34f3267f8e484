// Band LU with partial pivoting on the device (kernel group K16f): the factorisation of the starting basis of
// the sparse crossover (sx_crossover_band.hip).  The reference leaves every basis factorisation to Gurobi /
// CPLEX / Mosek (solver_caller/gurobi.py:202-210 model.optimize()); rounds 1-2 kept an explicit dense m x m
// inverse (8 m^2 bytes: 80 GB at 1e5 rows, installed by m pivots in 47 s).  A staged ("staircase") LP basis in
// its natural order is a BAND matrix once its few dense rows are set aside, so it is factored as one:
//
//   storage   LAPACK general-band layout, column major: AB(kl + ku + i - j, j) = A(i, j), ldab = 2 kl + ku + 1
//             (the first kl rows of every column are room for the fill partial pivoting creates);
//   factor    panels of 32 columns.  k_gb_panel: ONE workgroup holds the panel's kl + 32 rows in registers (a row
//             per lane, 32 doubles), per column: arg-max over the lanes -> row swap through LDS -> scale -> rank-one
//             update in registers.  k_gb_trail2: the
//             columns to the right (<= ku + kl + 32 of them), 8 per workgroup in LDS: a 32 x 32 triangle + one
//             combining pass, or -- when the panel swapped rows -- swap and elimination column by column.
//             A column without a usable pivot is REPLACED in place by the unit vector of the row on its diagonal
//             (pivot 1, no multipliers): the caller learns which columns were replaced and treats those rows as
//             covered by their logical variable -- a singular or ill-conditioned guess of a basis is repaired
//             instead of failing the factorisation;
//   solves    ONE launch per solve: right-hand sides 8 per workgroup, every workgroup walks all panels by itself
//             (forward, then backward; both orientations); per panel a 32 x 32 triangle in LDS + one combining
//             pass, or -- for a panel whose factorisation swapped rows -- the column-by-column step.
//
// fp64, FMA contraction off like the rest of the library.  No atomics; a fixed arithmetic order per entry.
#include "sx_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

struct sx_bandlu {
    sx_ctx *ctx = nullptr;
    int64_t n = 0;
    int kl = 0, ku = 0, ldab = 0;
    double *ab = nullptr;
    int32_t *ipiv = nullptr;     // [n] global row swapped with j at step j
    int32_t *replaced = nullptr; // [n] 1: column j was replaced by a unit vector
    int32_t *err = nullptr;      // [1] scatter found an entry outside the band
    uint8_t *d_swaps = nullptr;  // [panels] 1: the panel's factorisation swapped rows
    uint8_t *d_flags = nullptr;  // [groups][panels] of the last sparse solve (grown on demand)
    size_t flags_cap = 0;
    bool factored = false;
    std::vector<uint8_t> panel_swaps; // [panels] 1: the panel's factorisation swapped rows (host copy)
    // partitioned sweeps (k_gbp_*): per sweep kind the responses R of every block to its incoming rows; work buffers
    int part_state = 0; // 0: not tried yet, 1: ready, -1: not available (too few panels, or no memory): sequential sweeps
    int part_LBp = 0, part_P = 0;
    double *part_R[4] = {nullptr, nullptr, nullptr, nullptr};
    double *part_buf = nullptr, *part_delta = nullptr;
    int part_cols = 0; // right-hand sides part_buf / part_delta hold (grown on demand: gb_part_reserve)
};

namespace {

constexpr int GB_NB = 32; // panel width
// Independent diagonal blocks factored side by side (sx_bandlu_factor_blocks_dev): block b holds the columns
// [b stride, b stride + len_b), len_b = rl (rl_last for the last block), followed by identity padding up to the next block
struct GbBlocks {
    int nblk;
    int64_t stride, rl, rl_last;
    __host__ __device__ int64_t base(int b) const { return static_cast<int64_t>(b) * stride; }
    __host__ __device__ int64_t len(int b) const { return b == nblk - 1 ? rl_last : rl; }
};
constexpr int GB_CB = 8;  // target columns per workgroup of the apply / solve kernels
constexpr int GB_T2 = 256;

// -DSX_GB_TICKS: workgroup 0's lane 0 adds the 10 ns ticks between the marks of the solve bodies to g_gb_t[] (printed by
// the solve entry points): where a panel's step spends its time
#ifdef SX_GB_TICKS
__device__ unsigned long long g_gb_t[16];
__device__ unsigned long long g_gb_last;
#define GB_TICK(k)                                                       \
    do {                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                       \
            const unsigned long long now_ = wall_clock64();              \
            g_gb_t[k] += now_ - g_gb_last;                               \
            g_gb_last = now_;                                            \
        }                                                                \
    } while (0)
#else
#define GB_TICK(k) do { } while (0)
#endif

__device__ __forceinline__ double &AB(double *ab, int ldab, int kl, int ku, int64_t i, int64_t j) {
    return ab[static_cast<size_t>(kl + ku + i - j) + static_cast<size_t>(j) * ldab];
}

__global__ __launch_bounds__(256) void k_gb_scatter(int64_t nnz, const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, double *__restrict__ ab, int ldab, int kl,
                                                    int ku, int64_t n, int32_t *__restrict__ err) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (t >= nnz) return;
    const int64_t i = row[t], j = col[t];
    if (i < 0 || j < 0 || i >= n || j >= n || i - j > kl || j - i > ku) {
        *err = 1;
        return;
    }
    AB(ab, ldab, kl, ku, i, j) = val[t]; // (duplicates are the caller's business: last writer wins)
}

// maximum of a 64-bit key over the 64 lanes of a wave, in every lane: DPP moves (quad swaps, row rotations, the two row
// broadcasts), no LDS traffic -- __shfl_down on a double and an int costs three ds_bpermute round trips per stage
template <int CTRL>
__device__ __forceinline__ unsigned long long gb_dpp_max(unsigned long long k) {
    const int lo = static_cast<int>(static_cast<unsigned>(k)), hi = static_cast<int>(static_cast<unsigned>(k >> 32));
    const unsigned olo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
    const unsigned ohi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false));
    const unsigned long long o = (static_cast<unsigned long long>(ohi) << 32) | olo;
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long gb_wave_max(unsigned long long k) {
    k = gb_dpp_max<0xb1>(k);  // quad_perm [1,0,3,2]
    k = gb_dpp_max<0x4e>(k);  // quad_perm [2,3,0,1]
    k = gb_dpp_max<0x124>(k); // row_ror 4
    k = gb_dpp_max<0x128>(k); // row_ror 8: every lane holds its row's maximum
    k = gb_dpp_max<0x142>(k); // row_bcast 15: rows 1..3 take in the row before them
    k = gb_dpp_max<0x143>(k); // row_bcast 31: lane 63 holds the wave's maximum
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(k)), 63));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(k >> 32)), 63));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// ------------------------------------------------------------------------------------------- panel
// Rows j0 .. j0 + R - 1 of the panel's columns j0 .. j0 + ncol - 1, a row per lane and slot: lane t holds rows
// t, t + T, ... (RPT of them).  LDS: the two rows of a swap, the reduction of the arg-max.
template <int T, int RPT>
struct GbPanelShared {
    double sA[GB_NB], sB[GB_NB];
    unsigned long long key[GB_NB]; // per column: the winning arg-max key (zeroed when the kernel starts)
    int piv[GB_NB], rep[GB_NB]; // the panel's pivot rows / replaced flags: to global memory once, at the end (a store
                                // ahead of a barrier makes the barrier wait for the store's acknowledgement: ~1 us a column)
};

// one column step of the panel, C a compile-time constant so that v[][C] stays in registers
template <int T, int RPT, int C>
__device__ __forceinline__ void gb_panel_step(double (&v)[RPT][GB_NB], GbPanelShared<T, RPT> &sh, double *__restrict__ ab, int ldab,
                                              int kl, int ku, int64_t j0, int ncol, int R, double tol,
                                              int32_t *__restrict__ ipiv, int32_t *__restrict__ replaced) {
    constexpr int c = C;
    const int tid = threadIdx.x;
    if (c >= ncol) return; // uniform
    // ---- arg-max of |a(rho, c)| over rho in [c, min(c + kl, R - 1)], ties to the smaller row: ONE 64-bit key per row,
    //      the value's bits with the low 11 replaced by 2047 - rho (values that agree to 2^-41 count as tied), reduced
    //      inside a wave by DPP moves (no LDS round trips) and across the waves by one LDS atomic per wave
    unsigned long long key = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rho = tid + k * T;
        if (rho >= c && rho <= c + kl && rho < R) {
            const unsigned long long kk = (static_cast<unsigned long long>(__double_as_longlong(fabs(v[k][c]))) & ~0x7FFull) |
                                          static_cast<unsigned long long>(2047 - rho);
            key = kk > key ? kk : key;
        }
    }
    key = gb_wave_max(key);
    if ((tid & 63) == 0) atomicMax(&sh.key[c], key);
    __syncthreads();
    key = sh.key[c];
    const int p_row = 2047 - static_cast<int>(key & 0x7FFull);
    const bool bad = !(__longlong_as_double(static_cast<long long>(key & ~0x7FFull)) > tol); // (NaN counts as unusable)
    if (tid == 0) {
        sh.piv[c] = bad ? c : p_row;
        sh.rep[c] = bad ? 1 : 0;
    }
    const int p = bad ? -1 : p_row;
    if (p < 0) {
        // no usable pivot: the column becomes the unit vector of the row on its diagonal
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int rho = tid + k * T;
            v[k][c] = (rho == c) ? 1.0 : 0.0;
        }
        // ... including its part above the panel
        const int64_t j = j0 + c;
        const int64_t i_lo = (j - ku - kl > 0) ? j - ku - kl : 0;
        for (int64_t i = i_lo + tid; i < j0; i += T) AB(ab, ldab, kl, ku, i, j) = 0.0;
        return;
    }
    // ---- swap rows c and p in the columns from c on (the multipliers to the left stay where they are:
    //      LAPACK's band convention -- a row moved down would leave the band), pivot row to every lane
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rho = tid + k * T;
        if (rho == p) {
#pragma unroll
            for (int q = c; q < GB_NB; ++q) sh.sB[q] = v[k][q];
        }
        if (rho == c && p != c) {
#pragma unroll
            for (int q = c; q < GB_NB; ++q) sh.sA[q] = v[k][q];
        }
    }
    __syncthreads();
    if (p != c) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int rho = tid + k * T;
            if (rho == c) {
#pragma unroll
                for (int q = c; q < GB_NB; ++q) v[k][q] = sh.sB[q];
            }
            if (rho == p) {
#pragma unroll
                for (int q = c; q < GB_NB; ++q) v[k][q] = sh.sA[q];
            }
        }
    }
    const double piv = sh.sB[c];
    // ---- multipliers and rank-one update of the columns to the right inside the panel
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rho = tid + k * T;
        if (rho > c && rho <= c + kl && rho < R) {
            const double l = v[k][c] / piv;
            v[k][c] = l;
#pragma unroll
            for (int q = c + 1; q < GB_NB; ++q) v[k][q] = v[k][q] - l * sh.sB[q];
        }
    }
    __syncthreads(); // sA / sB are rewritten by the next column
}

template <int T, int RPT, int C>
struct GbPanelSteps {
    static __device__ __forceinline__ void run(double (&v)[RPT][GB_NB], GbPanelShared<T, RPT> &sh, double *__restrict__ ab, int ldab,
                                               int kl, int ku, int64_t j0, int ncol, int R, double tol,
                                               int32_t *__restrict__ ipiv, int32_t *__restrict__ replaced) {
        gb_panel_step<T, RPT, C>(v, sh, ab, ldab, kl, ku, j0, ncol, R, tol, ipiv, replaced);
        GbPanelSteps<T, RPT, C + 1>::run(v, sh, ab, ldab, kl, ku, j0, ncol, R, tol, ipiv, replaced);
    }
};
template <int T, int RPT>
struct GbPanelSteps<T, RPT, GB_NB> {
    static __device__ __forceinline__ void run(double (&)[RPT][GB_NB], GbPanelShared<T, RPT> &, double *, int, int, int, int64_t, int,
                                               int, double, int32_t *, int32_t *) {}
};

template <int T, int RPT>
__global__ __launch_bounds__(T) void k_gb_panel(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n, int64_t j0_rel,
                                                double tol, int32_t *__restrict__ ipiv, int32_t *__restrict__ replaced, GbBlocks B) {
    __shared__ GbPanelShared<T, RPT> sh;
    const int tid = threadIdx.x;
    const int blk = blockIdx.x;
    if (j0_rel >= B.len(blk)) return; // (uniform: this block has no panel at this step)
    const int64_t j0 = B.base(blk) + j0_rel;
    const int ncol = static_cast<int>((B.len(blk) - j0_rel < GB_NB) ? B.len(blk) - j0_rel : GB_NB);
    const int64_t R64 = (n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol;
    const int R = static_cast<int>(R64);
    double v[RPT][GB_NB];
    if (tid < GB_NB) sh.key[tid] = 0; // (the first barrier of the first column step comes before any read)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rho = tid + k * T;
#pragma unroll
        for (int c = 0; c < GB_NB; ++c) {
            double x = 0.0;
            // (i, j) = (j0 + rho, j0 + c) is stored iff  c - ku - kl <= rho <= c + kl
            if (rho < R && c < ncol && rho <= c + kl && rho >= c - ku - kl) x = AB(ab, ldab, kl, ku, j0 + rho, j0 + c);
            v[k][c] = x;
        }
    }
    GbPanelSteps<T, RPT, 0>::run(v, sh, ab, ldab, kl, ku, j0, ncol, R, tol, ipiv, replaced);
    __syncthreads();
    if (tid < ncol) {
        ipiv[j0 + tid] = static_cast<int32_t>(j0 + sh.piv[tid]);
        replaced[j0 + tid] = sh.rep[tid];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int rho = tid + k * T;
#pragma unroll
        for (int c = 0; c < GB_NB; ++c)
            if (rho < R && c < ncol && rho <= c + kl && rho >= c - ku - kl) AB(ab, ldab, kl, ku, j0 + rho, j0 + c) = v[k][c];
    }
}

// ------------------------------------------------------------------------------------------- apply a panel
// Target: GB_CB columns per workgroup, each a vector indexed by global row.  BAND: columns jt0 + k of the band
// matrix itself (stored rows [j - ku - kl, j + kl]); otherwise columns of a dense block X (ldx).
// LDS holds rows j0 .. j0 + R - 1 of the targets; the panel's swaps, then its elimination steps.
template <bool BAND>
__device__ __forceinline__ void gb_apply_body(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n, int64_t j0,
                                                    int ncol, const int32_t *__restrict__ ipiv, int64_t jt0, int64_t ntgt,
                                                    double *__restrict__ X, int64_t ldx) {
    extern __shared__ double w[]; // [R][GB_CB]
    const int tid = threadIdx.x;
    const int R = static_cast<int>((n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol);
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    // ---- load
    for (int k = 0; k < GB_CB; ++k) {
        const int64_t t = t0 + k;
        for (int rho = tid; rho < R; rho += GB_T2) {
            double x = 0.0;
            if (t < ntgt) {
                const int64_t i = j0 + rho;
                if (BAND) {
                    const int64_t j = jt0 + t;
                    if (i >= j - ku - kl && i <= j + kl) x = AB(ab, ldab, kl, ku, i, j);
                } else {
                    x = X[i + t * ldx];
                }
            }
            w[rho * GB_CB + k] = x;
        }
    }
    __syncthreads();
    // ---- column by column: the step's row swap, then its elimination  w[rho] -= L(rho, c) * w[c], rho in (c, c + kl]
    for (int c = 0; c < ncol; ++c) {
        const int p = static_cast<int>(ipiv[j0 + c] - j0);
        if (p != c) { // uniform
            if (tid < GB_CB) {
                const double a = w[c * GB_CB + tid];
                w[c * GB_CB + tid] = w[p * GB_CB + tid];
                w[p * GB_CB + tid] = a;
            }
            __syncthreads();
        }
        const int hi = (c + kl < R - 1) ? c + kl : R - 1;
        double u[GB_CB];
#pragma unroll
        for (int k = 0; k < GB_CB; ++k) u[k] = w[c * GB_CB + k];
        for (int rho = c + 1 + tid; rho <= hi; rho += GB_T2) {
            const double l = AB(ab, ldab, kl, ku, j0 + rho, j0 + c);
            if (l != 0.0) {
#pragma unroll
                for (int k = 0; k < GB_CB; ++k) w[rho * GB_CB + k] = w[rho * GB_CB + k] - l * u[k];
            }
        }
        __syncthreads();
    }
    // ---- store
    for (int k = 0; k < GB_CB; ++k) {
        const int64_t t = t0 + k;
        if (t >= ntgt) continue;
        for (int rho = tid; rho < R; rho += GB_T2) {
            const int64_t i = j0 + rho;
            if (BAND) {
                const int64_t j = jt0 + t;
                if (i >= j - ku - kl && i <= j + kl) AB(ab, ldab, kl, ku, i, j) = w[rho * GB_CB + k];
            } else {
                X[i + t * ldx] = w[rho * GB_CB + k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- L^T with swaps (backward)
// for j descending: x_j -= sum_{i=j+1..j+kl} L(i, j) x_i; swap x_j <-> x_ipiv(j).  Rows j0 .. j0 + R - 1 in LDS.
__device__ __forceinline__ void gb_ltsolve_body(const double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                      int64_t j0, int ncol, const int32_t *__restrict__ ipiv, int64_t ntgt,
                                                      double *__restrict__ X, int64_t ldx) {
    extern __shared__ double w[];
    __shared__ double red[GB_T2];
    const int tid = threadIdx.x;
    const int R = static_cast<int>((n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol);
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    for (int k = 0; k < GB_CB; ++k) {
        const int64_t t = t0 + k;
        for (int rho = tid; rho < R; rho += GB_T2) w[rho * GB_CB + k] = (t < ntgt) ? X[j0 + rho + t * ldx] : 0.0;
    }
    __syncthreads();
    const int k = tid & (GB_CB - 1), lane = tid / GB_CB;
    constexpr int LPT = GB_T2 / GB_CB;
    for (int c = ncol - 1; c >= 0; --c) {
        const int hi = (c + kl < R - 1) ? c + kl : R - 1;
        double s = 0.0;
        for (int rho = c + 1 + lane; rho <= hi; rho += LPT)
            s += ab[static_cast<size_t>(kl + ku + rho - c) + static_cast<size_t>(j0 + c) * ldab] * w[rho * GB_CB + k];
        red[tid] = s;
        __syncthreads();
        if (tid < GB_CB) {
            double acc = 0.0;
            for (int q = 0; q < LPT; ++q) acc += red[q * GB_CB + tid];
            const double xj = w[c * GB_CB + tid] - acc;
            const int p = static_cast<int>(ipiv[j0 + c] - j0);
            if (p != c) {
                w[c * GB_CB + tid] = w[p * GB_CB + tid];
                w[p * GB_CB + tid] = xj;
            } else {
                w[c * GB_CB + tid] = xj;
            }
        }
        __syncthreads();
    }
    for (int kk = 0; kk < GB_CB; ++kk) {
        const int64_t t = t0 + kk;
        if (t >= ntgt) continue;
        for (int rho = tid; rho < R; rho += GB_T2) X[j0 + rho + t * ldx] = w[rho * GB_CB + kk];
    }
}

// ------------------------------------------------------------------------------------------- phase-split solves
// The kernels above walk a panel column by column (32 steps, a barrier or two each: 27-112 us per launch).  For a
// dense right-hand-side block the work of a panel splits into (i) a 32 x 32 triangular solve inside the panel -- the
// panel's own block staged in LDS, a lane per (unknown, right-hand side) -- and (ii) one pass that combines the rows
// outside the panel with the panel's 32 entries at once; no step depends on another inside (ii).  Forward / backward
// with L need the panel to be free of row swaps (the host knows: ipiv), else the step-by-step kernels run.
// A panel's step is a chain of latencies (one workgroup, one panel after the other: -DSX_GB_TICKS shows 4 us of loads,
// 2-3 us of triangle and 9-16 us in (ii) when its loads came one pass after the other), so everything a step reads from
// HBM is asked for in ONE round at its start: the window of the right-hand sides, the panel's own block and the
// factor's entries outside the panel that (ii) needs (GB_NPF passes of 64 rows, 8 entries per lane and pass).
// LDS: w[R][8] | blk[32][33] | pa[32][8][8] | part[32][8]
struct GbLds {
    double *w, *blk, *pa, *part;
    __device__ __forceinline__ GbLds(double *base, int R) {
        w = base;
        blk = w + static_cast<size_t>(R) * GB_CB;
        pa = blk + GB_NB * (GB_NB + 1);
        part = pa + GB_NB * 8 * GB_CB;
    }
};
__host__ __device__ inline size_t gb_lds_bytes(int R) {
    return sizeof(double) * (static_cast<size_t>(R) * GB_CB + GB_NB * (GB_NB + 1) + GB_NB * 8 * GB_CB + GB_NB * GB_CB);
}

// the panel's own 32 x 32 block of the factors (L strictly below the diagonal, U on and above) -> blk[r][c]
__device__ __forceinline__ void gb_load_block(const double *__restrict__ ab, int ldab, int kl, int ku, int64_t j0, int ncol,
                                              double *blk) {
    // (the padding slot of every row is written too: for a panel of fewer than 32 columns -- the matrix' last -- gb_combine
    //  multiplies 32 rows of x, i.e. up to 248 doubles of THIS block behind the panel's own rows of w, by factor entries that
    //  are zero there; a never-written slot held a NaN now and then: test_solves_match_lapack[50-3-2-0] failed once in ~10 runs)
    if (threadIdx.x < GB_NB) blk[threadIdx.x * (GB_NB + 1) + GB_NB] = 0.0;
    for (int e = threadIdx.x; e < GB_NB * GB_NB; e += GB_T2) {
        const int r = e / GB_NB, c = e % GB_NB;
        double v = 0.0;
        if (r < ncol && c < ncol && r - c <= kl && c - r <= ku + kl)
            v = ab[static_cast<size_t>(kl + ku + r - c) + static_cast<size_t>(j0 + c) * ldab];
        blk[r * (GB_NB + 1) + c] = v;
    }
}

// The 32 x 32 triangle of a panel, a lane per (unknown r, right-hand side): wave w holds the right-hand sides 2 w and
// 2 w + 1 in its two halves.  Column by column: x_c is final when its turn comes (after the multiplication by the
// reciprocal of the diagonal with DIAG) and goes to the 32 lanes of its right-hand side through v_readlane -- no LDS
// round trip in the chain --, every other unknown takes its term at once.  M(r, c) = blk[r][c], or blk[c][r] with TRANS;
// FWD: c ascending, the unknowns after c are updated (L, U^T); else c descending, those before c (U, L^T).  x: the
// lane's entry of the right-hand side in, of the solution out.  (One lane per right-hand side with the 32 unknowns in
// registers was a chain of 32 x ~350 cycles and, with its 496 LDS loads gathered at the top by the compiler, 2.5 KB of
// spills per lane; a division inside the chain costs ~200 cycles a step.)
template <bool FWD, bool TRANS, bool DIAG>
__device__ __forceinline__ double gb_triangle_wave(double x, const double *blk, int ncol) {
    const int lane = threadIdx.x & 63, r = lane & 31;
    const bool upper = lane >= 32;
    double m[GB_NB];
#pragma unroll
    for (int c = 0; c < GB_NB; ++c) m[c] = TRANS ? blk[c * (GB_NB + 1) + r] : blk[r * (GB_NB + 1) + c];
    double dinv = 1.0;
    if (DIAG) {
        if (r < ncol) dinv = 1.0 / blk[r * (GB_NB + 1) + r];
    }
#pragma unroll
    for (int s = 0; s < GB_NB; ++s) {
        const int c = FWD ? s : GB_NB - 1 - s;
        if (DIAG) {
            if (r == c) x = x * dinv;
        }
        const int lo = __double2loint(x), hi = __double2hiint(x);
        const double x0 = __hiloint2double(__builtin_amdgcn_readlane(hi, c), __builtin_amdgcn_readlane(lo, c));
        const double x1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32 + c), __builtin_amdgcn_readlane(lo, 32 + c));
        const double xc = upper ? x1 : x0;
        if (FWD ? r > c : r < c) x = x - m[c] * xc;
    }
    return x;
}

// (ii) for the forward / backward sweeps: w[rho] -= sum_c F(rho, c) x_c over the rows rho0 + [0, nrows) outside the panel.
// Lane = (64 rows per pass) x (a quarter of the 32 columns): 8 entries of the factor per lane and pass -- those of the
// first GB_NPF passes are in `pre` already, loaded at the step's start --, all 8 right-hand sides; the four quarters meet
// through two DPP quad swaps.  xrow: LDS row of x_0.  entry(rho, c) loads F (0.0 outside the band / panel).
constexpr int GB_NPF = 4;
__device__ __forceinline__ double gb_quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0xb1, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0xb1, 0xf, 0xf, false));
    v = v + o; // lanes 2q and 2q + 1 add the same two numbers: the same sum in both
    lo = __double2loint(v), hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x4e, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x4e, 0xf, 0xf, false));
    return v + o;
}
template <class Entry>
__device__ __forceinline__ void gb_prefetch(double (&pre)[GB_NPF][8], int rho0, int nrows, Entry entry) {
    const int cq = threadIdx.x & 3, rq = threadIdx.x >> 2;
#pragma unroll
    for (int it = 0; it < GB_NPF; ++it) {
        const int rr = rq + 64 * it;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) pre[it][cc] = (rr < nrows) ? entry(rho0 + rr, cq * 8 + cc) : 0.0;
    }
}
template <class Entry>
__device__ __forceinline__ void gb_combine(double *w, int xrow, const double (&pre)[GB_NPF][8], int rho0, int nrows, Entry entry) {
    const int cq = threadIdx.x & 3, rq = threadIdx.x >> 2;
    auto pass = [&](const double (&av)[8], int rr) {
        double acc[GB_CB];
#pragma unroll
        for (int k = 0; k < GB_CB; ++k) acc[k] = 0.0;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const double *xp = w + (xrow + cq * 8 + cc) * GB_CB;
#pragma unroll
            for (int k = 0; k < GB_CB; ++k) acc[k] += av[cc] * xp[k];
        }
#pragma unroll
        for (int k = 0; k < GB_CB; ++k) acc[k] = gb_quad_sum(acc[k]);
        if (rr < nrows) { // lane cq writes the right-hand sides 2 cq and 2 cq + 1 of its row
            double *wr = w + (rho0 + rr) * GB_CB;
#pragma unroll
            for (int k = 0; k < GB_CB; ++k)
                if ((k >> 1) == cq) wr[k] = wr[k] - acc[k];
        }
    };
#pragma unroll
    for (int it = 0; it < GB_NPF; ++it)
        if (64 * it < nrows) pass(pre[it], rq + 64 * it); // (uniform)
    for (int base = 64 * GB_NPF; base < nrows; base += 64) { // a band wider than 256 rows outside the panel: the rest on demand
        const int rr = rq + base;
        double av[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) av[cc] = (rr < nrows) ? entry(rho0 + rr, cq * 8 + cc) : 0.0;
        pass(av, rr);
    }
}

// U x = b, panel j0: rows i0 = max(0, j0 - ku - kl) .. j0 + ncol - 1
__device__ __forceinline__ void gb_usolve2_body(const double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                      int64_t j0, int ncol, int64_t ntgt, double *__restrict__ X, int64_t ldx) {
    extern __shared__ double lds_raw[];
    const int kw = ku + kl;
    const int64_t i0 = (j0 - kw > 0) ? j0 - kw : 0;
    const int R = static_cast<int>(j0 + ncol - i0), top = static_cast<int>(j0 - i0);
    GbLds L(lds_raw, R);
    const int tid = threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    GB_TICK(7);
    for (int k = 0; k < GB_CB; ++k)
        for (int rho = tid; rho < R; rho += GB_T2) L.w[rho * GB_CB + k] = (t0 + k < ntgt) ? X[i0 + rho + (t0 + k) * ldx] : 0.0;
    gb_load_block(ab, ldab, kl, ku, j0, ncol, L.blk);
    // U(i0 + rho, j0 + c), rho < top: stored iff i >= j - kw
    auto entry = [&](int rho, int c) -> double {
        const int64_t i = i0 + rho, j = j0 + c;
        return (c < ncol && i >= j - kw) ? ab[static_cast<size_t>(kl + ku + i - j) + static_cast<size_t>(j) * ldab] : 0.0;
    };
    double pre[GB_NPF][8];
    gb_prefetch(pre, 0, top, entry);
    __syncthreads();
    GB_TICK(4);
    { // the triangle: x_c = (b_c - sum_{c'' > c} U(c, c'') x_c'') / U(c, c)   (blk is zero beyond ncol)
        const int r = tid & 31, k = tid >> 5;
        double x = (r < ncol) ? L.w[(top + r) * GB_CB + k] : 0.0;
        x = gb_triangle_wave<false, false, true>(x, L.blk, ncol);
        if (r < ncol) L.w[(top + r) * GB_CB + k] = x;
    }
    __syncthreads();
    GB_TICK(5);
    gb_combine(L.w, top, pre, 0, top, entry); // rows above the panel: w[i] -= sum_c U(i, j0 + c) x_c
    __syncthreads();
    GB_TICK(6);
    for (int k = 0; k < GB_CB; ++k)
        if (t0 + k < ntgt)
            for (int rho = tid; rho < R; rho += GB_T2) X[i0 + rho + (t0 + k) * ldx] = L.w[rho * GB_CB + k];
}

// L forward for a panel WITHOUT row swaps: rows j0 .. j0 + R - 1.
// BAND: the targets are the band matrix' own columns jt0 + t to the right of the panel (factorisation).
template <bool BAND>
__device__ __forceinline__ void gb_lsolve2_body(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n, int64_t j0, int ncol,
                                                int64_t jt0, int64_t ntgt, double *__restrict__ X, int64_t ldx) {
    extern __shared__ double lds_raw[];
    const int R = static_cast<int>((n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol);
    GbLds L(lds_raw, R);
    const int tid = threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    GB_TICK(3);
    for (int k = 0; k < GB_CB; ++k) {
        const int64_t t = t0 + k;
        for (int rho = tid; rho < R; rho += GB_T2) {
            double x = 0.0;
            if (t < ntgt) {
                const int64_t i = j0 + rho;
                if (BAND) {
                    const int64_t j = jt0 + t;
                    if (i >= j - ku - kl && i <= j + kl) x = AB(ab, ldab, kl, ku, i, j);
                } else {
                    x = X[i + t * ldx];
                }
            }
            L.w[rho * GB_CB + k] = x;
        }
    }
    gb_load_block(ab, ldab, kl, ku, j0, ncol, L.blk);
    // L(j0 + rho, j0 + c), rho >= ncol: stored iff rho <= c + kl
    auto entry = [&](int rho, int c) -> double {
        return (c < ncol && rho <= c + kl) ? ab[static_cast<size_t>(kl + ku + rho - c) + static_cast<size_t>(j0 + c) * ldab] : 0.0;
    };
    double pre[GB_NPF][8];
    gb_prefetch(pre, ncol, R - ncol, entry);
    __syncthreads();
    GB_TICK(0);
    { // the triangle: x_c = b_c - sum_{c' < c} L(c, c') x_c'   (blk is zero outside the panel)
        const int r = tid & 31, k = tid >> 5;
        double x = (r < ncol && r < R) ? L.w[r * GB_CB + k] : 0.0;
        x = gb_triangle_wave<true, false, false>(x, L.blk, ncol);
        if (r < ncol && r < R) L.w[r * GB_CB + k] = x;
    }
    __syncthreads();
    GB_TICK(1);
    if (R > ncol) gb_combine(L.w, 0, pre, ncol, R - ncol, entry); // rows below the panel
    __syncthreads();
    GB_TICK(2);
    for (int k = 0; k < GB_CB; ++k) {
        const int64_t t = t0 + k;
        if (t >= ntgt) continue;
        for (int rho = tid; rho < R; rho += GB_T2) {
            const int64_t i = j0 + rho;
            if (BAND) {
                const int64_t j = jt0 + t;
                if (i >= j - ku - kl && i <= j + kl) AB(ab, ldab, kl, ku, i, j) = L.w[rho * GB_CB + k];
            } else {
                X[i + t * ldx] = L.w[rho * GB_CB + k];
            }
        }
    }
}

// the dot products of the transposed sweeps: for each of the panel's columns c the sum over the rows outside the panel
// of F(row, c) x_row, 8 lanes per column, each a strided share of the rows -- up to GB_DOT entries per lane loaded in one
// round (a loop of four at a time was one round trip to HBM per four) -- partial sums to pa[c][g][k]
constexpr int GB_DOT = 32;
template <class Entry>
__device__ __forceinline__ void gb_dots(const double *w, double *pa, int ncol, int rho_lo_common, Entry entry) {
    const int c = threadIdx.x >> 3, g = threadIdx.x & 7;
    double acc[GB_CB];
#pragma unroll
    for (int k = 0; k < GB_CB; ++k) acc[k] = 0.0;
    if (c < ncol) {
        int lo, hi; // rows [lo, hi) of the LDS window meet column c
        entry.range(c, lo, hi);
        for (int rb = lo + g; rb < hi; rb += 8 * GB_DOT) {
            double av[GB_DOT];
#pragma unroll
            for (int q = 0; q < GB_DOT; ++q) {
                const int rho = rb + 8 * q;
                av[q] = (rho < hi) ? entry(rho, c) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < GB_DOT; ++q) {
                const int rho = rb + 8 * q;
                if (rho < hi) {
#pragma unroll
                    for (int k = 0; k < GB_CB; ++k) acc[k] += av[q] * w[rho * GB_CB + k];
                }
            }
        }
    }
    (void)rho_lo_common;
#pragma unroll
    for (int k = 0; k < GB_CB; ++k) pa[(c * 8 + g) * GB_CB + k] = acc[k];
}

// U^T forward: x_j = (b_j - sum_{i < j} U(i, j) x_i) / U(j, j); rows i0 .. j0 + ncol - 1 in LDS
__device__ __forceinline__ void gb_utsolve2_body(const double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                       int64_t j0, int ncol, int64_t ntgt, double *__restrict__ X, int64_t ldx) {
    extern __shared__ double lds_raw[];
    const int kw = ku + kl;
    const int64_t i0 = (j0 - kw > 0) ? j0 - kw : 0;
    const int R = static_cast<int>(j0 + ncol - i0), top = static_cast<int>(j0 - i0);
    GbLds L(lds_raw, R);
    const int tid = threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    for (int k = 0; k < GB_CB; ++k)
        for (int rho = tid; rho < R; rho += GB_T2) L.w[rho * GB_CB + k] = (t0 + k < ntgt) ? X[i0 + rho + (t0 + k) * ldx] : 0.0;
    gb_load_block(ab, ldab, kl, ku, j0, ncol, L.blk);
    __syncthreads();
    struct Above { // U(i0 + rho, j0 + c) for the rows above the panel: rho in [max(j - kw, i0) - i0, top)
        const double *ab;
        int ldab, kl, ku, kw, top;
        int64_t i0, j0;
        __device__ __forceinline__ void range(int c, int &lo, int &hi) const {
            const int64_t j = j0 + c;
            lo = static_cast<int>(((j - kw > i0) ? j - kw : i0) - i0);
            hi = top;
        }
        __device__ __forceinline__ double operator()(int rho, int c) const {
            const int64_t i = i0 + rho, j = j0 + c;
            return ab[static_cast<size_t>(kl + ku + i - j) + static_cast<size_t>(j) * ldab];
        }
    } above{ab, ldab, kl, ku, kw, top, i0, j0};
    gb_dots(L.w, L.pa, ncol, 0, above);
    __syncthreads();
    {
        const int c = tid >> 3, k = tid & 7;
        double s = 0.0;
        for (int g = 0; g < 8; ++g) s += L.pa[(c * 8 + g) * GB_CB + k];
        L.part[c * GB_CB + k] = s;
    }
    __syncthreads();
    { // the triangle on b - (what the rows above contribute)
        const int r = tid & 31, k = tid >> 5;
        double x = (r < ncol) ? L.w[(top + r) * GB_CB + k] - L.part[r * GB_CB + k] : 0.0;
        x = gb_triangle_wave<true, true, true>(x, L.blk, ncol);
        if (r < ncol) L.w[(top + r) * GB_CB + k] = x;
    }
    __syncthreads();
    for (int k = 0; k < GB_CB; ++k)
        if (t0 + k < ntgt)
            for (int rho = top + tid; rho < R; rho += GB_T2) X[i0 + rho + (t0 + k) * ldx] = L.w[rho * GB_CB + k];
}

// L^T backward for a panel WITHOUT row swaps: x_j -= sum_{i > j} L(i, j) x_i
__device__ __forceinline__ void gb_ltsolve2_body(const double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                       int64_t j0, int ncol, int64_t ntgt, double *__restrict__ X, int64_t ldx) {
    extern __shared__ double lds_raw[];
    const int R = static_cast<int>((n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol);
    GbLds L(lds_raw, R);
    const int tid = threadIdx.x;
    const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GB_CB;
    for (int k = 0; k < GB_CB; ++k)
        for (int rho = tid; rho < R; rho += GB_T2) L.w[rho * GB_CB + k] = (t0 + k < ntgt) ? X[j0 + rho + (t0 + k) * ldx] : 0.0;
    gb_load_block(ab, ldab, kl, ku, j0, ncol, L.blk);
    __syncthreads();
    struct Below { // L(j0 + rho, j0 + c) for the rows below the panel: rho in [ncol, min(c + kl, R - 1)]
        const double *ab;
        int ldab, kl, ku, ncol, R;
        int64_t j0;
        __device__ __forceinline__ void range(int c, int &lo, int &hi) const {
            lo = ncol;
            hi = ((c + kl < R - 1) ? c + kl : R - 1) + 1;
        }
        __device__ __forceinline__ double operator()(int rho, int c) const {
            return ab[static_cast<size_t>(kl + ku + rho - c) + static_cast<size_t>(j0 + c) * ldab];
        }
    } below{ab, ldab, kl, ku, ncol, R, j0};
    gb_dots(L.w, L.pa, ncol, 0, below);
    __syncthreads();
    {
        const int c = tid >> 3, k = tid & 7;
        double s = 0.0;
        for (int g = 0; g < 8; ++g) s += L.pa[(c * 8 + g) * GB_CB + k];
        L.part[c * GB_CB + k] = s;
    }
    __syncthreads();
    const int lim = (ncol < R) ? ncol : R;
    { // the triangle   (blk is zero outside the panel)
        const int r = tid & 31, k = tid >> 5;
        double x = (r < lim) ? L.w[r * GB_CB + k] - L.part[r * GB_CB + k] : 0.0;
        x = gb_triangle_wave<false, true, false>(x, L.blk, ncol);
        if (r < lim) L.w[r * GB_CB + k] = x;
    }
    __syncthreads();
    for (int k = 0; k < GB_CB; ++k)
        if (t0 + k < ntgt)
            for (int rho = tid; rho < lim; rho += GB_T2) X[j0 + rho + (t0 + k) * ldx] = L.w[rho * GB_CB + k];
}

// ------------------------------------------------------------------------------------------- launch wrappers
// the factorisation's update of the columns to the right of a panel that swapped no rows
// (which it did the panel kernel has just decided: the branch is taken on the device, uniformly)
__global__ __launch_bounds__(GB_T2) void k_gb_trail2(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n, int64_t j0_rel,
                                                     const int32_t *__restrict__ ipiv, GbBlocks B) {
    __shared__ int swapped;
    const int blk = blockIdx.y;
    if (j0_rel >= B.len(blk)) return; // (uniform)
    const int64_t j0 = B.base(blk) + j0_rel;
    const int ncol = static_cast<int>((B.len(blk) - j0_rel < GB_NB) ? B.len(blk) - j0_rel : GB_NB);
    // columns to the right that the panel's rows reach: up to j0 + ncol - 1 + ku + kl
    const int64_t jt0 = j0 + ncol;
    const int64_t jt1 = (n < j0 + ncol + ku + kl) ? n : j0 + ncol + ku + kl;
    const int64_t ntgt = jt1 - jt0;
    if (static_cast<int64_t>(blockIdx.x) * GB_CB >= ntgt) return; // (uniform)
    if (threadIdx.x == 0) swapped = 0;
    __syncthreads();
    if (threadIdx.x < ncol && ipiv[j0 + threadIdx.x] != j0 + threadIdx.x) swapped = 1; // (every writer stores 1)
    __syncthreads();
    if (swapped) gb_apply_body<true>(ab, ldab, kl, ku, n, j0, ncol, ipiv, jt0, ntgt, nullptr, 0);
    else gb_lsolve2_body<true>(ab, ldab, kl, ku, n, j0, ncol, jt0, ntgt, nullptr, 0);
}
__global__ __launch_bounds__(256) void k_gb_iota(int64_t n, int32_t *__restrict__ ipiv) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j < n) ipiv[j] = static_cast<int32_t>(j);
}

// A whole solve in ONE launch: the right-hand sides of different workgroups never meet, so every workgroup walks
// all panels by itself (forward, then backward) -- no launch per panel, the factors stream through L2.
// trans = 0: A x = b (L with the row swaps, then U); 1: A^T x = b (U^T, then L^T with the swaps).
__global__ __launch_bounds__(GB_T2) void k_gb_solve_loop(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                         const int32_t *__restrict__ ipiv, const uint8_t *__restrict__ swaps,
                                                         int64_t ntgt, double *__restrict__ X, int64_t ldx, int trans, int stepwise) {
    const int64_t npanel = (n + GB_NB - 1) / GB_NB;
#ifdef SX_GB_TICKS
    const unsigned long long c0_ = clock64(), w0_ = wall_clock64();
#endif
    if (!trans) {
        for (int64_t p = 0; p < npanel; ++p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            if (stepwise || swaps[p]) gb_apply_body<false>(ab, ldab, kl, ku, n, j0, ncol, ipiv, 0, ntgt, X, ldx);
            else gb_lsolve2_body<false>(ab, ldab, kl, ku, n, j0, ncol, 0, ntgt, X, ldx);
            __syncthreads();
        }
        for (int64_t p = npanel - 1; p >= 0; --p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            gb_usolve2_body(ab, ldab, kl, ku, n, j0, ncol, ntgt, X, ldx);
            __syncthreads();
        }
    } else {
        for (int64_t p = 0; p < npanel; ++p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            gb_utsolve2_body(ab, ldab, kl, ku, n, j0, ncol, ntgt, X, ldx);
            __syncthreads();
        }
        for (int64_t p = npanel - 1; p >= 0; --p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            if (stepwise || swaps[p]) gb_ltsolve_body(ab, ldab, kl, ku, n, j0, ncol, ipiv, ntgt, X, ldx);
            else gb_ltsolve2_body(ab, ldab, kl, ku, n, j0, ncol, ntgt, X, ldx);
            __syncthreads();
        }
    }
#ifdef SX_GB_TICKS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_gb_t[8] = clock64() - c0_;
        g_gb_t[9] = wall_clock64() - w0_;
    }
#endif
}

// ------------------------------------------------------------------------------------------- sparse right-hand sides
// A x = b for right-hand sides that are mostly zeros (the columns of an LP: a handful of entries each).  A panel of the
// forward sweep whose rows hold nothing but zeros does nothing, and neither does a panel of the backward sweep; past the
// last entry of b the forward sweep's values decay geometrically (row by row, for a factor without growth) and are
// treated as zeros once all of a window's entries are <= tiny in magnitude.  tiny = 0: exact zeros only, the result is
// the plain solve's.  flags[g][p] = 1: panel p of workgroup g's right-hand sides holds an entry (set by k_gb_flag for
// b, extended by the forward sweep for what it fills in).
__global__ __launch_bounds__(256) void k_gb_flag(int64_t n, int64_t ntgt, const double *__restrict__ X, int64_t ldx, double tiny,
                                                 int64_t npanel, uint8_t *__restrict__ flags) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < n * ntgt; e += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t t = e / n, i = e - t * n; // (grid-stride: a launch carries fewer than 2^32 work-items)
        if (fabs(X[i + t * ldx]) > tiny) flags[(t / GB_CB) * npanel + i / GB_NB] = 1;
    }
}

// first flagged panel in [p, npanel) (npanel if none) / last flagged panel in [0, p] (-1 if none); every wave scans for itself
__device__ __forceinline__ int64_t gb_next_flag(const uint8_t *__restrict__ f, int64_t p, int64_t npanel) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = p; base < npanel; base += 64) {
        const int64_t q = base + lane;
        const unsigned long long bal = __ballot(q < npanel && f[q] != 0);
        if (bal) return base + __ffsll(static_cast<long long>(bal)) - 1;
    }
    return npanel;
}
__device__ __forceinline__ int64_t gb_prev_flag(const uint8_t *__restrict__ f, int64_t p) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = p; base >= 0; base -= 64) {
        const int64_t q = base - lane;
        const unsigned long long bal = __ballot(q >= 0 && f[q] != 0);
        if (bal) return base - (__ffsll(static_cast<long long>(bal)) - 1);
    }
    return -1;
}

__global__ __launch_bounds__(GB_T2) void k_gb_solve_sparse(double *__restrict__ ab, int ldab, int kl, int ku, int64_t n,
                                                           const int32_t *__restrict__ ipiv, const uint8_t *__restrict__ swaps,
                                                           int64_t ntgt, double *__restrict__ X, int64_t ldx, double tiny,
                                                           uint8_t *__restrict__ flags) {
    extern __shared__ double lds_raw[]; // (w[R][GB_CB] first in every body's layout)
    const int64_t npanel = (n + GB_NB - 1) / GB_NB;
    uint8_t *f = flags + static_cast<size_t>(blockIdx.x) * npanel;
    const int tid = threadIdx.x;
    const int kq = (kl + GB_NB - 1) / GB_NB; // panels a row swap can reach down
    // ---- forward: L with the swaps
    int active = 0;
    int64_t p = 0;
    while (p < npanel) {
        if (!active) {
            const int64_t q = gb_next_flag(f, p, npanel);
            if (q >= npanel) break;
            if (q - kq > p) p = q - kq;
        }
        const int64_t j0 = p * GB_NB;
        const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
        if (swaps[p]) gb_apply_body<false>(ab, ldab, kl, ku, n, j0, ncol, ipiv, 0, ntgt, X, ldx);
        else gb_lsolve2_body<false>(ab, ldab, kl, ku, n, j0, ncol, 0, ntgt, X, ldx);
        __syncthreads();
        const int R = static_cast<int>((n - j0 < static_cast<int64_t>(kl) + ncol) ? n - j0 : static_cast<int64_t>(kl) + ncol);
        int any = 0, own = 0;
        for (int e = tid; e < R * GB_CB; e += GB_T2) {
            const int hit = fabs(lds_raw[e]) > tiny;
            if (e >= ncol * GB_CB) any |= hit;
            else own |= hit;
        }
        active = __syncthreads_or(any);
        own = __syncthreads_or(own);
        // what the backward sweep will find in the panel's rows
        if (tid == 0 && own) f[p] = 1; // (the rows below are the next panels' own rows: an active sweep goes on to them)
        __syncthreads();
        ++p;
    }
    __syncthreads();
    // ---- backward: U
    active = 0;
    p = npanel - 1;
    while (p >= 0) {
        if (!active) {
            p = gb_prev_flag(f, p);
            if (p < 0) break;
        }
        const int64_t j0 = p * GB_NB;
        const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
        gb_usolve2_body(ab, ldab, kl, ku, n, j0, ncol, ntgt, X, ldx);
        __syncthreads();
        const int kw = ku + kl;
        const int top = static_cast<int>(j0 - ((j0 - kw > 0) ? j0 - kw : 0));
        int any = 0;
        for (int e = tid; e < top * GB_CB; e += GB_T2) any |= fabs(lds_raw[e]) > tiny;
        active = __syncthreads_or(any);
        --p;
    }
}

// ------------------------------------------------------------------------------------------- partitioned sweeps
// A sweep of a solve is ONE chain of n / 32 panel steps on ONE workgroup (85-95 ms at 1e5 rows, whatever the number of
// right-hand sides).  A panel reaches w rows beyond itself (w = kl for the sweeps with L, kl + ku for those with U), so
// the panels are cut into P blocks of >= w rows: block i's result is a linear function of its own right-hand side and
// of w "incoming" rows that the block before it (in sweep order) finishes,
//      result_i = sweep_i(own rows, assumed incoming rows) + R_i (true incoming - assumed incoming),
// with R_i = the block's sweep applied to the w unit vectors of its incoming rows -- a property of the factors, computed
// once per handle.  A solve then is  (1) every block's sweep on a private copy of its rows, all blocks at once;
// (2) a chain over the blocks of w x w products for the true incoming rows (true outgoing_i = outgoing_i + M_i delta_i,
// M_i = the outgoing rows of R_i); (3) result += R_i delta_i, all blocks at once.  P = 64 .. 128: ~50 panel steps + P small
// products instead of n / 32 steps.  Incoming / outgoing rows by kind (s, e = the block's first row and the one after
// its last; the buffer holds the rows [lo, hi)):
//   0  L forward    buffer [s, e + w)   in = first w own rows (assumed: b)        out = the w rows after e (they ARE the next block's in)
//   1  U backward   buffer [s - w, e)   in = last w own rows (assumed: b)         out = the w rows before s
//   2  U^T forward  buffer [s - w, e)   in = the w rows before s (assumed: 0)     out = last w own rows
//   3  L^T backward buffer [s, e + w)   in = the w rows after e (assumed: 0)      out = first w own rows; the row swaps of the
//                                       block's columns reach into its in rows: their final values are this block's (k_gbp_fix)
struct GbPart {
    int kind, w, LBp, P;
    int64_t n, npanel;
    int ld_small, ld_last; // rows of a block's buffer (all blocks but the last / the last)
    __host__ __device__ int64_t p0(int i) const { return static_cast<int64_t>(i) * LBp; }
    __host__ __device__ int64_t p1(int i) const { return i == P - 1 ? npanel : static_cast<int64_t>(i + 1) * LBp; }
    __host__ __device__ int64_t s(int i) const { return p0(i) * GB_NB; }
    __host__ __device__ int64_t e(int i) const { return i == P - 1 ? n : p1(i) * GB_NB; }
    __host__ __device__ int64_t lo(int i) const { return (kind == 0 || kind == 3) ? s(i) : (s(i) - w > 0 ? s(i) - w : 0); }
    __host__ __device__ int64_t hi(int i) const { return (kind == 0 || kind == 3) ? (e(i) + w < n ? e(i) + w : n) : e(i); }
    __host__ __device__ int ld(int i) const { return i == P - 1 ? ld_last : ld_small; }
    __host__ __device__ size_t rows_before(int i) const { return static_cast<size_t>(i) * ld_small; } // buffer rows of the blocks before i
    __host__ __device__ size_t rows_total() const { return static_cast<size_t>(P - 1) * ld_small + ld_last; }
    __host__ __device__ bool forward() const { return kind == 0 || kind == 2; }
    // global rows of the incoming / outgoing rows (w of them; a block at the start of the sweep has no incoming ones)
    __host__ __device__ int64_t in0(int i) const { return kind == 0 ? s(i) : kind == 1 ? e(i) - w : kind == 2 ? s(i) - w : e(i); }
    __host__ __device__ int64_t out0(int i) const { return kind == 0 ? e(i) : kind == 1 ? s(i) - w : kind == 2 ? e(i) - w : s(i); }
    __host__ __device__ bool first_in_order(int i) const { return forward() ? i == 0 : i == P - 1; }
    __host__ __device__ bool last_in_order(int i) const { return forward() ? i == P - 1 : i == 0; }
};

// private copies: buf_i[t][r] = X[lo_i + r, t]; rows outside the block's own rows are zero for the pulling kinds (2, 3)
__global__ __launch_bounds__(256) void k_gbp_gather(GbPart D, int ncols, const double *__restrict__ X, int64_t ldx, double *__restrict__ bufs) {
    const int i = blockIdx.y;
    const int ld = D.ld(i);
    const int64_t lo = D.lo(i), hi = D.hi(i), s = D.s(i), e = D.e(i);
    double *b = bufs + D.rows_before(i) * ncols;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; q < static_cast<int64_t>(ld) * ncols; q += static_cast<int64_t>(gridDim.x) * 256) {
        const int t = static_cast<int>(q / ld);
        const int64_t g = lo + (q - static_cast<int64_t>(t) * ld);
        double v = 0.0;
        if (g < hi && (D.kind < 2 || (g >= s && g < e))) v = X[g + t * ldx];
        b[q] = v;
    }
}
// the w unit vectors of block i's incoming rows as its right-hand sides (-> R_i after the sweep)
__global__ __launch_bounds__(256) void k_gbp_unit(GbPart D, double *__restrict__ R) {
    const int i = blockIdx.y;
    const int ld = D.ld(i);
    double *b = R + D.rows_before(i) * D.w;
    const int64_t lo = D.lo(i), in0 = D.in0(i);
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; q < static_cast<int64_t>(ld) * D.w; q += static_cast<int64_t>(gridDim.x) * 256) {
        const int c = static_cast<int>(q / ld);
        const int64_t g = lo + (q - static_cast<int64_t>(c) * ld);
        b[q] = (!D.first_in_order(i) && g == in0 + c) ? 1.0 : 0.0;
    }
}
// block blockIdx.y's panels on its private rows; right-hand sides 8 per workgroup (blockIdx.x), as in k_gb_solve_loop
__global__ __launch_bounds__(GB_T2) void k_gbp_sweep(double *__restrict__ ab, int ldab, int kl, int ku, const int32_t *__restrict__ ipiv,
                                                     const uint8_t *__restrict__ swaps, GbPart D, int ncols, double *__restrict__ bufs) {
    const int i = blockIdx.y;
    const int64_t n = D.n;
    double *X = bufs + D.rows_before(i) * ncols - D.lo(i); // (global row numbers address the private copy)
    const int64_t ldx = D.ld(i);
    const int64_t p0 = D.p0(i), p1 = D.p1(i);
    if (D.forward()) {
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            if (D.kind == 0) {
                if (swaps[p]) gb_apply_body<false>(ab, ldab, kl, ku, n, j0, ncol, ipiv, 0, ncols, X, ldx);
                else gb_lsolve2_body<false>(ab, ldab, kl, ku, n, j0, ncol, 0, ncols, X, ldx);
            } else {
                gb_utsolve2_body(ab, ldab, kl, ku, n, j0, ncol, ncols, X, ldx);
            }
            __syncthreads();
        }
    } else {
        for (int64_t p = p1 - 1; p >= p0; --p) {
            const int64_t j0 = p * GB_NB;
            const int ncol = static_cast<int>((n - j0 < GB_NB) ? n - j0 : GB_NB);
            if (D.kind == 1) {
                gb_usolve2_body(ab, ldab, kl, ku, n, j0, ncol, ncols, X, ldx);
            } else {
                if (swaps[p]) gb_ltsolve_body(ab, ldab, kl, ku, n, j0, ncol, ipiv, ncols, X, ldx);
                else gb_ltsolve2_body(ab, ldab, kl, ku, n, j0, ncol, ncols, X, ldx);
            }
            __syncthreads();
        }
    }
}
// the chain: delta_i for every block, one workgroup of 1,024 lanes per right-hand side: a w x w product per block, rows
// along the lanes (R is stored column by column: coalesced), the columns cut into GBP_SL slices that are summed through
// LDS.  delta of the first block in order is zero.
constexpr int GBP_T = 1024, GBP_SL = 4;
constexpr int GB_PART_CHUNK = 128;  // right-hand sides per partitioned sweep
constexpr int GB_PART_MAX = 1024;   // beyond: one sequential walk of all panels for all right-hand sides at once is cheaper
__global__ __launch_bounds__(GBP_T) void k_gbp_chain(GbPart D, int ncols, const double *__restrict__ bufs, const double *__restrict__ R,
                                                      const double *__restrict__ X, int64_t ldx, double *__restrict__ delta) {
    extern __shared__ double sd[]; // cur[w] | part[GBP_SL][w]
    const int t = blockIdx.x, w = D.w;
    double *cur = sd, *part = sd + w;
    const int lane_r = threadIdx.x % (GBP_T / GBP_SL), slice = threadIdx.x / (GBP_T / GBP_SL);
    const int c_lo = static_cast<int>(static_cast<long long>(w) * slice / GBP_SL), c_hi = static_cast<int>(static_cast<long long>(w) * (slice + 1) / GBP_SL);
    for (int c = threadIdx.x; c < w; c += GBP_T) cur[c] = 0.0;
    __syncthreads();
    for (int step = 0; step < D.P; ++step) {
        const int i = D.forward() ? step : D.P - 1 - step;
        double *di = delta + (static_cast<size_t>(i) * ncols + t) * w;
        for (int c = threadIdx.x; c < w; c += GBP_T) di[c] = cur[c];
        if (D.last_in_order(i)) break;
        const int ld = D.ld(i);
        const int64_t lo = D.lo(i), out0 = D.out0(i);
        const double *b = bufs + D.rows_before(i) * ncols + static_cast<size_t>(t) * ld; // block i, right-hand side t
        const double *Ri = R + D.rows_before(i) * w;                                       // block i: column c at c * ld
        const int ro = static_cast<int>(out0 - lo);
        const bool zero = D.first_in_order(i);
        for (int r = lane_r; r < w; r += GBP_T / GBP_SL) {
            double acc = 0.0;
            if (!zero) {
                const double *col = Ri + ro + r;
                int c = c_lo;
                for (; c + 8 <= c_hi; c += 8) { // eight loads in flight
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = col[static_cast<size_t>(c + q) * ld];
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc += v[q] * cur[c + q];
                }
                for (; c < c_hi; ++c) acc += col[static_cast<size_t>(c) * ld] * cur[c];
            }
            part[slice * w + r] = acc;
        }
        __syncthreads();
        for (int r = threadIdx.x; r < w; r += GBP_T) {
            double acc = b[ro + r]; // what the block's own sweep left in its outgoing rows
            for (int q = 0; q < GBP_SL; ++q) acc += part[q * w + r];
            // the next block assumed b (kinds 0, 1) or zero (kinds 2, 3) in these rows
            cur[r] = acc - (D.kind < 2 ? X[out0 + r + t * ldx] : 0.0);
        }
        __syncthreads();
    }
}
// own rows of every block: X = private result + R_i delta_i
__global__ __launch_bounds__(256) void k_gbp_fix(GbPart D, int ncols, const double *__restrict__ bufs, const double *__restrict__ R,
                                                 const double *__restrict__ delta, double *__restrict__ X, int64_t ldx) {
    extern __shared__ double sd[]; // [w]
    const int i = blockIdx.y, t = blockIdx.z, w = D.w;
    const double *di = delta + (static_cast<size_t>(i) * ncols + t) * w;
    for (int c = threadIdx.x; c < w; c += 256) sd[c] = di[c];
    __syncthreads();
    const int ld = D.ld(i);
    const int64_t lo = D.lo(i);
    // the rows block i has the last word on: its own -- except for L^T with the row swaps (kind 3), where the step of
    // column j exchanges x_j with a row up to kl below it: the w rows after e are rewritten by block i after block i + 1
    // finished them, so they are block i's, and its own first w rows are block i - 1's
    int64_t s = D.s(i), e = D.e(i);
    if (D.kind == 3) {
        if (i > 0) s += w;
        if (i < D.P - 1) e += w;
    }
    const double *b = bufs + D.rows_before(i) * ncols + static_cast<size_t>(t) * ld;
    const double *Ri = R + D.rows_before(i) * w;
    const bool zero = D.first_in_order(i);
    for (int64_t g = s + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; g < e; g += static_cast<int64_t>(gridDim.x) * 256) {
        const int r = static_cast<int>(g - lo);
        double acc = b[r];
        if (!zero)
            for (int c = 0; c < w; ++c) acc += Ri[static_cast<size_t>(c) * ld + r] * sd[c];
        X[g + t * ldx] = acc;
    }
}

} // namespace

// (internal, sx_internal.h) the widths sx_bandlu_create_dev accepts: the panel kernel's rows and the solve kernels' LDS block
bool sx_bandlu_supports(int kl, int ku) {
    return kl >= 0 && ku >= 0 && static_cast<int64_t>(kl) + GB_NB <= 1536 && gb_lds_bytes(kl + ku + GB_NB) <= 150 * 1024;
}

SX_API int sx_bandlu_create_dev(sx_ctx *ctx, int64_t n, int kl, int ku, int64_t nnz, const int32_t *row, const int32_t *col,
                                const double *val, sx_bandlu **out) {
    SX_ENTER(ctx);
    SX_REQUIRE(out && n > 0 && kl >= 0 && ku >= 0 && nnz >= 0, "bad argument");
    SX_REQUIRE(nnz == 0 || (row && col && val), "NULL triplets");
    SX_REQUIRE(static_cast<int64_t>(kl) + GB_NB <= 1536, "band too wide for the panel kernel (kl + 32 > 1536 rows)");
    SX_REQUIRE(gb_lds_bytes(kl + ku + GB_NB) <= 150 * 1024, "band too wide for the solve kernels' LDS block (kl + ku + 32 > 1950 rows)");
    sx_bandlu *h = new (std::nothrow) sx_bandlu();
    SX_REQUIRE(h != nullptr, "out of host memory");
    h->ctx = ctx;
    h->n = n;
    h->kl = kl;
    h->ku = ku;
    h->ldab = 2 * kl + ku + 1;
    struct Guard {
        sx_bandlu *h;
        ~Guard() {
            if (h) {
                (void)sx_dfree(h->ab);
                (void)sx_dfree(h->ipiv);
                (void)sx_dfree(h->d_swaps);
                (void)sx_dfree(h->d_flags);
                delete h;
            }
        }
    } guard{h};
    const size_t bytes = sizeof(double) * static_cast<size_t>(h->ldab) * static_cast<size_t>(n);
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->ab), bytes));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->ipiv), sizeof(int32_t) * (2 * static_cast<size_t>(n) + 4)));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->d_swaps), static_cast<size_t>((n + GB_NB - 1) / GB_NB) + 8));
    h->replaced = h->ipiv + n;
    h->err = h->replaced + n;
    hipStream_t s = ctx->stream;
    {   // the LDS blocks of the apply / solve kernels pass the 64 KiB a launch gets by default
        const int cap = 150 * 1024;
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gb_solve_loop), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gb_trail2), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gb_solve_sparse), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gbp_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gbp_chain), hipFuncAttributeMaxDynamicSharedMemorySize, cap));
    }
    SX_HIP(hipMemsetAsync(h->ab, 0, bytes, s));
    SX_HIP(hipMemsetAsync(h->ipiv, 0, sizeof(int32_t) * (2 * static_cast<size_t>(n) + 4), s));
    if (nnz > 0)
        hipLaunchKernelGGL(k_gb_scatter, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0, s, nnz, row, col, val, h->ab,
                           h->ldab, kl, ku, n, h->err);
    int32_t err = 0;
    SX_HIP(hipMemcpyAsync(&err, h->err, sizeof(err), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    SX_HIP(hipGetLastError());
    SX_REQUIRE(err == 0, "an entry lies outside the band (kl = %d, ku = %d) or outside the matrix", kl, ku);
    guard.h = nullptr;
    *out = h;
    return SX_OK;
}

SX_API int sx_bandlu_destroy(sx_bandlu *h) {
    if (!h) return SX_OK;
    sx_device_guard guard(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    (void)sx_dfree(h->ab);
    (void)sx_dfree(h->ipiv);
    (void)sx_dfree(h->d_swaps);
    (void)sx_dfree(h->d_flags);
    for (double *r : h->part_R) (void)sx_dfree(r);
    (void)sx_dfree(h->part_buf);
    (void)sx_dfree(h->part_delta);
    delete h;
    return SX_OK;
}

namespace {
int gb_part_prepare(sx_bandlu *h);
}

SX_API int sx_bandlu_factor_dev(sx_bandlu *h, double pivot_tol, int64_t *n_replaced_out, int32_t *replaced_host,
                                int32_t *ipiv_host) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    return sx_bandlu_factor_blocks_dev(h, pivot_tol, 1, h->n, h->n, h->n, n_replaced_out, replaced_host, ipiv_host);
}

// The same factorisation for a matrix that is BLOCK DIAGONAL with identity padding between the blocks: block b = columns
// (and rows) [b stride, b stride + real_len) -- real_len_last for the last one --, then identity up to the next block; no
// entry couples two blocks.  The blocks' panels are factored side by side, one launch per panel STEP instead of one per
// panel (a panel is a chain of 32 dependent column steps on one workgroup: 53 us at kl + ku = 230; 3,100 of them in a row
// at 1e5 rows).  The padding (>= kl + ku + 32 positions, required) keeps what a block's last panels touch -- rows up to
// kl + 32 below, columns up to ku + kl + 32 to the right -- away from the next block's entries.
SX_API int sx_bandlu_factor_blocks_dev(sx_bandlu *h, double pivot_tol, int nblocks, int64_t stride, int64_t real_len,
                                       int64_t real_len_last, int64_t *n_replaced_out, int32_t *replaced_host, int32_t *ipiv_host) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(!h->factored, "already factored");
    hipStream_t s = ctx->stream;
    const int kl = h->kl, ku = h->ku;
    const int64_t n = h->n;
    SX_REQUIRE(nblocks >= 1 && real_len > 0 && real_len_last > 0, "bad block description");
    if (nblocks == 1) {
        SX_REQUIRE(real_len_last == n, "one block covers the matrix");
    } else {
        SX_REQUIRE(stride >= real_len + kl + ku + GB_NB, "blocks need kl + ku + 32 positions of identity padding between them");
        SX_REQUIRE(static_cast<int64_t>(nblocks - 1) * stride + real_len_last == n, "the blocks do not add up to the matrix");
    }
    const GbBlocks B{nblocks, stride, nblocks == 1 ? n : real_len, real_len_last};
    hipLaunchKernelGGL(k_gb_iota, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, n, h->ipiv); // (padding: no swaps)
    const int rows = kl + GB_NB; // rows of a panel: a row per lane and register slot
    const int64_t longest = std::max<int64_t>(B.rl, B.rl_last);
    const unsigned tg = static_cast<unsigned>(std::max(1, (ku + kl + GB_CB - 1) / GB_CB));
    const unsigned pb = static_cast<unsigned>(nblocks);
    for (int64_t j0 = 0; j0 < longest; j0 += GB_NB) {
        if (rows <= 256) // (one wave with three rows per lane measured slower: 88 us per panel against 74 us)
            hipLaunchKernelGGL((k_gb_panel<256, 1>), dim3(pb), dim3(256), 0, s, h->ab, h->ldab, kl, ku, n, j0, pivot_tol, h->ipiv, h->replaced, B);
        else if (rows <= 512)
            hipLaunchKernelGGL((k_gb_panel<512, 1>), dim3(pb), dim3(512), 0, s, h->ab, h->ldab, kl, ku, n, j0, pivot_tol, h->ipiv, h->replaced, B);
        else if (rows <= 1024)
            hipLaunchKernelGGL((k_gb_panel<512, 2>), dim3(pb), dim3(512), 0, s, h->ab, h->ldab, kl, ku, n, j0, pivot_tol, h->ipiv, h->replaced, B);
        else
            hipLaunchKernelGGL((k_gb_panel<512, 3>), dim3(pb), dim3(512), 0, s, h->ab, h->ldab, kl, ku, n, j0, pivot_tol, h->ipiv, h->replaced, B);
        if (ku + kl > 0)
            hipLaunchKernelGGL(k_gb_trail2, dim3(tg, pb), dim3(GB_T2), gb_lds_bytes(kl + GB_NB), s, h->ab, h->ldab, kl, ku, n, j0, h->ipiv, B);
    }
    SX_HIP(hipGetLastError());
    h->factored = true;
    std::vector<int32_t> rep(static_cast<size_t>(n)), piv(static_cast<size_t>(n));
    SX_HIP(hipMemcpyAsync(rep.data(), h->replaced, sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(piv.data(), h->ipiv, sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (ipiv_host) std::memcpy(ipiv_host, piv.data(), sizeof(int32_t) * static_cast<size_t>(n));
    h->panel_swaps.assign(static_cast<size_t>((n + GB_NB - 1) / GB_NB), 0);
    for (int64_t j = 0; j < n; ++j)
        if (piv[static_cast<size_t>(j)] != j) h->panel_swaps[static_cast<size_t>(j / GB_NB)] = 1;
    SX_HIP(hipMemcpyAsync(h->d_swaps, h->panel_swaps.data(), h->panel_swaps.size(), hipMemcpyHostToDevice, s));
    SX_HIP(hipStreamSynchronize(s));
    int64_t cnt = 0;
    for (int64_t j = 0; j < n; ++j) cnt += rep[static_cast<size_t>(j)] != 0;
    if (n_replaced_out) *n_replaced_out = cnt;
    if (replaced_host) std::memcpy(replaced_host, rep.data(), sizeof(int32_t) * static_cast<size_t>(n));
    // the partitioned sweeps' responses now (a few block sweeps), while the memory they need is still free
    SX_TRY(gb_part_prepare(h));
    return SX_OK;
}


namespace {
GbPart gb_part_desc(const sx_bandlu *h, int kind) {
    GbPart D{};
    D.kind = kind;
    D.w = (kind == 0 || kind == 3) ? h->kl : h->kl + h->ku;
    D.LBp = h->part_LBp;
    D.P = h->part_P;
    D.n = h->n;
    D.npanel = (h->n + GB_NB - 1) / GB_NB;
    D.ld_small = D.LBp * GB_NB + D.w;
    D.ld_last = static_cast<int>(h->n - D.s(D.P - 1)) + D.w;
    return D;
}
// one sweep of kind `kind` over `ncols` columns held in `bufs` (every block's private rows)
void gb_part_launch_sweep(sx_bandlu *h, const GbPart &D, int ncols, double *bufs) {
    hipLaunchKernelGGL(k_gbp_sweep, dim3(static_cast<unsigned>((ncols + GB_CB - 1) / GB_CB), static_cast<unsigned>(D.P)), dim3(GB_T2),
                       gb_lds_bytes(h->kl + h->ku + GB_NB), h->ctx->stream, h->ab, h->ldab, h->kl, h->ku, h->ipiv, h->d_swaps, D, ncols, bufs);
}
// blocks, responses and work buffers of the partitioned sweeps; leaves part_state = -1 when they are not worth it or do
// not fit (the caller then takes the sequential sweeps)
int gb_part_prepare(sx_bandlu *h) {
    h->part_state = -1;
    if (getenv("SX_BANDLU_SEQ")) return SX_OK;
    const int64_t npanel = (h->n + GB_NB - 1) / GB_NB;
    const int kw = h->kl + h->ku;
    const int min_lbp = (kw + GB_NB - 1) / GB_NB + 1; // a block holds the reach of a panel
    int64_t P = std::min<int64_t>(128, npanel / std::max(min_lbp, 8));
    if (P < 4 || kw < 1) return SX_OK;
    h->part_LBp = static_cast<int>(npanel / P);
    h->part_P = static_cast<int>(P);
    if (static_cast<int64_t>(h->part_LBp) * GB_NB < kw) return SX_OK;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return SX_OK;
    size_t need = 0;
    for (int kind = 0; kind < 4; ++kind) {
        const GbPart D = gb_part_desc(h, kind);
        need += sizeof(double) * D.rows_total() * D.w;
    }
    const GbPart DU = gb_part_desc(h, 1);
    need += sizeof(double) * (DU.rows_total() * GB_CB + static_cast<size_t>(DU.P) * GB_CB * DU.w);
    if (static_cast<double>(need) > 0.05 * static_cast<double>(free_b)) return SX_OK; // (the crossover's tableau takes what is free)
    hipStream_t s = h->ctx->stream;
    for (int kind = 0; kind < 4; ++kind) {
        const GbPart D = gb_part_desc(h, kind);
        if (sx_dmalloc(&h->part_R[kind], sizeof(double) * D.rows_total() * D.w) != hipSuccess) return SX_OK;
        hipLaunchKernelGGL(k_gbp_unit, dim3(64, static_cast<unsigned>(D.P)), dim3(256), 0, s, D, h->part_R[kind]);
        gb_part_launch_sweep(h, D, D.w, h->part_R[kind]);
    }
    if (sx_dmalloc(&h->part_buf, sizeof(double) * DU.rows_total() * GB_CB) != hipSuccess) return SX_OK;
    if (sx_dmalloc(&h->part_delta, sizeof(double) * static_cast<size_t>(DU.P) * GB_CB * DU.w) != hipSuccess) return SX_OK;
    SX_HIP(hipGetLastError());
    h->part_cols = GB_CB;
    h->part_state = 1;
    return SX_OK;
}
// work buffers of the partitioned sweeps for `ncols` right-hand sides at once (the blocks of one sweep run side by side per
// group of 8 columns, the chain per column: 64 columns cost about what 8 do); keeps what it has when the memory is short
void gb_part_reserve(sx_bandlu *h, int ncols) {
    if (ncols <= h->part_cols) return;
    const GbPart DU = gb_part_desc(h, 1);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const size_t need = sizeof(double) * (DU.rows_total() + static_cast<size_t>(DU.P) * DU.w) * static_cast<size_t>(ncols);
    if (static_cast<double>(need) > 0.25 * static_cast<double>(free_b)) return;
    (void)hipStreamSynchronize(h->ctx->stream);
    double *buf = nullptr, *delta = nullptr;
    if (sx_dmalloc(&buf, sizeof(double) * DU.rows_total() * static_cast<size_t>(ncols)) != hipSuccess) return;
    if (sx_dmalloc(&delta, sizeof(double) * static_cast<size_t>(DU.P) * static_cast<size_t>(ncols) * DU.w) != hipSuccess) {
        (void)sx_dfree(buf);
        return;
    }
    (void)sx_dfree(h->part_buf);
    (void)sx_dfree(h->part_delta);
    h->part_buf = buf;
    h->part_delta = delta;
    h->part_cols = ncols;
}
void gb_part_sweep(sx_bandlu *h, int kind, int ncols, double *X, int64_t ldx) {
    const GbPart D = gb_part_desc(h, kind);
    hipStream_t s = h->ctx->stream;
    hipLaunchKernelGGL(k_gbp_gather, dim3(16, static_cast<unsigned>(D.P)), dim3(256), 0, s, D, ncols, X, ldx, h->part_buf);
    gb_part_launch_sweep(h, D, ncols, h->part_buf);
    hipLaunchKernelGGL(k_gbp_chain, dim3(static_cast<unsigned>(ncols)), dim3(GBP_T), sizeof(double) * (1 + GBP_SL) * D.w, s, D, ncols, h->part_buf, h->part_R[kind], X, ldx,
                       h->part_delta);
    hipLaunchKernelGGL(k_gbp_fix, dim3(8, static_cast<unsigned>(D.P), static_cast<unsigned>(ncols)), dim3(256), sizeof(double) * D.w, s, D, ncols, h->part_buf,
                       h->part_R[kind], h->part_delta, X, ldx);
}
} // namespace

// In-place solve of nrhs right-hand sides X (column major, ldx >= n): trans = 0: A x = b, 1: A^T x = b.
SX_API int sx_bandlu_solve_dev(sx_bandlu *h, int trans, int64_t nrhs, double *X, int64_t ldx) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(h->factored, "factor first");
    SX_REQUIRE(X && ldx >= h->n && nrhs >= 0, "bad right-hand side block");
    if (nrhs == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    const int kl = h->kl, ku = h->ku;
    const int64_t n = h->n;
    const unsigned grid = static_cast<unsigned>((nrhs + GB_CB - 1) / GB_CB);
    const int64_t npanel = (n + GB_NB - 1) / GB_NB;
    static const bool slow = getenv("SX_BANDLU_STEPWISE") != nullptr; // the step-by-step panel bodies everywhere (A/B runs, tests)
    (void)npanel;
    if (nrhs <= GB_PART_MAX && !slow) { // dense right-hand sides, not thousands: the sweeps cut into blocks that run side by side
        if (h->part_state == 0) SX_TRY(gb_part_prepare(h));
        if (h->part_state == 1) {
            gb_part_reserve(h, static_cast<int>(std::min<int64_t>(nrhs, GB_PART_CHUNK)));
            for (int64_t c0 = 0; c0 < nrhs; c0 += h->part_cols) {
                const int kc = static_cast<int>(std::min<int64_t>(h->part_cols, nrhs - c0));
                double *Xc = X + static_cast<size_t>(c0) * ldx;
                gb_part_sweep(h, trans ? 2 : 0, kc, Xc, ldx);
                gb_part_sweep(h, trans ? 3 : 1, kc, Xc, ldx);
            }
            SX_HIP(hipGetLastError());
            return SX_OK;
        }
    }
    hipLaunchKernelGGL(k_gb_solve_loop, dim3(grid), dim3(GB_T2), gb_lds_bytes(kl + ku + GB_NB), s, h->ab, h->ldab, kl, ku, n, h->ipiv,
                       h->d_swaps, nrhs, X, ldx, trans ? 1 : 0, slow ? 1 : 0);
    SX_HIP(hipGetLastError());
#ifdef SX_GB_TICKS
    {
        unsigned long long t[16];
        SX_HIP(hipStreamSynchronize(s));
        SX_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_gb_t), sizeof(t)));
        const double np = static_cast<double>((n + GB_NB - 1) / GB_NB) * 100.0; // ticks of 10 ns -> us per panel
        fprintf(stderr, "[sx_bandlu] us per panel (cumulative over the calls so far / panels of this one): L loads %.2f triangle %.2f combine %.2f "
                        "stores+gap %.2f | U loads %.2f triangle %.2f combine %.2f stores+gap %.2f\n",
                t[0] / np, t[1] / np, t[2] / np, t[3] / np, t[4] / np, t[5] / np, t[6] / np, t[7] / np);
        fprintf(stderr, "[sx_bandlu] shader clock during the solve: %.0f MHz\n", t[9] ? 100.0 * static_cast<double>(t[8]) / static_cast<double>(t[9]) : 0.0);
        unsigned long long z[16] = {0};
        SX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gb_t), z, sizeof(z)));
    }
#endif
    return SX_OK;
}

// A x = b in place for nrhs right-hand sides with few entries each (see k_gb_solve_sparse): panels whose rows hold
// nothing above `tiny` in magnitude are skipped.  tiny = 0 gives the plain solve's result; the sparse crossover passes
// 1e-60 (46 orders below what its tableau drops).
SX_API int sx_bandlu_solve_sparse_dev(sx_bandlu *h, int64_t nrhs, double *X, int64_t ldx, double tiny) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(h->factored, "factor first");
    SX_REQUIRE(X && ldx >= h->n && nrhs >= 0 && tiny >= 0.0, "bad right-hand side block");
    if (nrhs == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    const int kl = h->kl, ku = h->ku;
    const int64_t n = h->n;
    const int64_t groups = (nrhs + GB_CB - 1) / GB_CB, npanel = (n + GB_NB - 1) / GB_NB;
    const size_t need = static_cast<size_t>(groups) * static_cast<size_t>(npanel);
    if (need > h->flags_cap) {
        SX_HIP(hipStreamSynchronize(s));
        (void)sx_dfree(h->d_flags);
        h->d_flags = nullptr;
        h->flags_cap = 0;
        SX_HIP(sx_dmalloc(&h->d_flags, need));
        h->flags_cap = need;
    }
    SX_HIP(hipMemsetAsync(h->d_flags, 0, need, s));
    hipLaunchKernelGGL(k_gb_flag, dim3(static_cast<unsigned>(std::min<int64_t>((n * nrhs + 255) / 256, 1 << 20))), dim3(256), 0, s, n, nrhs, X, ldx, tiny, npanel,
                       h->d_flags);
    hipLaunchKernelGGL(k_gb_solve_sparse, dim3(static_cast<unsigned>(groups)), dim3(GB_T2), gb_lds_bytes(kl + ku + GB_NB), s, h->ab, h->ldab,
                       kl, ku, n, h->ipiv, h->d_swaps, nrhs, X, ldx, tiny, h->d_flags);
    SX_HIP(hipGetLastError());
    return SX_OK;
}
