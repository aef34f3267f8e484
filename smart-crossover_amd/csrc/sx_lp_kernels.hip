// LP-side kernels of the perturbation crossover (gfx950): K1 column scoring, K2 row scoring,
// index-set selection, K3 cost perturbation, K10 pricing.  See include/sxhip.h for the contract
// and DESIGN.md for layout / roofline notes.  Built with -ffp-contract=off: every product and
// every sum below is a separately rounded binary64 operation.
#include "sx_internal.h"
#include "sx_rowblock.h"
#include "sx_segwalk.h"
#include "sx_runwalk.h"
#include "sx_slabs.h"

#include <cmath>

namespace {

struct StageDot {
    const double *__restrict__ vec;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const {
        o[0] = v * vec[i];
    }
};

// ------------------------------------------------------------------------------------- K1
template <int CHUNK, int NT>
__global__ __launch_bounds__(SX_WG) void k_score_columns(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle,
    const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
    const double *__restrict__ val, const double *__restrict__ y, const double *__restrict__ c,
    const double *__restrict__ x, const double *__restrict__ l, const double *__restrict__ u,
    double gamma, double *__restrict__ s_d, uint8_t *__restrict__ code, const double *__restrict__ carry_in) {
    __shared__ sx_walk_lds<1, CHUNK> lds;
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    double acc[1] = {0.0};
    int64_t j;
    bool valid;
    // the epilogue's operands are requested before the walk so their latency hides under it
    double cj = 0.0, xj = 0.0, lj = 0.0, uj = 0.0;
    auto pre = [&](int64_t seg, bool ok) {
        if (ok) {
            if (carry_in) acc[0] = carry_in[seg]; // last slab of a slabbed walk (sx_slabs.h): the sum continues
            cj = c[seg];
            if (code) {
                xj = x[seg];
                lj = l[seg];
                uj = u[seg];
            }
        }
    };
    sx_segwalk<1, CHUNK, NT>(tiles, tile, colptr, rowidx, val, StageDot{y}, lds, j, valid, acc, pre, true);
    if (!valid) return;
    const double sd = cj - acc[0];
    if (s_d) s_d[j] = sd;
    if (code) {
        const bool low = (xj - lj) < (gamma * sd);
        const bool up = (uj - xj) < (gamma * (-sd));
        code[j] = static_cast<uint8_t>((low ? SX_CODE_LOW : 0u) | (up ? SX_CODE_UP : 0u));
    }
}

// A pass of a slabbed walk that is not the last one (sx_slabs.h): carry[seg] (+0.0 in the first pass) continued by
// the slab's products, left to right.  Serves the column and the row walk alike.
template <int CHUNK, int NT>
__global__ __launch_bounds__(SX_WG) void k_slab_accumulate(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle, const int64_t *__restrict__ ptr,
    const int32_t *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ operand,
    int first, double *__restrict__ carry) {
    __shared__ sx_walk_lds<1, CHUNK> lds;
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    double acc[1] = {0.0};
    int64_t seg;
    bool valid;
    auto pre = [&](int64_t sg, bool ok) {
        if (ok && !first) acc[0] = carry[sg];
    };
    sx_segwalk<1, CHUNK, NT>(tiles, tile, ptr, idx, val, StageDot{operand}, lds, seg, valid, acc, pre, true);
    if (valid) carry[seg] = acc[0];
}

// K1 behind an LDS operand window: one workgroup scores RUN consecutive tiles per window load
// (sx_window.h).  Same sums, same roundings, same outputs as k_score_columns.
template <int RUN, int PF>
__global__ __launch_bounds__(SX_WG) void k_score_columns_lw(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle, const int32_t *__restrict__ win_lo,
    const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, const double *__restrict__ val,
    int64_t m, const double *__restrict__ y, const double *__restrict__ c, const double *__restrict__ x,
    const double *__restrict__ l, const double *__restrict__ u, double gamma, double *__restrict__ s_d,
    uint8_t *__restrict__ code) {
    __shared__ sx_walk_lds<1, SXL_CHUNK> lds;
    __shared__ double win[SXL_CAP];
    const int64_t nruns = (ntiles + RUN - 1) / RUN;
    const int64_t run = sx_tile_of_block(blockIdx.x, nruns, swizzle);
    if (run >= nruns) return;
    const int64_t t0 = run * RUN;
    const int64_t t1 = (t0 + RUN < ntiles) ? t0 + RUN : ntiles;
    const int64_t wlo = win_lo[(t0 + t1 - 1) >> 1]; // window of the run's middle tile
    sx_window_fill(win, y, wlo, m);
    __syncthreads();
    if constexpr (PF) { // loads one step ahead (sx_runwalk.h)
        struct Ops {
            double cj, xj, lj, uj;
        };
        auto pre = [&](int64_t seg) {
            Ops o{c[seg], 0.0, 0.0, 0.0};
            if (code) { // uniform
                o.xj = x[seg];
                o.lj = l[seg];
                o.uj = u[seg];
            }
            return o;
        };
        auto epi = [&](int64_t j, bool valid, double sum, const Ops &o) {
            if (!valid) return;
            const double sd = o.cj - sum;
            if (s_d) s_d[j] = sd;
            if (code) {
                const bool low = (o.xj - o.lj) < (gamma * sd);
                const bool up = (o.uj - o.xj) < (gamma * (-sd));
                code[j] = static_cast<uint8_t>((low ? SX_CODE_LOW : 0u) | (up ? SX_CODE_UP : 0u));
            }
        };
        sx_runwalk<RUN>(tiles, t0, t1, colptr, rowidx, val, sx_stage_win{y, win, wlo}, lds, pre, epi);
        return;
    }
    for (int64_t tile = t0; tile < t1; ++tile) {
        double acc[1];
        int64_t j;
        bool valid;
        double cj = 0.0, xj = 0.0, lj = 0.0, uj = 0.0;
        auto pre = [&](int64_t seg, bool ok) {
            if (ok) {
                cj = c[seg];
                if (code) {
                    xj = x[seg];
                    lj = l[seg];
                    uj = u[seg];
                }
            }
        };
        sx_segwalk<1, SXL_CHUNK, 0>(tiles, tile, colptr, rowidx, val, sx_stage_win{y, win, wlo}, lds, j, valid,
                                    acc, pre);
        if (valid) {
            const double sd = cj - acc[0];
            if (s_d) s_d[j] = sd;
            if (code) {
                const bool low = (xj - lj) < (gamma * sd);
                const bool up = (uj - xj) < (gamma * (-sd));
                code[j] = static_cast<uint8_t>((low ? SX_CODE_LOW : 0u) | (up ? SX_CODE_UP : 0u));
            }
        }
    }
}

// ------------------------------------------------------------------------------------- K2
template <int CHUNK, int NT>
__global__ __launch_bounds__(SX_WG) void k_score_rows(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle,
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ x, const double *__restrict__ b,
    const double *__restrict__ y, double gamma_dual, double *__restrict__ s_p,
    uint8_t *__restrict__ flag, const double *__restrict__ carry_in) {
    __shared__ sx_walk_lds<1, CHUNK> lds;
    const int64_t tile = sx_tile_of_block(blockIdx.x, ntiles, swizzle);
    if (tile >= ntiles) return;
    double acc[1] = {0.0};
    int64_t i;
    bool valid;
    double bi = 0.0, yi = 0.0;
    auto pre = [&](int64_t seg, bool ok) {
        if (ok) {
            if (carry_in) acc[0] = carry_in[seg]; // last slab of a slabbed walk (sx_slabs.h)
            bi = b[seg];
            if (flag) yi = y[seg];
        }
    };
    sx_segwalk<1, CHUNK, NT>(tiles, tile, rowptr, colidx, val, StageDot{x}, lds, i, valid, acc, pre, true);
    if (!valid) return;
    const double sp = bi - acc[0];
    if (s_p) s_p[i] = sp;
    if (flag) flag[i] = (sp < (gamma_dual * (-yi))) ? 1 : 0;
}

// ------------------------------------------------------------------------------------- K10
struct PricePartial {
    double min_rc;
    long long argmin;
    long long n_bad;
};

__device__ __forceinline__ void price_combine(double &v, long long &ix, double v2, long long ix2) {
    // lexicographic (value, index); indices < 0 mark "no candidate"
    if (ix2 >= 0 && (ix < 0 || v2 < v || (v2 == v && ix2 < ix))) {
        v = v2;
        ix = ix2;
    }
}

// `scratch`: 3 * SX_WG / 64 doubles of LDS that nobody else is using any more (the walk kernels hand over their
// product buffer after a barrier, so that the reduction does not add to the static LDS that bounds their occupancy)
__device__ __forceinline__ void price_block_reduce(double v, long long ix, long long bad,
                                                   PricePartial *out_slot, double *scratch) {
    double *sv = scratch;
    long long *si = reinterpret_cast<long long *>(scratch + SX_WG / 64);
    long long *sb = reinterpret_cast<long long *>(scratch + 2 * (SX_WG / 64));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double v2 = __shfl_down(v, o, 64);
        long long i2 = __shfl_down(ix, o, 64);
        price_combine(v, ix, v2, i2);
        bad += __shfl_down(bad, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wave] = v;
        si[wave] = ix;
        sb[wave] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SX_WG / 64; ++w) {
            price_combine(v, ix, sv[w], si[w]);
            bad += sb[w];
        }
        out_slot->min_rc = v;
        out_slot->argmin = ix;
        out_slot->n_bad = bad;
    }
}

// grid-stride over tiles: one partial per workgroup (at most PRICE_GRID of them)
constexpr int PRICE_GRID = 2048;

template <int CHUNK, int NT>
__global__ __launch_bounds__(SX_WG) void k_price(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle,
    const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
    const double *__restrict__ val, const double *__restrict__ y, const double *__restrict__ c,
    const int8_t *__restrict__ vbasis, double tol, double *__restrict__ rc_out,
    PricePartial *__restrict__ partial, const double *__restrict__ carry_in) {
    __shared__ sx_walk_lds<1, CHUNK> lds;
    double v = 0.0;
    long long ix = -1, bad = 0;
    // tile range and stride of this workgroup: with the XCD swizzle (gridDim.x is a multiple of 8)
    // XCD k = blockIdx % 8 walks the contiguous range [k*per, (k+1)*per) with its gridDim/8 blocks
    int64_t t = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
    if (swizzle) {
        const int64_t per = (ntiles + 7) >> 3;
        t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        t_end = ((blockIdx.x & 7) + 1) * per;
        if (t_end > ntiles) t_end = ntiles;
        t_step = gridDim.x >> 3;
    }
    for (; t < t_end; t += t_step) {
        double acc[1] = {0.0};
        int64_t j;
        bool valid;
        double cj = 0.0;
        int vbj = 0; // requested before the walk so that their latency hides under it
        auto pre = [&](int64_t seg, bool ok) {
            if (ok) {
                if (carry_in) acc[0] = carry_in[seg]; // last slab of a slabbed walk (sx_slabs.h)
                cj = c[seg];
                if (vbasis) vbj = vbasis[seg];
            }
        };
        sx_segwalk<1, CHUNK, NT>(tiles, t, colptr, rowidx, val, StageDot{y}, lds, j, valid, acc, pre, true);
        if (valid) {
            double rc = cj - acc[0];
            if (vbj == -2) rc = -rc;
            if (rc_out) rc_out[j] = rc;
            bad += (rc >= -tol) ? 0 : 1;
            if (rc == rc) price_combine(v, ix, rc, j); // NaN never becomes the minimum
        }
    }
    __syncthreads(); // the last tile's sums are out of the product buffer
    price_block_reduce(v, ix, bad, &partial[blockIdx.x], lds.v[0]);
}

// K10 behind the LDS operand window: grid-stride over runs of RUN tiles, one window load per run
template <int RUN, int PF>
__global__ __launch_bounds__(SX_WG, 4) void k_price_lw(
    const int64_t *__restrict__ tiles, int64_t ntiles, int swizzle, const int32_t *__restrict__ win_lo,
    const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx, const double *__restrict__ val,
    int64_t m, const double *__restrict__ y, const double *__restrict__ c, const int8_t *__restrict__ vbasis,
    double tol, double *__restrict__ rc_out, PricePartial *__restrict__ partial) {
    __shared__ sx_walk_lds<1, SXL_CHUNK> lds;
    __shared__ double win[SXL_CAP];
    double v = 0.0;
    long long ix = -1, bad = 0;
    const int64_t nruns = (ntiles + RUN - 1) / RUN;
    int64_t r = blockIdx.x, r_end = nruns, r_step = gridDim.x;
    if (swizzle) { // gridDim.x is a multiple of 8: XCD k walks the contiguous runs [k*per, (k+1)*per)
        const int64_t per = (nruns + 7) >> 3;
        r = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        r_end = ((blockIdx.x & 7) + 1) * per;
        if (r_end > nruns) r_end = nruns;
        r_step = gridDim.x >> 3;
    }
    for (; r < r_end; r += r_step) {
        const int64_t t0 = r * RUN;
        const int64_t t1 = (t0 + RUN < ntiles) ? t0 + RUN : ntiles;
        const int64_t wlo = win_lo[(t0 + t1 - 1) >> 1];
        __syncthreads(); // the previous run's gathers are done with the window
        sx_window_fill(win, y, wlo, m);
        __syncthreads();
        if constexpr (PF) { // loads one step ahead (sx_runwalk.h)
            struct Ops {
                double cj;
                int vbj;
            };
            auto pre = [&](int64_t seg) {
                Ops o{c[seg], 0};
                if (vbasis) o.vbj = vbasis[seg]; // uniform
                return o;
            };
            auto epi = [&](int64_t j, bool valid, double sum, const Ops &o) {
                if (!valid) return;
                double rc = o.cj - sum;
                if (o.vbj == -2) rc = -rc;
                if (rc_out) rc_out[j] = rc;
                bad += (rc >= -tol) ? 0 : 1;
                if (rc == rc) price_combine(v, ix, rc, j);
            };
            sx_runwalk<RUN>(tiles, t0, t1, colptr, rowidx, val, sx_stage_win{y, win, wlo}, lds, pre, epi);
            continue;
        }
        for (int64_t t = t0; t < t1; ++t) {
            double acc[1];
            int64_t j;
            bool valid;
            double cj = 0.0;
            int vbj = 0; // the epilogue's operands are requested before the walk (as in K1)
            auto pre = [&](int64_t seg, bool ok) {
                if (ok) {
                    cj = c[seg];
                    if (vbasis) vbj = vbasis[seg];
                }
            };
            sx_segwalk<1, SXL_CHUNK, 0>(tiles, t, colptr, rowidx, val, sx_stage_win{y, win, wlo}, lds, j, valid, acc,
                                        pre);
            if (valid) {
                double rc = cj - acc[0];
                if (vbj == -2) rc = -rc;
                if (rc_out) rc_out[j] = rc;
                bad += (rc >= -tol) ? 0 : 1;
                if (rc == rc) price_combine(v, ix, rc, j);
            }
        }
    }
    __syncthreads(); // the last tile's sums are out of the product buffer
    price_block_reduce(v, ix, bad, &partial[blockIdx.x], lds.v[0]);
}

__global__ __launch_bounds__(SX_WG) void k_price_final(const PricePartial *__restrict__ partial,
                                                       int64_t nblocks, sx_price_result *out) {
    double v = 0.0;
    long long ix = -1, bad = 0;
    for (int64_t b = threadIdx.x; b < nblocks; b += SX_WG) {
        price_combine(v, ix, partial[b].min_rc, partial[b].argmin);
        bad += partial[b].n_bad;
    }
    __shared__ PricePartial one;
    __shared__ double scratch[3 * (SX_WG / 64)];
    price_block_reduce(v, ix, bad, &one, scratch);
    __syncthreads();
    if (threadIdx.x == 0) {
        out->min_rc = (one.argmin >= 0) ? one.min_rc : NAN;
        out->argmin = one.argmin;
        out->n_violating = one.n_bad;
    }
}

// ------------------------------------------------------------------------------------- select
constexpr int SEL_PER_THREAD = 16;
constexpr int SEL_TILE = SX_WG * SEL_PER_THREAD; // 4096 flags per workgroup

__device__ __forceinline__ uint32_t sel_load_mask(const uint8_t *__restrict__ flags, int64_t n,
                                                  int64_t first, uint8_t mask) {
    // bit t of the result <=> (flags[first+t] & mask) != 0, t in [0,16)
    uint32_t bits = 0;
    if (first + SEL_PER_THREAD <= n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(flags + first);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((w[k] >> (8 * b)) & mask) bits |= 1u << (4 * k + b);
    } else {
        for (int t = 0; t < SEL_PER_THREAD; ++t)
            if (first + t < n && (flags[first + t] & mask)) bits |= 1u << t;
    }
    return bits;
}

__global__ __launch_bounds__(SX_WG) void k_select_count(const uint8_t *__restrict__ flags,
                                                        int64_t n, uint8_t mask,
                                                        int64_t *__restrict__ block_count) {
    const int64_t first = static_cast<int64_t>(blockIdx.x) * SEL_TILE +
                          static_cast<int64_t>(threadIdx.x) * SEL_PER_THREAD;
    long long cnt = (first < n) ? __popc(sel_load_mask(flags, n, first, mask)) : 0;
    cnt = sx_wave_sum(cnt);
    __shared__ long long s[SX_WG / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of block_count in place (single workgroup), total -> *count_out.  Every lane owns a
// run of consecutive counts (serial prefix), the run totals are scanned by wave shuffles and four wave
// totals: six barriers per 4096 counts instead of a Hillis-Steele ladder.
constexpr int SCAN_RUN = 16;
__global__ __launch_bounds__(SX_WG) void k_select_scan(int64_t *__restrict__ block_count,
                                                       int64_t nblocks,
                                                       int64_t *__restrict__ count_out) {
    __shared__ long long wtot[SX_WG / 64];
    __shared__ long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += SX_WG * SCAN_RUN) {
        const int64_t first = b0 + static_cast<int64_t>(threadIdx.x) * SCAN_RUN;
        long long v[SCAN_RUN];
        long long run = 0;
#pragma unroll
        for (int k = 0; k < SCAN_RUN; ++k) {
            v[k] = (first + k < nblocks) ? block_count[first + k] : 0;
            run += v[k];
        }
        long long incl = run; // inclusive scan of the run totals inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        long long before = carry + incl - run;
        for (int w = 0; w < wave; ++w) before += wtot[w];
#pragma unroll
        for (int k = 0; k < SCAN_RUN; ++k) {
            if (first + k < nblocks) block_count[first + k] = before;
            before += v[k];
        }
        __syncthreads();
        if (threadIdx.x == SX_WG - 1) carry = before; // end of the last lane's run = total so far
        __syncthreads();
    }
    if (threadIdx.x == 0) *count_out = carry;
}

// The selected positions of a tile are first collected in LDS as 16-bit offsets, in order, and then
// written out by all lanes with consecutive 8-byte stores (a lane writing its own run directly would
// scatter every store instruction over up to 64 cache lines).
__global__ __launch_bounds__(SX_WG) void k_select_write(const uint8_t *__restrict__ flags,
                                                        int64_t n, uint8_t mask,
                                                        const int64_t *__restrict__ block_off,
                                                        int64_t *__restrict__ idx_out) {
    __shared__ uint16_t off[SEL_TILE];
    __shared__ int wsum[SX_WG / 64];
    const int64_t tile0 = static_cast<int64_t>(blockIdx.x) * SEL_TILE;
    const int64_t first = tile0 + static_cast<int64_t>(threadIdx.x) * SEL_PER_THREAD;
    uint32_t bits = (first < n) ? sel_load_mask(flags, n, first, mask) : 0u;
    const int mine = __popc(bits);
    // exclusive prefix of `mine` across the workgroup: wave scan + wave offsets
    int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < SX_WG / 64; ++w) {
        if (w < wave) woff += wsum[w];
        total += wsum[w];
    }
    int k = woff + incl - mine;
    const int base = threadIdx.x * SEL_PER_THREAD;
    while (bits) {
        const int t = __ffs(bits) - 1;
        bits &= bits - 1;
        off[k++] = static_cast<uint16_t>(base + t);
    }
    __syncthreads();
    const int64_t dst = block_off[blockIdx.x];
    for (int q = threadIdx.x; q < total; q += SX_WG) idx_out[dst + q] = tile0 + off[q];
}

// ------------------------------------------------------------------------------------- K3
__device__ __forceinline__ double np_minimum(double a, double b) {
    return (a < b || a != a) ? a : b; // numpy.minimum: NaN wins
}

__global__ __launch_bounds__(SX_WG) void k_perturb_cost(int64_t n, const double *__restrict__ x,
                                                        const double *__restrict__ l,
                                                        const double *__restrict__ u,
                                                        const double *__restrict__ c,
                                                        const double *__restrict__ xi,
                                                        double scale_factor, int is_feas,
                                                        double *__restrict__ c_pt) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double cj = c[j], xij = xi[j];
        if (is_feas) {
            c_pt[j] = cj + xij;
            continue;
        }
        const double xj = x[j], lj = l[j], uj = u[j];
        const bool is_free = (lj == -INFINITY) && (uj == INFINITY);
        double xr = np_minimum(xj - lj, uj - xj);
        if (is_free) xr = xj;
        if (xr < 1e-6) xr = 1e-6;
        if (is_free) xr = 1.0;
        double p = np_minimum(xij / xr * scale_factor / 1e-2, 1e6);
        if (is_free) p = 0.0;
        c_pt[j] = cj + p;
    }
}

// x_real of perturb_c (lp_methods/algorithms.py:130-132 after :196-202): distance to the nearer
// bound, free columns keep x for the floor test, floor at 1e-6, free columns end at 1
__global__ __launch_bounds__(SX_WG) void k_x_real(int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ l,
                                                  const double *__restrict__ u, int apply_floor,
                                                  double *__restrict__ x_real) {
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * SX_WG) {
        const double xj = x[j], lj = l[j], uj = u[j];
        const bool is_free = (lj == -INFINITY) && (uj == INFINITY);
        double xr = np_minimum(xj - lj, uj - xj);
        if (is_free) xr = xj;
        if (apply_floor) {
            if (xr < 1e-6) xr = 1e-6;
            if (is_free) xr = 1.0;
        }
        x_real[j] = xr;
    }
}

// dst[i] = mask[i] ? src[i] : 0   (slack part of the standard-form vector scattered to all rows)
__global__ __launch_bounds__(SX_WG) void k_mask_f64(int64_t n, const double *__restrict__ src,
                                                    const uint8_t *__restrict__ mask,
                                                    double *__restrict__ dst) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * SX_WG)
        dst[i] = mask[i] ? src[i] : 0.0;
}

// grid of a one-tile-per-workgroup launch: with the XCD swizzle every XCD gets ceil(T/8) slots
inline unsigned walk_grid(const sx_ctx *ctx, int64_t ntiles) {
    if (!ctx->opt_xcd_swizzle) return static_cast<unsigned>(ntiles);
    return static_cast<unsigned>(((ntiles + 7) >> 3) << 3);
}

// passes 0 .. R-2 of a slabbed walk: S->carry holds the running sums the last pass starts from.  The entry stream is
// read once and non-temporal, so that it does not displace the operand slab from L2.
static int slab_passes(sx_ctx *ctx, const sx_slabs *S, const double *operand) {
    for (int s = 0; s + 1 < S->R; ++s) {
        const sx_slab &L = S->slab[s];
        hipLaunchKernelGGL((k_slab_accumulate<4096, 2>), dim3(walk_grid(ctx, L.ntiles)), dim3(SX_WG), 0, ctx->stream, L.tiles,
                           L.ntiles, ctx->opt_xcd_swizzle, L.ptr, L.idx, L.val, operand + L.off, s == 0 ? 1 : 0, S->carry);
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}

#define SX_DISPATCH_VARIANT(ctx, LAUNCH)                                                           \
    do {                                                                                           \
        if ((ctx)->opt_chunk == 2048) {                                                            \
            if ((ctx)->opt_nt_stream) LAUNCH(2048, 1);                                             \
            else LAUNCH(2048, 0);                                                                  \
        } else {                                                                                   \
            switch ((ctx)->opt_nt_stream) {                                                        \
            case 1: LAUNCH(4096, 1); break;                                                        \
            case 2: LAUNCH(4096, 2); break;                                                        \
            case 16: LAUNCH(4096, 16); break;                                                      \
            case 17: LAUNCH(4096, 17); break;                                                      \
            case 18: LAUNCH(4096, 18); break;                                                      \
            default: LAUNCH(4096, 0); break;                                                       \
            }                                                                                      \
        }                                                                                          \
    } while (0)

} // namespace

// ====================================================================================== API
// The LDS operand window of the column walk by MEASUREMENT (option "window" = -1, matrices of >= 2^22 entries, first K1 /
// K10 call): the sampling rule of sx_window.hip (half of the indices inside the window AND a median extent of >= 1,024 rows)
// was fitted to the lp_shard staircase, where a narrow window makes the plain walk the faster one (0.305 against 0.329 ms);
// on netlib_lp at config-5 size (extent ~100 rows, linking rows at the head) the plain walk takes 0.70 ms and the windowed
// one 0.35 (profiles/r04/experiments/k1_netlib_knobs.txt).  So the variants -- plain / windowed, with / without the
// XCD-contiguous tile map -- are timed once per matrix with the very kernel of the first K1 or K10 call, on its own operands
// (writing its outputs four times over), and the fastest is kept; results are bit-identical either way.  Not under capture.
template <class Launch> // launch(windowed, swizzle): enqueues the calling walk's kernel in that variant
int window_autotune(sx_ctx *ctx, const sx_matrix *A, Launch launch) {
    if (ctx->opt_window >= 0 || A->csc_win_tuned || A->nnz < (1 << 22) || A->n_csc_tiles < 64) return SX_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return SX_OK;
    }
    A->csc_win_tuned = 1;
    int run = 0;
    SX_TRY(sx_window_run_csc(ctx, A, &run)); // (builds the table)
    if (!A->csc_win_lo) return SX_OK;
    hipEvent_t ev[5];
    for (auto &e : ev) SX_HIP(hipEventCreate(&e));
    float t[4] = {0.f, 0.f, 0.f, 0.f}; // [XCD map on: plain, windowed; off: plain, windowed]
    const int nsw = ctx->opt_xcd_swizzle ? 2 : 1;
    for (int rep = 0; rep < 2; ++rep) { // (the first round warms all up)
        SX_HIP(hipEventRecord(ev[0], ctx->stream));
        for (int k = 0; k < 2 * nsw; ++k) {
            SX_TRY(launch(k & 1, (ctx->opt_xcd_swizzle && k < 2) ? 1 : 0));
            SX_HIP(hipEventRecord(ev[k + 1], ctx->stream));
        }
        SX_HIP(hipEventSynchronize(ev[2 * nsw]));
        for (int k = 0; k < 2 * nsw; ++k) SX_HIP(hipEventElapsedTime(&t[k], ev[k], ev[k + 1]));
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    SX_HIP(hipGetLastError());
    // the fastest of the (up to) four; the map in force and the rule's own choice win ties of less than 5 %
    int best = A->csc_win_useful ? 1 : 0;
    for (int k = 0; k < 2 * nsw; ++k)
        if (t[k] < 0.95f * t[best]) best = k;
    A->csc_win_useful = best & 1;
    A->csc_swizzle_off = (ctx->opt_xcd_swizzle && best >= 2) ? 1 : 0;
    if (getenv("SX_SPX_TRACE"))
        fprintf(stderr, "[sx_window] column walk timed on this matrix (ms): XCD map on: plain %.3f, windowed %.3f; off: plain %.3f, windowed %.3f -> %s, map %s\n", t[0], t[1],
                t[2], t[3], A->csc_win_useful ? "windowed" : "plain", A->csc_swizzle_off ? "off" : "on");
    return SX_OK;
}

SX_API int sx_score_columns_dev(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                                const double *x, const double *l, const double *u, double gamma,
                                double *s_d, uint8_t *code) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(A->ctx->device == ctx->device, "matrix lives on another device");
    SX_REQUIRE(A->csc_ptr != nullptr, "matrix has no CSC layout (row shard?)");
    SX_REQUIRE(y && c, "y or c is NULL");
    SX_REQUIRE(!code || (x && l && u), "code requested but x/l/u is NULL");
    if (A->n == 0) return SX_OK;
    int run = 0; // LDS operand window (sx_window.h): table built on first use; a large matrix is timed once, else the rule decides
    SX_TRY(window_autotune(ctx, A, [&](int windowed, int swz) -> int {
        if (windowed) {
            const int64_t nruns = (A->n_csc_tiles + 3) / 4;
            hipLaunchKernelGGL((k_score_columns_lw<4, 0>), dim3(walk_grid(ctx, nruns)), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, A->n_csc_tiles, swz,
                               A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, A->m, y, c, x, l, u, gamma, s_d, code);
        } else {
            hipLaunchKernelGGL((k_score_columns<4096, 0>), dim3(walk_grid(ctx, A->n_csc_tiles)), dim3(SX_WG), 0, ctx->stream, A->csc_tiles,
                               A->n_csc_tiles, swz, A->csc_ptr, A->csc_idx, A->csc_val, y, c, x, l, u, gamma, s_d, code,
                               static_cast<const double *>(nullptr));
        }
        return SX_OK;
    }));
    SX_TRY(sx_window_run_csc(ctx, A, &run));
    if (run) {
        const int swz = sx_csc_swizzle(ctx, A);
#define SX_LAUNCH_K1W(R)                                                                           \
    do {                                                                                           \
        const int64_t nruns = (A->n_csc_tiles + (R)-1) / (R);                                      \
        if (ctx->opt_run_prefetch)                                                                 \
            hipLaunchKernelGGL((k_score_columns_lw<R, 1>), dim3(walk_grid(ctx, nruns)), dim3(SX_WG), 0, \
                               ctx->stream, A->csc_tiles, A->n_csc_tiles, swz, A->csc_win_lo,      \
                               A->csc_ptr, A->csc_idx, A->csc_val, A->m, y, c, x, l, u, gamma, s_d, \
                               code);                                                              \
        else                                                                                       \
            hipLaunchKernelGGL((k_score_columns_lw<R, 0>), dim3(walk_grid(ctx, nruns)), dim3(SX_WG), 0, \
                               ctx->stream, A->csc_tiles, A->n_csc_tiles, swz, A->csc_win_lo,      \
                               A->csc_ptr, A->csc_idx, A->csc_val, A->m, y, c, x, l, u, gamma, s_d, \
                               code);                                                              \
    } while (0)
        if (run == 8) SX_LAUNCH_K1W(8);
        else if (run == 4) SX_LAUNCH_K1W(4);
        else if (run == 2) SX_LAUNCH_K1W(2);
        else SX_LAUNCH_K1W(1);
#undef SX_LAUNCH_K1W
        SX_HIP(hipGetLastError());
        return SX_OK;
    }
    {   // no locality and an operand beyond L2: slab after slab (sx_slabs.h), the epilogue in the last pass
        const sx_slabs *S = nullptr;
        SX_TRY(sx_slabs_get(ctx, A, 1, &S));
        if (S) {
            SX_TRY(slab_passes(ctx, S, y));
            const sx_slab &L = S->slab[S->R - 1];
            hipLaunchKernelGGL((k_score_columns<4096, 2>), dim3(walk_grid(ctx, L.ntiles)), dim3(SX_WG), 0, ctx->stream, L.tiles,
                               L.ntiles, ctx->opt_xcd_swizzle, L.ptr, L.idx, L.val, y + L.off, c, x, l, u, gamma, s_d, code,
                               S->carry);
            SX_HIP(hipGetLastError());
            return SX_OK;
        }
    }
    const unsigned grid = walk_grid(ctx, A->n_csc_tiles);
#define SX_LAUNCH_K1(CH, NTV)                                                                      \
    hipLaunchKernelGGL((k_score_columns<CH, NTV>), dim3(grid), dim3(SX_WG), 0, ctx->stream,        \
                       A->csc_tiles, A->n_csc_tiles, sx_csc_swizzle(ctx, A), A->csc_ptr, A->csc_idx, \
                       A->csc_val, y, c, x, l, u, gamma, s_d, code, static_cast<const double *>(nullptr))
    SX_DISPATCH_VARIANT(ctx, SX_LAUNCH_K1);
#undef SX_LAUNCH_K1
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_score_rows_dev(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *b,
                             const double *y, double gamma_dual, double *s_p, uint8_t *flag) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(A->ctx->device == ctx->device, "matrix lives on another device");
    SX_REQUIRE(A->csr_ptr != nullptr, "matrix has no CSR layout (column shard?)");
    SX_REQUIRE(x && b, "x or b is NULL");
    SX_REQUIRE(!flag || y, "flag requested but y is NULL");
    if (A->m == 0) return SX_OK;
    {   // column-blocked copy of the rows, when the matrix has (or is due) one: same bits, fewer L2 requests
        const sx_rowblock *rb = nullptr;
        SX_TRY(sx_rowblock_get(ctx, A, &rb));
        if (rb) return sx_rb_score_rows(ctx, rb, A->n, x, b, y, gamma_dual, s_p, flag);
    }
    {   // no locality and an operand beyond L2: slab after slab (sx_slabs.h)
        const sx_slabs *S = nullptr;
        SX_TRY(sx_slabs_get(ctx, A, 0, &S));
        if (S) {
            SX_TRY(slab_passes(ctx, S, x));
            const sx_slab &L = S->slab[S->R - 1];
            hipLaunchKernelGGL((k_score_rows<4096, 2>), dim3(walk_grid(ctx, L.ntiles)), dim3(SX_WG), 0, ctx->stream, L.tiles,
                               L.ntiles, ctx->opt_xcd_swizzle, L.ptr, L.idx, L.val, x + L.off, b, y, gamma_dual, s_p, flag,
                               S->carry);
            SX_HIP(hipGetLastError());
            return SX_OK;
        }
    }
    // (linking rows that sit together would all queue on one XCD under the contiguous map: sx_build_tiles)
    const int swz2 = (ctx->opt_xcd_swizzle && A->csr_imbalance <= SX_SWIZZLE_MAX_IMBALANCE) ? 1 : 0;
    const unsigned grid = swz2 ? walk_grid(ctx, A->n_csr_tiles) : static_cast<unsigned>(A->n_csr_tiles);
#define SX_LAUNCH_K2(CH, NTV)                                                                      \
    hipLaunchKernelGGL((k_score_rows<CH, NTV>), dim3(grid), dim3(SX_WG), 0, ctx->stream,           \
                       A->csr_tiles, A->n_csr_tiles, swz2, A->csr_ptr, A->csr_idx,                 \
                       A->csr_val, x, b, y, gamma_dual, s_p, flag, static_cast<const double *>(nullptr))
    SX_DISPATCH_VARIANT(ctx, SX_LAUNCH_K2);
#undef SX_LAUNCH_K2
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_select_indices_dev(sx_ctx *ctx, int64_t n, const uint8_t *flags, uint8_t mask,
                                 int64_t *idx_out, int64_t *count_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    SX_REQUIRE(count_out != nullptr, "count_out is NULL");
    if (n == 0) {
        SX_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), ctx->stream));
        return SX_OK;
    }
    SX_REQUIRE(flags && idx_out, "flags or idx_out is NULL");
    SX_REQUIRE((reinterpret_cast<uintptr_t>(flags) & 15) == 0, "flags must be 16-byte aligned");
    const int64_t nb = (n + SEL_TILE - 1) / SEL_TILE;
    SX_TRY(sx_reserve(ctx, static_cast<size_t>(nb) * sizeof(int64_t)));
    int64_t *bc = static_cast<int64_t *>(ctx->ws);
    hipLaunchKernelGGL(k_select_count, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream,
                       flags, n, mask, bc);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(SX_WG), 0, ctx->stream, bc, nb, count_out);
    hipLaunchKernelGGL(k_select_write, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream,
                       flags, n, mask, bc, idx_out);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_perturb_cost_dev(sx_ctx *ctx, int64_t n, const double *x, const double *l,
                               const double *u, const double *c, const double *xi,
                               double scale_factor, int is_feas, double *c_pt) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return SX_OK;
    SX_REQUIRE(c && xi && c_pt, "c, xi or c_pt is NULL");
    SX_REQUIRE(is_feas || (x && l && u), "x, l or u is NULL");
    int64_t nb = (n + SX_WG - 1) / SX_WG;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_perturb_cost, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream,
                       n, x, l, u, c, xi, scale_factor, is_feas, c_pt);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_x_real_dev(sx_ctx *ctx, int64_t n, const double *x, const double *l, const double *u,
                         int apply_floor, double *x_real) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return SX_OK;
    SX_REQUIRE(x && l && u && x_real, "NULL argument");
    int64_t nb = (n + SX_WG - 1) / SX_WG;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_x_real, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream, n, x, l, u,
                       apply_floor, x_real);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_mask_f64_dev(sx_ctx *ctx, int64_t n, const double *src, const uint8_t *mask, double *dst) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return SX_OK;
    SX_REQUIRE(src && mask && dst, "NULL argument");
    int64_t nb = (n + SX_WG - 1) / SX_WG;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_mask_f64, dim3(static_cast<unsigned>(nb)), dim3(SX_WG), 0, ctx->stream, n, src, mask, dst);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

SX_API int sx_price_dev(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                        const int8_t *vbasis, double tol, double *rc, sx_price_result *result_dev) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(A->ctx->device == ctx->device, "matrix lives on another device");
    SX_REQUIRE(A->csc_ptr != nullptr, "matrix has no CSC layout (row shard?)");
    SX_REQUIRE(y && c && result_dev, "y, c or result is NULL");
    // the swizzled walk needs a grid that is a multiple of 8 (one slice per XCD)
    SX_TRY(window_autotune(ctx, A, [&](int windowed, int swz_) -> int {
        int nb_ = static_cast<int>(A->n_csc_tiles < PRICE_GRID ? A->n_csc_tiles : PRICE_GRID) & ~7;
        if (nb_ < 8) return SX_OK;
        SX_TRY(sx_reserve(ctx, static_cast<size_t>(nb_) * sizeof(PricePartial)));
        PricePartial *part_ = static_cast<PricePartial *>(ctx->ws);
        if (windowed)
            hipLaunchKernelGGL((k_price_lw<4, 0>), dim3(nb_), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, A->n_csc_tiles, swz_, A->csc_win_lo, A->csc_ptr,
                               A->csc_idx, A->csc_val, A->m, y, c, vbasis, tol, rc, part_);
        else
            hipLaunchKernelGGL((k_price<4096, 0>), dim3(nb_), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, A->n_csc_tiles, swz_, A->csc_ptr, A->csc_idx,
                               A->csc_val, y, c, vbasis, tol, rc, part_, static_cast<const double *>(nullptr));
        return SX_OK;
    }));
    const int swz = sx_csc_swizzle(ctx, A);
    int nb = static_cast<int>(A->n_csc_tiles < PRICE_GRID ? A->n_csc_tiles : PRICE_GRID);
    if (swz) nb &= ~7;
    if (nb == 0) {
        sx_price_result empty = {NAN, -1, 0};
        SX_HIP(hipMemcpyAsync(result_dev, &empty, sizeof(empty), hipMemcpyHostToDevice, ctx->stream));
        SX_HIP(hipStreamSynchronize(ctx->stream));
        return SX_OK;
    }
    SX_TRY(sx_reserve(ctx, static_cast<size_t>(nb) * sizeof(PricePartial)));
    PricePartial *partial = static_cast<PricePartial *>(ctx->ws);
    int run = 0;
    SX_TRY(sx_window_run_csc(ctx, A, &run));
    if (run) {
#define SX_LAUNCH_K10W(R)                                                                          \
    do {                                                                                           \
        if (ctx->opt_run_prefetch)                                                                 \
            hipLaunchKernelGGL((k_price_lw<R, 1>), dim3(nb), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, \
                               A->n_csc_tiles, swz, A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, \
                               A->m, y, c, vbasis, tol, rc, partial);                              \
        else                                                                                       \
            hipLaunchKernelGGL((k_price_lw<R, 0>), dim3(nb), dim3(SX_WG), 0, ctx->stream, A->csc_tiles, \
                               A->n_csc_tiles, swz, A->csc_win_lo, A->csc_ptr, A->csc_idx, A->csc_val, \
                               A->m, y, c, vbasis, tol, rc, partial);                              \
    } while (0)
        if (run == 8) SX_LAUNCH_K10W(8);
        else if (run == 4) SX_LAUNCH_K10W(4);
        else if (run == 2) SX_LAUNCH_K10W(2);
        else SX_LAUNCH_K10W(1);
#undef SX_LAUNCH_K10W
        hipLaunchKernelGGL(k_price_final, dim3(1), dim3(SX_WG), 0, ctx->stream, partial,
                           static_cast<int64_t>(nb), result_dev);
        SX_HIP(hipGetLastError());
        return SX_OK;
    }
    {   // no locality and an operand beyond L2: slab after slab (sx_slabs.h), pricing in the last pass
        const sx_slabs *S = nullptr;
        SX_TRY(sx_slabs_get(ctx, A, 1, &S));
        // (a first call builds the slabs with scans in the context's workspace, which may have moved it)
        SX_TRY(sx_reserve(ctx, static_cast<size_t>(nb > 0 ? nb : 1) * sizeof(PricePartial)));
        partial = static_cast<PricePartial *>(ctx->ws);
        if (S) {
            SX_TRY(slab_passes(ctx, S, y));
            const sx_slab &L = S->slab[S->R - 1];
            const int swl = (ctx->opt_xcd_swizzle && L.ntiles >= 64) ? 1 : 0;
            int nbl = static_cast<int>(L.ntiles < PRICE_GRID ? L.ntiles : PRICE_GRID);
            if (swl) nbl &= ~7;
            if (nbl < 1) nbl = 1;
            SX_TRY(sx_reserve(ctx, static_cast<size_t>(nbl) * sizeof(PricePartial)));
            partial = static_cast<PricePartial *>(ctx->ws);
            hipLaunchKernelGGL((k_price<4096, 2>), dim3(nbl), dim3(SX_WG), 0, ctx->stream, L.tiles, L.ntiles, swl, L.ptr, L.idx,
                               L.val, y + L.off, c, vbasis, tol, rc, partial, S->carry);
            hipLaunchKernelGGL(k_price_final, dim3(1), dim3(SX_WG), 0, ctx->stream, partial, static_cast<int64_t>(nbl), result_dev);
            SX_HIP(hipGetLastError());
            return SX_OK;
        }
    }
#define SX_LAUNCH_K10(CH, NTV)                                                                     \
    hipLaunchKernelGGL((k_price<CH, NTV>), dim3(nb), dim3(SX_WG), 0, ctx->stream, A->csc_tiles,    \
                       A->n_csc_tiles, swz, A->csc_ptr, A->csc_idx, A->csc_val, y, c, vbasis, tol, \
                       rc, partial, static_cast<const double *>(nullptr))
    SX_DISPATCH_VARIANT(ctx, SX_LAUNCH_K10);
#undef SX_LAUNCH_K10
    hipLaunchKernelGGL(k_price_final, dim3(1), dim3(SX_WG), 0, ctx->stream, partial,
                       static_cast<int64_t>(nb), result_dev);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

// ------------------------------------------------------------------ host-pointer wrappers
SX_API int sx_score_columns(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                            const double *x, const double *l, const double *u, double gamma,
                            double *s_d, uint8_t *code) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(y && c, "y or c is NULL");
    SX_REQUIRE(!code || (x && l && u), "code requested but x/l/u is NULL");
    const size_t nb = sizeof(double) * A->n, mb = sizeof(double) * A->m;
    sx_stage st(ctx);
    void *dy, *dc, *dx = nullptr, *dl = nullptr, *du = nullptr, *dsd = nullptr, *dcode = nullptr;
    SX_TRY(st.in(y, mb, &dy));
    SX_TRY(st.in(c, nb, &dc));
    if (code) {
        SX_TRY(st.in(x, nb, &dx));
        SX_TRY(st.in(l, nb, &dl));
        SX_TRY(st.in(u, nb, &du));
        SX_TRY(st.in(nullptr, A->n, &dcode));
    }
    if (s_d) SX_TRY(st.in(nullptr, nb, &dsd));
    SX_TRY(sx_score_columns_dev(ctx, A, (double *)dy, (double *)dc, (double *)dx, (double *)dl,
                                (double *)du, gamma, (double *)dsd, (uint8_t *)dcode));
    SX_TRY(st.out(s_d, dsd, nb));
    SX_TRY(st.out(code, dcode, A->n));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_score_rows(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *b,
                         const double *y, double gamma_dual, double *s_p, uint8_t *flag) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(x && b, "x or b is NULL");
    SX_REQUIRE(!flag || y, "flag requested but y is NULL");
    const size_t nb = sizeof(double) * A->n, mb = sizeof(double) * A->m;
    sx_stage st(ctx);
    void *dx, *db, *dy = nullptr, *dsp = nullptr, *dflag = nullptr;
    SX_TRY(st.in(x, nb, &dx));
    SX_TRY(st.in(b, mb, &db));
    if (flag) {
        SX_TRY(st.in(y, mb, &dy));
        SX_TRY(st.in(nullptr, A->m, &dflag));
    }
    if (s_p) SX_TRY(st.in(nullptr, mb, &dsp));
    SX_TRY(sx_score_rows_dev(ctx, A, (double *)dx, (double *)db, (double *)dy, gamma_dual,
                             (double *)dsp, (uint8_t *)dflag));
    SX_TRY(st.out(s_p, dsp, mb));
    SX_TRY(st.out(flag, dflag, A->m));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_select_indices(sx_ctx *ctx, int64_t n, const uint8_t *flags, uint8_t mask,
                             int64_t *idx_out, int64_t *count_out) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0 && count_out, "bad arguments");
    SX_REQUIRE(n == 0 || (flags && idx_out), "flags or idx_out is NULL");
    sx_stage st(ctx);
    void *df, *didx, *dcount;
    SX_TRY(st.in(flags, static_cast<size_t>(n), &df));
    SX_TRY(st.in(nullptr, sizeof(int64_t) * static_cast<size_t>(n), &didx));
    SX_TRY(st.in(nullptr, sizeof(int64_t), &dcount));
    SX_TRY(sx_select_indices_dev(ctx, n, (uint8_t *)df, mask, (int64_t *)didx, (int64_t *)dcount));
    SX_TRY(st.out(count_out, dcount, sizeof(int64_t)));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    SX_TRY(st.out(idx_out, didx, sizeof(int64_t) * static_cast<size_t>(*count_out)));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_perturb_cost(sx_ctx *ctx, int64_t n, const double *x, const double *l,
                           const double *u, const double *c, const double *xi, double scale_factor,
                           int is_feas, double *c_pt) {
    SX_ENTER(ctx);
    SX_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return SX_OK;
    SX_REQUIRE(c && xi && c_pt, "c, xi or c_pt is NULL");
    SX_REQUIRE(is_feas || (x && l && u), "x, l or u is NULL");
    const size_t nb = sizeof(double) * static_cast<size_t>(n);
    sx_stage st(ctx);
    void *dx = nullptr, *dl = nullptr, *du = nullptr, *dc, *dxi, *dout;
    if (!is_feas) {
        SX_TRY(st.in(x, nb, &dx));
        SX_TRY(st.in(l, nb, &dl));
        SX_TRY(st.in(u, nb, &du));
    }
    SX_TRY(st.in(c, nb, &dc));
    SX_TRY(st.in(xi, nb, &dxi));
    SX_TRY(st.in(nullptr, nb, &dout));
    SX_TRY(sx_perturb_cost_dev(ctx, n, (double *)dx, (double *)dl, (double *)du, (double *)dc,
                               (double *)dxi, scale_factor, is_feas, (double *)dout));
    SX_TRY(st.out(c_pt, dout, nb));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}

SX_API int sx_price(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                    const int8_t *vbasis, double tol, double *rc, sx_price_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A != nullptr, "matrix is NULL");
    SX_REQUIRE(y && c && result, "y, c or result is NULL");
    const size_t nb = sizeof(double) * A->n, mb = sizeof(double) * A->m;
    sx_stage st(ctx);
    void *dy, *dc, *dvb = nullptr, *drc = nullptr, *dres;
    SX_TRY(st.in(y, mb, &dy));
    SX_TRY(st.in(c, nb, &dc));
    if (vbasis) SX_TRY(st.in(vbasis, A->n, &dvb));
    if (rc) SX_TRY(st.in(nullptr, nb, &drc));
    SX_TRY(st.in(nullptr, sizeof(sx_price_result), &dres));
    SX_TRY(sx_price_dev(ctx, A, (double *)dy, (double *)dc, (int8_t *)dvb, tol, (double *)drc,
                        (sx_price_result *)dres));
    SX_TRY(st.out(rc, drc, nb));
    SX_TRY(st.out(result, dres, sizeof(sx_price_result)));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    return SX_OK;
}
