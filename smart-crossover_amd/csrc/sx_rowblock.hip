// Row walk over the column-blocked row layout (sx_rowblock.h): K2 score_rows
// (reference formats.py:74-76 + lp_methods/algorithms.py:106) and the CSR product of the projector CG
// (lp_methods/algorithms.py:183-187).  Same sums, same roundings as the plain walk of sx_segwalk.h;
// built with -ffp-contract=off.
//
// One workgroup of RB_TW lanes owns a super-tile (RB_RPL rows per lane: the wider the super-tile, the fewer
// times an x window is loaded) and steps through its chunks.  Step k:
//   issue    if chunk k+1 opens a new cell: its row starts and its x window (registers); then the entries of
//            chunk k+1 (registers) -- the youngest loads, which the waits of this step leave in flight
//   stage    rounded products of chunk k -> LDS; the operand comes from the LDS window of the cell, or from
//            global memory for a direct cell
//   consume  lane t adds the products of row t that lie in this chunk, left to right
//   publish  window of chunk k+1 -> LDS (behind the stage barrier: every gather of chunk k is done)
// Loads are raw buffer loads: descriptor in scalar registers, lanes past the end read zeros, no clamps.
// The entry stream is read once and marked non-temporal so that it does not evict the x windows, which
// neighbouring super-tiles re-read, from the XCD's L2.
// MI355X, config 5 (1e6 x 1e7, 8e7 entries): 0.36 ms against 0.61 ms for the plain walk
// (profiles/r02/rb_bench_*.txt), L1<->L2 requests 8.45e7 -> ~2e7.
#include "sx_internal.h"
#include "sx_rowblock.h"
#include "sx_segwalk.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int RB_NQ = RB_CHUNK / (RB_TW * 4);   // 16-byte index loads per lane and chunk
constexpr int RB_NW = RB_CWIN / (RB_TW * 2);    // 16-byte window pieces per lane
constexpr int RB_NT = 2;                        // cache policy of the entry stream: non-temporal
constexpr int RB_MINW = 6;                      // waves per SIMD the register budget allows (3 workgroups / CU)
static_assert(RB_CHUNK % (RB_TW * 4) == 0 && RB_CWIN % (RB_TW * 2) == 0, "chunk / window vs workgroup");

struct RbEntries { // the entries one lane stages of one chunk
    sx_v4i i[RB_NQ];
    sx_v2d v01[RB_NQ], v23[RB_NQ];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rb_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, static_cast<int>(bytes), 0x00020000);
}

__device__ __forceinline__ void rb_load(RbEntries &E, const sx_rb_chunk &c, const int32_t *__restrict__ idx,
                                        const double *__restrict__ val, const double *__restrict__ lprod) {
    const uint32_t ne4 = static_cast<uint32_t>((c.ne + 3) & ~3);
    if (c.staged && lprod) { // the chunk's products are there already: 8 bytes per entry, no gather
        const __amdgpu_buffer_rsrc_t rp = rb_rsrc(lprod + c.e0, ne4 * 8u);
        const int tid = threadIdx.x;
#pragma unroll
        for (int q = 0; q < RB_NQ; ++q) {
            E.v01[q] = __builtin_bit_cast(sx_v2d, __builtin_amdgcn_raw_buffer_load_b128(rp, tid * 32, q * RB_TW * 32, RB_NT));
            E.v23[q] = __builtin_bit_cast(sx_v2d, __builtin_amdgcn_raw_buffer_load_b128(rp, tid * 32 + 16, q * RB_TW * 32, RB_NT));
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t ri = rb_rsrc(idx + c.e0, ne4 * 4u), rv = rb_rsrc(val + c.e0, ne4 * 8u);
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < RB_NQ; ++q) {
        E.i[q] = __builtin_amdgcn_raw_buffer_load_b128(ri, tid * 16, q * RB_TW * 16, RB_NT);
        E.v01[q] = __builtin_bit_cast(sx_v2d, __builtin_amdgcn_raw_buffer_load_b128(rv, tid * 32, q * RB_TW * 32, RB_NT));
        E.v23[q] = __builtin_bit_cast(sx_v2d, __builtin_amdgcn_raw_buffer_load_b128(rv, tid * 32 + 16, q * RB_TW * 32, RB_NT));
    }
}

// lanes past the chunk hold zeros (index 0, value 0.0): their products land in slots no row segment covers
__device__ __forceinline__ void rb_stage(const RbEntries &E, const sx_rb_chunk &c, const double *win,
                                         const double *__restrict__ x, double *prod, bool staged) {
    const int tid = threadIdx.x;
    if (staged) {
#pragma unroll
        for (int q = 0; q < RB_NQ; ++q) {
            const int off = q * RB_TW * 4 + tid * 4;
            double2 *dst = reinterpret_cast<double2 *>(prod + off);
            dst[0] = make_double2(E.v01[q].x, E.v01[q].y);
            dst[1] = make_double2(E.v23[q].x, E.v23[q].y);
        }
    } else if (c.col0 != RB_NO_WINDOW) {
        const double *w0 = win - c.col0; // only dereferenced inside [win, win + RB_CWIN)
#pragma unroll
        for (int q = 0; q < RB_NQ; ++q) {
            const int off = q * RB_TW * 4 + tid * 4;
            const bool live = off < c.ne; // the gap behind the cell holds (col0, 0.0): in range as well
            const int i0 = live ? E.i[q].x : c.col0, i1 = live ? E.i[q].y : c.col0;
            const int i2 = live ? E.i[q].z : c.col0, i3 = live ? E.i[q].w : c.col0;
            double2 *dst = reinterpret_cast<double2 *>(prod + off);
            dst[0] = make_double2(E.v01[q].x * w0[i0], E.v01[q].y * w0[i1]);
            dst[1] = make_double2(E.v23[q].x * w0[i2], E.v23[q].y * w0[i3]);
        }
    } else {
        double xv[RB_NQ][4];
#pragma unroll
        for (int q = 0; q < RB_NQ; ++q) {
            xv[q][0] = x[E.i[q].x];
            xv[q][1] = x[E.i[q].y];
            xv[q][2] = x[E.i[q].z];
            xv[q][3] = x[E.i[q].w];
        }
#pragma unroll
        for (int q = 0; q < RB_NQ; ++q) {
            const int off = q * RB_TW * 4 + tid * 4;
            double2 *dst = reinterpret_cast<double2 *>(prod + off);
            dst[0] = make_double2(E.v01[q].x * xv[q][0], E.v01[q].y * xv[q][1]);
            dst[1] = make_double2(E.v23[q].x * xv[q][2], E.v23[q].y * xv[q][3]);
        }
    }
}

// one row per lane: the batched loop for long segments, then at most two rounds of four with the adds
// selected by the remaining count (prod[] has 8 slack slots behind the chunk, so the reads need no clamp)
__device__ __forceinline__ void rb_consume(double &acc, int rs0, int rs1, int base, int ne, const double *prod) {
    const int k0 = rs0 > base ? rs0 : base;
    const int k1 = rs1 < base + ne ? rs1 : base + ne;
    int o = k0 - base;
    int left = k1 - k0;
    double a = acc;
    for (; left >= 8; left -= 8, o += 8) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = prod[o + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) a = a + t[q];
    }
    if (__builtin_amdgcn_ballot_w64(left > 0)) {
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = prod[o + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double s2 = a + t[q];
                a = (q < left) ? s2 : a;
            }
            o += 4;
            left -= 4;
            if (round == 0 && !__builtin_amdgcn_ballot_w64(left > 0)) break;
        }
    }
    acc = a;
}

// chunk records of the super-tile are read through LDS, RB_TAB at a time (+ the look-ahead of the last one)
constexpr int RB_TAB = 64;
constexpr int RB_TABN = RB_TAB + 2;

struct RbLds {
    double win[RB_CWIN];
    double prod[RB_CHUNK + 8];
    int tab[RB_TABN * 8];
};

__device__ __forceinline__ sx_rb_chunk rb_record(const int *tab, int slot) {
    const int *p = tab + slot * 8;
    sx_rb_chunk c;
    const int lo = __builtin_amdgcn_readfirstlane(p[0]), hi = __builtin_amdgcn_readfirstlane(p[1]);
    c.e0 = (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
    c.ne = __builtin_amdgcn_readfirstlane(p[2]);
    c.col0 = __builtin_amdgcn_readfirstlane(p[3]);
    c.cell = __builtin_amdgcn_readfirstlane(p[4]);
    c.base = __builtin_amdgcn_readfirstlane(p[5]);
    c.fresh = __builtin_amdgcn_readfirstlane(p[6]);
    c.staged = __builtin_amdgcn_readfirstlane(p[7]);
    return c;
}

struct RbLayout {
    const sx_rb_supertile *__restrict__ st;
    const sx_rb_chunk *__restrict__ chunks;
    const uint16_t *__restrict__ rowstart;
    const int32_t *__restrict__ idx;
    const double *__restrict__ val;
    int64_t nst;
    const double *__restrict__ lprod; // products of the staged (long-row) chunks, or nullptr
};

// acc[r] = sum of row S.row0 + tid + r * RB_TW over the super-tile (0 for rows beyond its end); all lanes of the
// workgroup call it
__device__ __forceinline__ void rb_supertile_sum(const RbLayout &L, const sx_rb_supertile &S, const double *__restrict__ x,
                                                 int64_t ncols, RbLds &lds, double (&acc)[RB_RPL]) {
    const int tid = threadIdx.x;
    const int *ck = reinterpret_cast<const int *>(L.chunks + S.chunk0);
    const int n = S.nchunks;
#pragma unroll
    for (int r = 0; r < RB_RPL; ++r) acc[r] = 0.0;
    if (n <= 0) return;
    int g = 0;
    auto load_tab = [&]() {
        for (int i = tid; i < RB_TABN * 8; i += RB_TW) {
            int rec = g + i / 8;
            rec = rec < n ? rec : n - 1;
            lds.tab[i] = ck[static_cast<int64_t>(rec) * 8 + (i & 7)];
        }
    };
    auto load_rs = [&](int (&rs)[RB_RPL][2], const sx_rb_chunk &c) {
        const __amdgpu_buffer_rsrc_t rr = rb_rsrc(L.rowstart + static_cast<int64_t>(c.cell) * RB_RS_STRIDE, RB_RS_STRIDE * 2u);
#pragma unroll
        for (int r = 0; r < RB_RPL; ++r) {
            const int row = tid + r * RB_TW;
            const int k0 = row < S.nrows ? row : S.nrows, k1 = row + 1 < S.nrows ? row + 1 : S.nrows;
            rs[r][0] = static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rr, k0 * 2, 0, 0));
            rs[r][1] = static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rr, k1 * 2, 0, 0));
        }
    };
    auto load_win = [&](sx_v2d (&wreg)[RB_NW], const sx_rb_chunk &c) {
        const int64_t room = ncols - c.col0;
        const __amdgpu_buffer_rsrc_t rw = rb_rsrc(x + c.col0, static_cast<uint32_t>(room < RB_CWIN ? room : RB_CWIN) * 8u);
#pragma unroll
        for (int w = 0; w < RB_NW; ++w)
            wreg[w] = __builtin_bit_cast(sx_v2d, __builtin_amdgcn_raw_buffer_load_b128(rw, tid * 16, w * RB_TW * 16, 0));
    };
    auto store_win = [&](const sx_v2d (&wreg)[RB_NW]) {
#pragma unroll
        for (int w = 0; w < RB_NW; ++w) *reinterpret_cast<sx_v2d *>(lds.win + w * RB_TW * 2 + tid * 2) = wreg[w];
    };
    int rs[RB_RPL][2], rs_next[RB_RPL][2];
    RbEntries E[2];
    sx_v2d wreg[RB_NW];
    __syncthreads(); // the previous super-tile of a grid-stride caller is done with the LDS
    load_tab();
    __syncthreads();
    sx_rb_chunk c = rb_record(lds.tab, 0);
    load_rs(rs, c);
    if (c.col0 != RB_NO_WINDOW) load_win(wreg, c);
    rb_load(E[0], c, L.idx, L.val, L.lprod);
    if (c.col0 != RB_NO_WINDOW) store_win(wreg);
    __syncthreads();
    auto step = [&](RbEntries &Ecur, RbEntries &Enext, int k) {
        const bool has_next = k + 1 < n;
        const sx_rb_chunk cn = rb_record(lds.tab, (has_next ? k + 1 : k) - g);
        const bool open = has_next && cn.fresh;
        const bool new_win = open && cn.col0 != RB_NO_WINDOW;
        if (open) load_rs(rs_next, cn);
        if (new_win) load_win(wreg, cn);
        if (has_next) rb_load(Enext, cn, L.idx, L.val, L.lprod);
        rb_stage(Ecur, c, lds.win, x, lds.prod, c.staged && L.lprod);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RB_RPL; ++r) rb_consume(acc[r], rs[r][0], rs[r][1], c.base, c.ne, lds.prod);
        if (new_win) store_win(wreg);
        if (open) {
#pragma unroll
            for (int r = 0; r < RB_RPL; ++r) {
                rs[r][0] = rs_next[r][0];
                rs[r][1] = rs_next[r][1];
            }
        }
        if (has_next && k + 1 == g + RB_TAB) { // next group of records
            __syncthreads();
            g += RB_TAB;
            load_tab();
        }
        __syncthreads();
        c = cn;
    };
    for (int k = 0; k < n; k += 2) {
        step(E[0], E[1], k);
        if (k + 1 < n) step(E[1], E[0], k + 1);
    }
}

// ------------------------------------------------------------------------------------- K2
__global__ __launch_bounds__(RB_TW, RB_MINW) void k_rb_score_rows(RbLayout L, int swizzle, const int32_t *__restrict__ order,
                                                                  const double *__restrict__ x,
                                                                  int64_t ncols, const double *__restrict__ b,
                                                                  const double *__restrict__ y, double gamma_dual,
                                                                  double *__restrict__ s_p, uint8_t *__restrict__ flag) {
    __shared__ RbLds lds;
    const int64_t tile = order ? static_cast<int64_t>(order[blockIdx.x]) : sx_tile_of_block(blockIdx.x, L.nst, swizzle);
    if (tile < 0 || tile >= L.nst) return;
    const sx_rb_supertile S = L.st[tile];
    double sum[RB_RPL];
    rb_supertile_sum(L, S, x, ncols, lds, sum);
#pragma unroll
    for (int r = 0; r < RB_RPL; ++r) {
        const int lr = threadIdx.x + r * RB_TW;
        if (lr < S.nrows) {
            const int64_t row = S.row0 + lr;
            const double sp = b[row] - sum[r];
            if (s_p) s_p[row] = sp;
            if (flag) flag[row] = (sp < (gamma_dual * (-y[row]))) ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------- CG, CSR pass
// q[i] = (A w)[i] + xs[i]^2 * p[i];  partial[block] = sum p[i]*q[i]; with p == nullptr: q = A w,
// partial = sum q[i]^2 -- k_cg_a of sx_cg.hip over the layout.
struct RbCgState { // leading fields of CgState (sx_cg.hip): only `done` is read here
    double rho[2];
    double atol;
    double sumsq;
    long long iters;
    int done;
    int converged;
};

__global__ __launch_bounds__(RB_TW, RB_MINW) void k_rb_cg_a(RbLayout L, int swizzle, const int32_t *__restrict__ order, int64_t order_n,
                                                            const RbCgState *st, const double *__restrict__ w, int64_t ncols,
                                                            const double *__restrict__ xs, const double *__restrict__ p,
                                                            double *__restrict__ q, double *__restrict__ partial) {
    if (st->done) return;
    __shared__ RbLds lds;
    __shared__ double wave_sum[RB_TW / 64];
    int64_t t = blockIdx.x, t_end = L.nst, t_step = gridDim.x;
    if (swizzle) { // gridDim.x is a multiple of 8: XCD k walks the contiguous super-tiles [k*per, (k+1)*per)
        const int64_t per = (L.nst + 7) >> 3;
        t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        t_end = ((blockIdx.x & 7) + 1) * per;
        if (t_end > L.nst) t_end = L.nst;
        t_step = gridDim.x >> 3;
    }
    double dot = 0.0;
    if (order) { // slots of the dealt map: a block keeps to the slots of its XCD (gridDim.x is a multiple of 8)
        t = blockIdx.x;
        t_end = order_n;
        t_step = gridDim.x;
    }
    for (; t < t_end; t += t_step) {
        const int64_t tile = order ? static_cast<int64_t>(order[t]) : t;
        if (tile < 0) continue;
        const sx_rb_supertile S = L.st[tile];
        double sum[RB_RPL];
        rb_supertile_sum(L, S, w, ncols, lds, sum);
#pragma unroll
        for (int r = 0; r < RB_RPL; ++r) {
            const int lr = threadIdx.x + r * RB_TW;
            if (lr < S.nrows) {
                const int64_t row = S.row0 + lr;
                double qi = sum[r];
                if (p) {
                    const double s = xs[row], pi = p[row];
                    qi = qi + (s * s) * pi;
                    dot += pi * qi;
                } else {
                    dot += qi * qi;
                }
                q[row] = qi;
            }
        }
    }
    // fixed-order block sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = wave_sum[0];
#pragma unroll
        for (int k = 1; k < RB_TW / 64; ++k) tot += wave_sum[k];
        partial[blockIdx.x] = tot;
    }
}

inline RbLayout layout_of(const sx_rowblock *rb) {
    return RbLayout{rb->st, rb->chunks, rb->rowstart, rb->idx, rb->val, rb->nst, rb->nl > 0 ? rb->lprod : nullptr};
}

// pre-pass of the long rows: products in column order (x streamed), scattered to their slots.
// A line of lprod holds 16 consecutive positions of ONE long row, i.e. entries whose columns lie far apart (160,000
// columns at config 5), so it is completed by many workgroups.  When those sit on different XCDs every L2 writes its
// part of the line back on its own and the memory side merges them (read-modify-write: measured slower than the
// gathers it replaces).  Therefore XCD k takes the contiguous k-th eighth of the column-ordered list, tile after tile
// in dispatch order: the lines a row has open (128 bytes x the number of long rows per XCD) stay in that XCD's L2
// until they are complete and leave as whole lines.  The three list streams are read once (non-temporal).
constexpr int RB_LP_TILE = 8 * SX_WG; // entries per workgroup of the pre-pass
__global__ __launch_bounds__(SX_WG) void k_rb_long_products(int64_t nl, const int32_t *__restrict__ lcol, const double *__restrict__ lval,
                                                            const int32_t *__restrict__ le, const double *__restrict__ x,
                                                            double *__restrict__ lprod, const RbCgState *st) {
    if (st && st->done) return;
    const int64_t ntiles = (nl + RB_LP_TILE - 1) / RB_LP_TILE;
    const int64_t per = (ntiles + 7) >> 3;
    const int64_t tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || tile >= ntiles) return;
    const int64_t i0 = tile * RB_LP_TILE + threadIdx.x;
    int32_t c[8], e[8];
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t i = i0 + q * SX_WG;
        const bool ok = i < nl;
        c[q] = ok ? __builtin_nontemporal_load(lcol + i) : 0;
        e[q] = ok ? __builtin_nontemporal_load(le + i) : -1;
        v[q] = ok ? __builtin_nontemporal_load(lval + i) : 0.0;
    }
    double xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) xv[q] = x[c[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (e[q] >= 0) lprod[e[q]] = v[q] * xv[q];
}

inline unsigned rb_lp_grid(int64_t nl) {
    const int64_t ntiles = (nl + RB_LP_TILE - 1) / RB_LP_TILE;
    return static_cast<unsigned>(((ntiles + 7) >> 3) << 3);
}

} // namespace

namespace {
// order[] of sx_rowblock.h.  The XCD-contiguous map (block b -> XCD b mod 8 walks the b-th eighth of the super-tiles) keeps
// the x windows of neighbouring super-tiles in one L2, but it puts the long-row super-tiles -- one lane per linking row, a
// gather per entry: bound by latency, several times slower than their entry count says -- wherever the matrix has them:
// linking rows at the head of the LP = all of them on XCD 0 (netlib_lp at config-5 size: K2 0.92 ms against 0.37 ms for
// the lp_shard staircase, whose eight regions each bring their own).  Here the long super-tiles are DEALT over the eight
// XCDs, first in dispatch order, and the ordinary ones fill up behind them in ascending order, every XCD to the same cost
// (entries; a long super-tile counts LONG_COST times).  (Gathering them on ONE XCD so that the x lines they share are
// fetched once was measured too: 0.90 ms against 0.37 -- that XCD becomes the critical path; profiles/r04/experiments/k2_long_xcd.md.)
// mode 1: always; -1: only when the natural map is uneven (heaviest XCD's long-tile cost > 1.5 x the mean)
constexpr double RB_LONG_COST = 3.0;
int rb_build_order(sx_ctx *ctx, const sx_rowblock *rb, int mode) {
    rb->order_tried = mode ? mode : 1000;
    (void)sx_dfree(rb->order);
    rb->order = nullptr;
    rb->order_n = 0;
    const int64_t nst = rb->nst;
    if (nst < 64) return SX_OK;
    std::vector<sx_rb_supertile> st(static_cast<size_t>(nst));
    std::vector<sx_rb_chunk> ck(static_cast<size_t>(rb->nchunks));
    SX_HIP(hipMemcpyAsync(st.data(), rb->st, sizeof(sx_rb_supertile) * st.size(), hipMemcpyDeviceToHost, ctx->stream));
    SX_HIP(hipMemcpyAsync(ck.data(), rb->chunks, sizeof(sx_rb_chunk) * ck.size(), hipMemcpyDeviceToHost, ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<double> cost(static_cast<size_t>(nst));
    std::vector<int32_t> longs, plain;
    double total = 0.0, nat_long[8] = {0, 0, 0, 0, 0, 0, 0, 0}, long_total = 0.0;
    const int64_t per_nat = (nst + 7) >> 3;
    for (int64_t t = 0; t < nst; ++t) {
        int64_t ne = 0;
        for (int64_t k = 0; k < st[t].nchunks; ++k) ne += ck[static_cast<size_t>(st[t].chunk0 + k)].ne;
        const bool is_long = st[t].nrows > 0 && st[t].nrows <= RB_LONG_ROWS && ne > static_cast<int64_t>(RB_LONG_ROW) * st[t].nrows;
        cost[t] = static_cast<double>(ne) * (is_long ? RB_LONG_COST : 1.0) + 4096.0;
        total += cost[t];
        if (is_long) {
            nat_long[t / per_nat] += cost[t];
            long_total += cost[t];
        }
        (is_long ? longs : plain).push_back(static_cast<int32_t>(t));
    }
    if (longs.empty()) return SX_OK; // nothing to deal: the plain map stays
    if (mode < 0) {
        double worst = 0.0;
        for (double v : nat_long) worst = std::max(worst, v);
        if (worst <= 1.5 * long_total / 8.0) return SX_OK; // the matrix spreads them by itself
    }
    std::vector<std::vector<int32_t>> per(8);
    const double share = total / 8.0;
    double have[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < longs.size(); ++i) {
        per[i & 7].push_back(longs[i]);
        have[i & 7] += cost[longs[i]];
    }
    int xcd = 0;
    for (int32_t t : plain) {
        while (xcd < 7 && have[xcd] + 0.5 * cost[t] > share) ++xcd;
        per[xcd].push_back(t);
        have[xcd] += cost[t];
    }
    size_t most = 0;
    for (auto &v : per) most = std::max(most, v.size());
    std::vector<int32_t> order(most * 8, -1);
    for (int x = 0; x < 8; ++x)
        for (size_t k = 0; k < per[x].size(); ++k) order[k * 8 + static_cast<size_t>(x)] = per[x][k];
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&rb->order), sizeof(int32_t) * order.size()));
    SX_HIP(hipMemcpyAsync(rb->order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice, ctx->stream));
    SX_HIP(hipStreamSynchronize(ctx->stream));
    rb->order_n = static_cast<int64_t>(order.size());
    return SX_OK;
}
const int32_t *rb_order(sx_ctx *ctx, const sx_rowblock *rb) {
    if (ctx->opt_rb_long_xcd == 0 || !ctx->opt_xcd_swizzle) return nullptr;
    const int want = ctx->opt_rb_long_xcd;
    if (rb->order_tried != want && rb_build_order(ctx, rb, want) != SX_OK) return nullptr;
    return rb->order;
}
} // namespace

int sx_rb_score_rows(sx_ctx *ctx, const sx_rowblock *rb, int64_t ncols, const double *x, const double *b,
                     const double *y, double gamma_dual, double *s_p, uint8_t *flag) {
    if (rb->nst == 0) return SX_OK;
    const int swz = ctx->opt_xcd_swizzle;
    unsigned grid = swz ? static_cast<unsigned>(((rb->nst + 7) >> 3) << 3) : static_cast<unsigned>(rb->nst);
    const int32_t *order = rb_order(ctx, rb);
    if (order) grid = static_cast<unsigned>(rb->order_n);
    if (rb->nl > 0)
        hipLaunchKernelGGL(k_rb_long_products, dim3(rb_lp_grid(rb->nl)), dim3(SX_WG), 0, ctx->stream,
                           rb->nl, rb->lcol, rb->lval, rb->le, x, rb->lprod, static_cast<const RbCgState *>(nullptr));
    hipLaunchKernelGGL(k_rb_score_rows, dim3(grid), dim3(RB_TW), 0, ctx->stream, layout_of(rb), swz, order, x, ncols, b, y,
                       gamma_dual, s_p, flag);
    SX_HIP(hipGetLastError());
    return SX_OK;
}

int sx_rb_cg_a(sx_ctx *ctx, const sx_rowblock *rb, int64_t ncols, const void *cg_state, const double *w,
               const double *xs, const double *p, double *q, double *partial, int max_parts, int *nparts) {
    int64_t g = rb->nst < max_parts ? rb->nst : max_parts;
    const int swz = (ctx->opt_xcd_swizzle && rb->nst >= 64) ? 1 : 0;
    if (swz) g &= ~static_cast<int64_t>(7);
    if (g < 1) g = 1;
    if (rb->nl > 0)
        hipLaunchKernelGGL(k_rb_long_products, dim3(rb_lp_grid(rb->nl)), dim3(SX_WG), 0, ctx->stream,
                           rb->nl, rb->lcol, rb->lval, rb->le, w, rb->lprod, static_cast<const RbCgState *>(cg_state));
    const int32_t *order = swz ? rb_order(ctx, rb) : nullptr;
    hipLaunchKernelGGL(k_rb_cg_a, dim3(static_cast<unsigned>(g)), dim3(RB_TW), 0, ctx->stream, layout_of(rb), swz, order, rb->order_n,
                       static_cast<const RbCgState *>(cg_state), w, ncols, xs, p, q, partial);
    SX_HIP(hipGetLastError());
    *nparts = static_cast<int>(g);
    return SX_OK;
}
