// The windowed column walk over a RUN of tiles with its loads one step ahead (K1 score_columns, K10 price, the column
// pass of the projector CG).
//
// sx_segwalk (sx_segwalk.h) starts every tile with a chain of dependent loads -- tiles[t] -> ptr[s0], ptr[seg] -> the
// first entries -- and every further chunk with another round trip: at 4 workgroups per CU a tile of 2,048 entries
// takes ~8.8 us, ~7 of them waiting (profiles/r03/experiments/runwalk.md), and the kernels sit at 0.55 of the HBM peak
// although they move exactly the algorithmic bytes.  Here a workgroup that owns tiles [t0, t1)
//   * reads tiles[t0..t1] and ptr[tiles[.]] ONCE, by the first lanes of every wave, and hands them round by v_readlane
//     (no LDS: window + product chunk fill the 40 KiB that let four workgroups share a CU);
//   * requests the entries of the NEXT chunk -- the next tile's first chunk at a tile's end -- before it multiplies the
//     current one (two register sets, used alternately: 12 VGPRs);
//   * requests the next tile's segment bounds and epilogue operands at the start of the current tile.
// Sums, roundings and outputs are those of sx_segwalk: same products, same left-to-right adds per segment.
#pragma once

#include "sx_window.h"

struct sx_quad {
    sx_v4i i;
    sx_v2d v01, v23;
};

__device__ __forceinline__ int64_t sx_readlane64(int64_t v, int lane) {
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), lane);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), lane);
    return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

// Pre:  P operator()(int64_t seg)                        -- requests the epilogue's operands of segment seg (always a
//                                                            valid segment: lanes beyond the tile get a clamped one)
// Epi:  void operator()(int64_t seg, bool valid, double sum, const P &p)
// RUN <= 8 tiles; all lanes of the workgroup call it; the caller has filled the window and synchronised.
//
// Every load below is issued unconditionally, with a clamped address where the entry or segment does not exist: the
// hardware counts outstanding loads, so a load that is issued on one path only makes the compiler drain the queue
// (s_waitcnt vmcnt(0)) in front of the next use of ANY loaded register -- the first version of this walk did exactly
// that and was slower than the plain one.  The tile loop is unrolled (static register sets), the chunk loop is not:
// the next chunk's entries are requested at the top of an iteration and waited for at its bottom (the copy into the
// current set), behind the multiply, the barrier and the adds.
template <int RUN, class Stage /* sx_stage_win */, class Pre, class Epi>
__device__ __forceinline__ void sx_runwalk(const int64_t *__restrict__ tiles, int64_t t0, int64_t t1,
                                           const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                           const double *__restrict__ val, const Stage &stage,
                                           sx_walk_lds<1, SXL_CHUNK> &lds, const Pre &pre, const Epi &epi) {
    static_assert(SXL_CHUNK == SX_SWEEP, "one 4-entry load per lane and chunk");
    static_assert(RUN + 1 <= 64, "run header in one wave");
    using P = decltype(pre(int64_t(0)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int nt = static_cast<int>(t1 - t0);
    // run header: lane k of every wave holds tiles[t0 + k] and ptr[tiles[t0 + k]] (lanes past the run: the last ones)
    const int64_t hs = tiles[t0 + (lane < nt ? lane : nt)];
    const int64_t hp = ptr[hs];
    auto seg0 = [&](int k) { return sx_readlane64(hs, k < nt ? k : nt); };
    auto pos0 = [&](int k) { return sx_readlane64(hp, k < nt ? k : nt); };
    const int64_t seg_last = seg0(nt) - 1; // >= 0: a run holds at least one segment

    struct Lane {
        int64_t seg, cs, ce;
        bool valid;
    };
    auto load_lane = [&](int k, Lane &L) {
        const int64_t s1 = seg0(k + 1);
        L.seg = seg0(k) + tid;
        L.valid = L.seg < s1;
        const int64_t sc = L.seg < seg_last ? L.seg : seg_last; // lanes beyond the tile: unused values of a real segment
        L.cs = ptr[sc];
        L.ce = ptr[sc + 1];
        return pre(sc);
    };
    // entries [base + 4 tid, + 4) of a tile that ends at p_hi; lanes past the slice re-read its first quad (their LDS
    // slots are never consumed; the arrays are padded by 8 entries, so base = nnz & ~3 is still readable)
    auto load_quad = [&](int64_t base, int64_t p_hi, sx_quad &q) {
        const int64_t e = base + tid * 4;
        const int64_t at = (e < p_hi) ? e : base;
        q.i = *reinterpret_cast<const sx_v4i *>(idx + at);
        q.v01 = *reinterpret_cast<const sx_v2d *>(val + at);
        q.v23 = *reinterpret_cast<const sx_v2d *>(val + at + 2);
    };

    Lane L, Ln;
    sx_quad Q, Qn;
    P pr = load_lane(0, L), prn = pr;
    Ln = L;
    load_quad(pos0(0) & ~static_cast<int64_t>(3), pos0(1), Q);
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
        if (k < nt) {
            prn = load_lane(k + 1, Ln); // (k + 1 == nt: the clamped header repeats the last tile's end; nobody uses it)
            const int64_t p_hi = pos0(k + 1);
            const int64_t nbase0 = pos0(k + 1) & ~static_cast<int64_t>(3), np_hi = pos0(k + 2);
            int64_t base = pos0(k) & ~static_cast<int64_t>(3);
            const int64_t cs = L.valid ? L.cs : p_hi, ce = L.valid ? L.ce : p_hi;
            double acc = 0.0;
            do { // a tile without entries takes one empty turn
                const int64_t nb = base + SXL_CHUNK;
                const bool same = nb < p_hi;
                // Gathers in two unconditional halves -- the window by ds_read (clamped slot), the rows outside it by a
                // global load (lanes inside the window all read vec[wlo]: one line) -- instead of the per-lane select
                // of sx_stage_win, which the compiler turns into FLAT loads: flat results are waited for with
                // vmcnt(0) and would drain the request of the next chunk.  Order: global gathers, then the request,
                // then the LDS reads; the products wait for the gathers only (vmcnt(3)).
                const int32_t gi[4] = {Q.i.x, Q.i.y, Q.i.z, Q.i.w};
                double yg[4], yl[4];
                bool in[4];
                uint32_t dd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t d = static_cast<uint64_t>(static_cast<int64_t>(gi[q]) - stage.wlo);
                    in[q] = d < static_cast<uint64_t>(SXL_CAP);
                    dd[q] = in[q] ? static_cast<uint32_t>(d) : 0u;
                    yg[q] = stage.vec[in[q] ? stage.wlo : static_cast<int64_t>(gi[q])];
                }
                __builtin_amdgcn_sched_barrier(0);
                load_quad(same ? nb : nbase0, same ? p_hi : np_hi, Qn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) yl[q] = stage.win[dd[q]];
                double o0[1], o1[1], o2[1], o3[1];
                o0[0] = Q.v01.x * (in[0] ? yl[0] : yg[0]);
                o1[0] = Q.v01.y * (in[1] ? yl[1] : yg[1]);
                o2[0] = Q.v23.x * (in[2] ? yl[2] : yg[2]);
                o3[0] = Q.v23.y * (in[3] ? yl[3] : yg[3]);
                double2 *dst = reinterpret_cast<double2 *>(&lds.v[0][tid * 4]);
                dst[0] = make_double2(o0[0], o1[0]);
                dst[1] = make_double2(o2[0], o3[0]);
                __syncthreads();
                const int64_t k0 = cs > base ? cs : base;
                const int64_t k1 = ce < nb ? ce : nb;
                int o = static_cast<int>(k0 - base);
                int left = (k1 > k0) ? static_cast<int>(k1 - k0) : 0;
                for (; left >= 8; left -= 8, o += 8) {
                    double t[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = lds.v[0][o + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = acc + t[q];
                }
                for (; left > 0; --left, ++o) acc = acc + lds.v[0][o];
                __syncthreads();
                Q = Qn;
                base = nb;
            } while (base < p_hi);
            epi(L.seg, L.valid, acc, pr);
            L = Ln;
            pr = prn;
        }
    }
}
