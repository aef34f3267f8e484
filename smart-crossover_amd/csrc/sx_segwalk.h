// Sequential segment walk: the device pattern behind every kernel that must reproduce a
// per-column (CSC) or per-row (CSR) sum *in stored order with separately rounded products*
// -- the rounding order of scipy's csc_matvec / csr_matvec, which the reference's index sets
// depend on.
//
// Work is cut into *tiles*: runs of at most 256 consecutive segments holding at most about
// SX_TILE_BUDGET entries (sx_tiles.hip builds the table once per matrix), so that a few very long
// segments (linking rows of an LP) do not serialise inside one workgroup.  A workgroup of 256
// lanes owns one tile, i.e. one contiguous slice [ptr[s0], ptr[s1]) of the entry arrays, and
// streams it in chunks:
//   stage   : all lanes read entries with 16-byte loads (4 x int32 index, 2 x 2 x double value),
//             gather the vector operand, form the rounded products and park them in LDS;
//             this is where the HBM traffic and the memory-level parallelism are;
//   consume : lane t adds the products of segment s0+t, in stored order, to its running sum
//             (LDS reads are issued eight ahead of the dependent adds).
// Chunks are visited in ascending order, so a segment that straddles chunks (or is longer than
// a chunk) is still summed strictly left to right.  Compile with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int SX_WG = 256;          // lanes per workgroup = max segments per tile
constexpr int SX_SWEEP = SX_WG * 4; // entries staged by one sweep of 4-wide loads
constexpr int SX_TILE_BUDGET = 8192; // a tile is cut when its entry count would pass this

template <int NACC, int CHUNK>
struct sx_walk_lds {
    static_assert(CHUNK % SX_SWEEP == 0, "CHUNK must be a multiple of one sweep");
    double v[NACC][CHUNK];
};

// XCD-aware block -> tile map (speed only): blocks b and b+8 share an XCD (round-robin dispatch),
// so giving XCD k the contiguous tile range [k*T/8, (k+1)*T/8) keeps the vector operand that
// neighbouring tiles gather from in that XCD's L2.
__device__ __forceinline__ int64_t sx_tile_of_block(int64_t b, int64_t ntiles, int swizzle) {
    if (!swizzle) return b;
    const int64_t per = (ntiles + 7) >> 3;
    const int64_t t = (b & 7) * per + (b >> 3);
    return t; // may be >= ntiles for the last XCD's tail: caller skips
}

typedef int sx_v4i __attribute__((ext_vector_type(4)));
typedef double sx_v2d __attribute__((ext_vector_type(2)));

// 16-byte loads of the streamed entry arrays.  NT selects how they are issued:
//   0        plain global loads;
//   1        global loads marked non-temporal;
//   >= 2     raw buffer loads whose cache-policy bits are NT (2 = nt, 16 = sc1, 17 = sc0 sc1,
//            18 = nt sc1): the once-read stream is fetched at agent scope so that it does not
//            displace the gathered operand from the 32 KB vector L1.  The descriptor is rebased
//            to the chunk, so the 32-bit byte offsets never leave the chunk.
struct sx_stream {
    const int32_t *idx;
    const double *val;
    __amdgpu_buffer_rsrc_t ridx, rval;
};

template <int NT>
__device__ __forceinline__ sx_stream sx_stream_at(const int32_t *idx, const double *val, int64_t base) {
    sx_stream s;
    s.idx = idx + base;
    s.val = val + base;
    if constexpr (NT >= 2) {
        s.ridx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(s.idx), 0, 0x7fffffff, 0x00020000);
        s.rval = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(s.val), 0, 0x7fffffff, 0x00020000);
    }
    return s;
}

// entries [off, off+4) of the chunk
template <int NT>
__device__ __forceinline__ void sx_ld_quad(const sx_stream &s, int off, sx_v4i &i4, sx_v2d &v01, sx_v2d &v23) {
    if constexpr (NT >= 2) {
        i4 = __builtin_amdgcn_raw_buffer_load_b128(s.ridx, off * 4, 0, NT);
        const sx_v4i a = __builtin_amdgcn_raw_buffer_load_b128(s.rval, off * 8, 0, NT);
        const sx_v4i b = __builtin_amdgcn_raw_buffer_load_b128(s.rval, off * 8 + 16, 0, NT);
        v01 = __builtin_bit_cast(sx_v2d, a);
        v23 = __builtin_bit_cast(sx_v2d, b);
    } else if constexpr (NT == 1) {
        i4 = __builtin_nontemporal_load(reinterpret_cast<const sx_v4i *>(s.idx + off));
        v01 = __builtin_nontemporal_load(reinterpret_cast<const sx_v2d *>(s.val + off));
        v23 = __builtin_nontemporal_load(reinterpret_cast<const sx_v2d *>(s.val + off + 2));
    } else {
        i4 = *reinterpret_cast<const sx_v4i *>(s.idx + off);
        v01 = *reinterpret_cast<const sx_v2d *>(s.val + off);
        v23 = *reinterpret_cast<const sx_v2d *>(s.val + off + 2);
    }
}

// Stage functor:  void operator()(double value, int32_t index, double (&out)[NACC]) const
// Walks tile `tile` (segments [tiles[tile], tiles[tile+1])).  On return acc[] holds the NACC
// running sums of segment `seg` (valid lanes only).
struct sx_no_prologue {
    __device__ __forceinline__ void operator()(int64_t, bool) const {}
};

// `pre(seg, valid)` runs as soon as the lane knows its segment, before any entry is streamed: the
// place to issue the loads an epilogue will need (c, x, l, u ...) so that their HBM latency overlaps
// the walk instead of following it.
template <int NACC, int CHUNK, int NT = 0, class Stage, class Pre = sx_no_prologue>
__device__ __forceinline__ void sx_segwalk(const int64_t *__restrict__ tiles, int64_t tile,
                                           const int64_t *__restrict__ ptr,
                                           const int32_t *__restrict__ idx,
                                           const double *__restrict__ val, const Stage &stage,
                                           sx_walk_lds<NACC, CHUNK> &lds, int64_t &seg, bool &valid,
                                           double (&acc)[NACC], const Pre &pre = Pre(), const bool keep = false) {
    const int tid = threadIdx.x;
    const int64_t s0 = tiles[tile];
    const int64_t s1 = tiles[tile + 1];
    seg = s0 + tid;
    valid = seg < s1;
    pre(seg, valid);
    const int64_t p_lo = ptr[s0];
    const int64_t p_hi = ptr[s1];
    int64_t cs = p_hi, ce = p_hi;
    if (valid) {
        cs = ptr[seg];
        ce = ptr[seg + 1];
    }
    if (!keep) { // keep: the sums continue from what the caller (or its `pre`) left in acc[] -- sx_slabs.h
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
    }

    for (int64_t base = p_lo & ~static_cast<int64_t>(3); base < p_hi; base += CHUNK) {
        // ---- stage: CHUNK / SX_SWEEP sweeps of 1024 entries
        const sx_stream src = sx_stream_at<NT>(idx, val, base);
#pragma unroll
        for (int r = 0; r < CHUNK / SX_SWEEP; ++r) {
            const int64_t sweep0 = base + static_cast<int64_t>(r) * SX_SWEEP;
            if (sweep0 < p_hi) { // uniform across the workgroup
                const int off = r * SX_SWEEP + tid * 4;
                const int64_t e = base + off;
                // lanes past the slice re-read its first quad (cached) instead of branching;
                // their LDS slots are never consumed
                const int eo = (e < p_hi) ? off : 0;
                sx_v4i i4;
                sx_v2d v01, v23;
                sx_ld_quad<NT>(src, eo, i4, v01, v23);
                double o0[NACC], o1[NACC], o2[NACC], o3[NACC];
                stage(v01.x, i4.x, o0);
                stage(v01.y, i4.y, o1);
                stage(v23.x, i4.z, o2);
                stage(v23.y, i4.w, o3);
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    double2 *dst = reinterpret_cast<double2 *>(&lds.v[a][off]);
                    dst[0] = make_double2(o0[a], o1[a]);
                    dst[1] = make_double2(o2[a], o3[a]);
                }
            }
        }
        __syncthreads();
        // ---- consume: strictly sequential per lane, reads batched ahead of the adds
        const int64_t k0 = cs > base ? cs : base;
        const int64_t k1 = ce < base + CHUNK ? ce : base + CHUNK;
        int o = static_cast<int>(k0 - base);
        int left = (k1 > k0) ? static_cast<int>(k1 - k0) : 0;
        for (; left >= 8; left -= 8, o += 8) {
            double t[NACC][8];
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int q = 0; q < 8; ++q) t[a][q] = lds.v[a][o + q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = acc[a] + t[a][q];
        }
        for (; left > 0; --left, ++o) {
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = acc[a] + lds.v[a][o];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ small reductions
__device__ __forceinline__ long long sx_wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
