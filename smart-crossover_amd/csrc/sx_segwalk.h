// Sequential segment walk: the device pattern behind every kernel that must reproduce a
// per-column (CSC) or per-row (CSR) sum *in stored order with separately rounded products*
// -- the rounding order of scipy's csc_matvec / csr_matvec, which the reference's index sets
// depend on.
//
// A workgroup of 256 lanes owns 256 consecutive segments, i.e. one contiguous slice
// [ptr[s0], ptr[s0+256]) of the entry arrays.  The slice is streamed in chunks:
//   stage   : all lanes read entries with 16-byte loads (4 x int32 index, 2 x 2 x double value),
//             gather the vector operand, form the rounded products and park them in LDS;
//             this is where the HBM traffic and the memory-level parallelism are;
//   consume : lane t adds the products of segment s0+t, in stored order, to its running sum.
// Chunks are visited in ascending order, so a segment that straddles chunks (or is longer than
// a chunk) is still summed strictly left to right.  Compile with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int SX_WG = 256;        // lanes per workgroup = segments per workgroup
constexpr int SX_SWEEP = SX_WG * 4; // entries staged by one sweep of 4-wide loads

template <int NACC, int CHUNK>
struct sx_walk_lds {
    static_assert(CHUNK % SX_SWEEP == 0, "CHUNK must be a multiple of one sweep");
    double v[NACC][CHUNK];
};

// Stage functor:  void operator()(double value, int32_t index, double (&out)[NACC]) const
// On return acc[] holds the NACC running sums of segment `seg` (valid lanes only).
template <int NACC, int CHUNK, class Stage>
__device__ __forceinline__ void sx_segwalk(const int64_t *__restrict__ ptr,
                                           const int32_t *__restrict__ idx,
                                           const double *__restrict__ val, int64_t nseg,
                                           const Stage &stage, sx_walk_lds<NACC, CHUNK> &lds,
                                           int64_t &seg, bool &valid, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const int64_t s0 = static_cast<int64_t>(blockIdx.x) * SX_WG;
    const int64_t s1 = (s0 + SX_WG < nseg) ? s0 + SX_WG : nseg;
    seg = s0 + tid;
    valid = seg < nseg;
    const int64_t p_lo = ptr[s0];
    const int64_t p_hi = ptr[s1];
    int64_t cs = p_hi, ce = p_hi;
    if (valid) {
        cs = ptr[seg];
        ce = ptr[seg + 1];
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;

    for (int64_t base = p_lo & ~static_cast<int64_t>(3); base < p_hi; base += CHUNK) {
        // ---- stage: CHUNK / SX_SWEEP sweeps of 1024 entries
#pragma unroll
        for (int r = 0; r < CHUNK / SX_SWEEP; ++r) {
            const int64_t sweep0 = base + static_cast<int64_t>(r) * SX_SWEEP;
            if (sweep0 < p_hi) { // uniform across the workgroup
                const int off = r * SX_SWEEP + tid * 4;
                int64_t e = base + off;
                // lanes past the slice re-read its first quad (cached) instead of branching;
                // their LDS slots are never consumed
                const int64_t ee = (e < p_hi) ? e : base;
                const int4 i4 = *reinterpret_cast<const int4 *>(idx + ee);
                const double2 v01 = *reinterpret_cast<const double2 *>(val + ee);
                const double2 v23 = *reinterpret_cast<const double2 *>(val + ee + 2);
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
        // ---- consume: strictly sequential per lane
        const int64_t k0 = cs > base ? cs : base;
        const int64_t k1 = ce < base + CHUNK ? ce : base + CHUNK;
        for (int64_t k = k0; k < k1; ++k) {
            const int o = static_cast<int>(k - base);
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = acc[a] + lds.v[a][o];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ small reductions
__device__ __forceinline__ double sx_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_down(v, o, 64);
        v = (w < v) ? w : v;
    }
    return v;
}

__device__ __forceinline__ long long sx_wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
