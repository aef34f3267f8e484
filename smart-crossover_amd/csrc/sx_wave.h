// Wave-level helpers shared by the factorisation kernels (gfx950, 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

// maximum of a 64-bit key over the 64 lanes of a wave, in every lane: DPP moves (quad swaps, row rotations, the two row
// broadcasts), no LDS traffic
template <int CTRL>
__device__ __forceinline__ unsigned long long sx_dpp_max_u64(unsigned long long k) {
    const int lo = static_cast<int>(static_cast<unsigned>(k)), hi = static_cast<int>(static_cast<unsigned>(k >> 32));
    const unsigned olo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
    const unsigned ohi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false));
    const unsigned long long o = (static_cast<unsigned long long>(ohi) << 32) | olo;
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long sx_wave_max_u64(unsigned long long k) {
    k = sx_dpp_max_u64<0xb1>(k);  // quad_perm [1,0,3,2]
    k = sx_dpp_max_u64<0x4e>(k);  // quad_perm [2,3,0,1]
    k = sx_dpp_max_u64<0x124>(k); // row_ror 4
    k = sx_dpp_max_u64<0x128>(k); // row_ror 8: every lane holds its row's maximum
    k = sx_dpp_max_u64<0x142>(k); // row_bcast 15: rows 1..3 take in the row before them
    k = sx_dpp_max_u64<0x143>(k); // row_bcast 31: lane 63 holds the wave's maximum
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(k)), 63));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(k >> 32)), 63));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
