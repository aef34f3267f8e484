// LDS operand window of the column walk (K1 score_columns, K10 price).
//
// Both kernels are bound by the L1<->L2 fabric, not by HBM, when the y[row] gathers miss the 32 KiB
// vector L1: every miss moves a 128-byte line for 8 useful bytes (profiles/r01: 59 M L2 requests per
// K1 launch against 11 M needed by the streamed arrays, i.e. ~18 TB/s of L2 traffic).  When the row
// indices of a tile cluster inside a window of SXL_CAP rows -- block-angular / staircase LPs, time-
// expanded networks -- a workgroup instead loads that window once with coalesced loads into LDS and
// serves the gathers from there; SXL_RUN neighbouring tiles share one window load because their
// windows overlap almost entirely.  Indices outside the window fall back to global memory, so the
// result is bit-identical to the plain walk whatever the matrix looks like.
#pragma once

#include "sx_segwalk.h"

#ifndef SXL_CAP_V
#define SXL_CAP_V 4096
#endif
constexpr int SXL_CAP = SXL_CAP_V;   // window length in doubles (32 KiB of LDS)
constexpr int SXL_CHUNK = 1024; // staged entries per chunk of the windowed walk: 8 KiB of LDS, so that window + chunk
                                // = 40 KiB and FOUR workgroups share a CU (2048: three; K1 0.341 -> 0.312 ms at c5)

struct sx_stage_win {
    const double *__restrict__ vec;
    const double *win;
    int64_t wlo;
    __device__ __forceinline__ void operator()(double v, int32_t i, double (&o)[1]) const {
        const uint64_t d = static_cast<uint64_t>(static_cast<int64_t>(i) - wlo);
        const double yv = (d < static_cast<uint64_t>(SXL_CAP)) ? win[d] : vec[i];
        o[0] = v * yv;
    }
};

// all 256 lanes: win[k] = vec[min(wlo + k, bound - 1)], 16 independent coalesced loads per lane.
// The caller synchronises before the first gather.
__device__ __forceinline__ void sx_window_fill(double *win, const double *__restrict__ vec, int64_t wlo,
                                               int64_t bound) {
#pragma unroll
    for (int r = 0; r < SXL_CAP / SX_WG; ++r) {
        const int k = r * SX_WG + threadIdx.x;
        int64_t row = wlo + k;
        if (row > bound - 1) row = bound - 1;
        win[k] = vec[row];
    }
}
