// Operand slabs of a segment walk whose gathers have no locality (uniformly random row indices in the column
// walk K1 / K10, column indices in the row walk K2).
//
// Without locality every gather is a cache miss of its own once the operand (8 bytes x rows resp. columns) no longer
// fits an XCD's 4 MiB L2: K1 0.19, K2 0.09, K10 0.16 of the HBM peak at config 5 with uniform rows
// (bench.py roofline_uniform).  The stable layouts keep the indices of a segment ASCENDING, so the entries can be cut
// into R slabs by operand index -- slab s holds, segment by segment, the entries whose index lies in
// [s W, (s + 1) W) -- and walked slab after slab: pass s gathers from 8 W bytes only (W chosen so that they stay in
// L2), starts every segment's sum from what pass s - 1 left (`carry`, 8 bytes in + 8 bytes out per segment and pass)
// and the last pass runs the kernel's epilogue.  The adds of a segment are the same adds in the same left-to-right
// order, each pass continuing the chain: bit-identical sums.  Cost: + (8 ptr + 16 carry) bytes per segment and slab,
// which the automatic rule weighs against the 12 bytes per entry of the stream.
//
// A matrix with a segment whose indices descend somewhere stays on the plain walk (the builder refuses it).
#pragma once

#include <cstdint>

struct sx_ctx;
struct sx_matrix;

struct sx_slab {
    int64_t *ptr = nullptr;    // nseg + 1 entry offsets of this slab
    int32_t *idx = nullptr;    // operand index minus `off`
    double *val = nullptr;
    int64_t *tiles = nullptr;  // tile table of ptr (sx_tiles.hip)
    int64_t ntiles = 0, nnz = 0;
    int64_t off = 0;           // first operand index of the slab
};

struct sx_slabs {
    int R = 0;
    int64_t nseg = 0, width = 0;
    sx_slab *slab = nullptr;   // host array of R
    double *carry = nullptr;   // nseg running sums between passes (device)
};

// Slabs of A's row walk (which = 0: CSR, operand = x) or column walk (which = 1: CSC, operand = y) under ctx's
// "slabs" option: -1 auto (operand beyond 3.6 MB, R = operand / 3.2 MB, >= 2^22 entries, (8 + 16) R nseg <= 1.25 x 12 nnz), 0 never,
// R >= 2 forced.  *out = nullptr: plain walk.  Built on first use, kept by the matrix.
int sx_slabs_get(sx_ctx *ctx, const sx_matrix *A, int which, const sx_slabs **out);
void sx_slabs_free(sx_slabs *S);
