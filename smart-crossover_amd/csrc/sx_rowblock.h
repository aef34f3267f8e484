// Column-blocked row layout ("row blocks") of the row walk: K2 score_rows and the CSR product of the
// projector CG.
//
// Why: the plain row walk gathers x[col] straight from global memory, one 128-byte L2 line per 8 useful
// bytes; at config 5 that is 8.45e7 L2 requests per launch and the walk sits on the L1<->L2 fabric
// ceiling (profiles/r01/pmc_hbm_bench_c5.txt).  Here the rows are cut into *super-tiles* (<= RB_R
// consecutive rows, entry budget) and the columns into blocks of RB_CWIN columns; the entries of a
// super-tile are stored cell by cell -- a cell is one column block that holds enough entries to pay for
// a window ("windowed"), or a run of consecutive sparse blocks ("direct") -- and inside a cell by
// (row, stored order).  A workgroup walks the cells of its super-tile in ascending column order, loads
// the x block of a windowed cell into LDS with coalesced loads and gathers from LDS; lane t keeps the
// running sums of its rows (t, t + RB_TW, ...) in registers.  A row whose columns do not descend meets its entries in stored
// order, so the sums are the sequential sums of the plain walk, bit for bit (a matrix with a row whose
// columns descend somewhere stays on the plain walk: the builder refuses it).
//
// Long rows (more than RB_LONG_ROW entries: the linking rows of a block-angular LP) are summed by one lane
// each, entry after entry; they get super-tiles of their own (<= RB_LONG_ROWS rows) whose cells are slices
// by *position inside the row* (RB_CHUNK / rows entries of every row per cell, gathered from global
// memory), so that every staged chunk holds a piece of every row and all its lanes add at the same time.
//
// Device arrays of a layout:
//   idx, val      entries, cell after cell; every cell starts at a multiple of 4 entries, the gap is
//                 filled with (col0, 0.0) pairs -- (0, 0.0) in a direct cell -- that no row segment covers
//   chunk[]       one record per staged chunk (<= RB_CHUNK entries of one cell), in walk order
//   rowstart[]    per cell: RB_RS_STRIDE uint16 offsets, rowstart[r] = first entry of local row r inside
//                 the cell, rowstart[r >= nrows] = entries of the cell (cells hold <= 65535 entries)
//   st[]          per super-tile: first row, row count, first chunk, chunk count
// tools/rb_layout.py is the numpy statement of the same construction; the tests compare the two array by
// array.
#pragma once

#include <cstdint>

struct sx_ctx;
struct sx_matrix;

struct sx_rb_chunk {
    int64_t e0;    // first entry of the chunk in idx / val (multiple of 4)
    int32_t ne;    // entries in this chunk, <= RB_CHUNK
    int32_t col0;  // first column of the cell's LDS window, or RB_NO_WINDOW for a direct cell
    int32_t cell;  // cell number (-> rowstart)
    int32_t base;  // offset of the chunk's first entry inside its cell
    int32_t fresh; // 1: first chunk of its cell (new rowstart, new window)
    int32_t staged; // 1: chunk of a long-row super-tile whose products a pre-pass leaves in lprod (below)
};

struct sx_rb_supertile {
    int64_t row0;
    int64_t chunk0;
    int32_t nrows;
    int32_t nchunks;
};

constexpr int32_t RB_NO_WINDOW = -(1 << 30);

// parameters of the layout the library builds (measured on MI355X, profiles/r02/rb_bench_*.txt)
constexpr int RB_TW = 512;             // lanes per workgroup of the walk
constexpr int RB_RPL = 1;              // rows per lane: lane t owns rows t, t + RB_TW, ... of its super-tile
constexpr int RB_R = RB_TW * RB_RPL;   // rows per super-tile
constexpr int RB_CWIN = 4096;          // columns per block = doubles of the LDS window
constexpr int RB_CHUNK = 2048;         // entries staged per step
constexpr int RB_DENSE_MIN = 512;      // a block with at least this many entries gets a window
constexpr int RB_BUDGET = 96 * RB_R;   // entries per super-tile of ordinary rows
constexpr int RB_MERGE_MAX = 32768;    // entries of a direct cell made of merged sparse blocks
constexpr int RB_LONG_ROW = 512;       // a row with more entries is "long"
constexpr int RB_LONG_ROWS = 64;       // rows per super-tile of long rows
constexpr int RB_LONG_BUDGET = 65536;  // entries per super-tile of long rows
constexpr int RB_RS_STRIDE = RB_R + 4; // uint16 slots per cell in rowstart[]
constexpr int RB_CELL_MAX = 65535;

struct sx_rowblock {
    int64_t nst = 0, ncells = 0, nchunks = 0, nent = 0; // nent: entries incl. gaps (idx / val hold nent + 8)
    int64_t nnz = 0, windowed = 0;                      // entries of the matrix / of windowed cells
    sx_rb_supertile *st = nullptr;
    sx_rb_chunk *chunks = nullptr;
    uint16_t *rowstart = nullptr;
    int32_t *idx = nullptr;
    double *val = nullptr;
    // Long rows (the linking rows of an LP) gather x one 128-byte line per entry -- 1.28 GB of the 2.26 GB K2 moved at
    // config 5.  Their entries are therefore ALSO kept sorted by column (lcol / lval, and le = the entry's slot in
    // idx / val): a pre-pass streams that list and x in column order, forms the same rounded products and scatters
    // them to lprod[le]; the walk then reads the products of a staged chunk instead of entries + gathers.  Same
    // products, same order of the adds: bit-identical sums.
    int64_t nl = 0;
    int32_t *lcol = nullptr, *le = nullptr;
    double *lval = nullptr, *lprod = nullptr;
    // Optional block -> super-tile map of the walks (option "rb_long_xcd", built on first use; sx_rowblock.hip rb_build_order):
    // the long-row super-tiles dealt over the eight XCDs, the ordinary ones behind them in ascending order.  order[b] < 0: no tile.
    mutable int32_t *order = nullptr;
    mutable int64_t order_n = 0;
    mutable int order_tried = 0;
};

// Layout of A's rows under ctx's "rowblock" option (-1 auto: built for matrices of >= RB_AUTO_NNZ entries
// when at least half of them land in windowed cells; 0 never; 1 whenever the matrix admits it).  *out is
// nullptr when the plain walk should be used.  Built on first use and kept by the matrix.
constexpr int64_t RB_AUTO_NNZ = 1 << 22;
constexpr int64_t RB_AUTO_ROWS = 1 << 18; // ... and of at least 512 super-tiles of 512 rows: below, the walk has too few
                                          // workgroups to fill the chip (1e5 rows x 8e6 entries: 175 us against 110 us plain)
int sx_rowblock_get(sx_ctx *ctx, const sx_matrix *A, const sx_rowblock **out);
void sx_rowblock_free(sx_rowblock *rb);

// K2 over the layout (sx_rowblock.hip); same outputs as k_score_rows
int sx_rb_score_rows(sx_ctx *ctx, const sx_rowblock *rb, int64_t ncols, const double *x, const double *b,
                     const double *y, double gamma_dual, double *s_p, uint8_t *flag);
// CSR pass of the projector CG over the layout: q = A w (+ xs^2 .* p), partial[block] as k_cg_a;
// *nparts = number of partials written
int sx_rb_cg_a(sx_ctx *ctx, const sx_rowblock *rb, int64_t ncols, const void *cg_state, const double *w,
               const double *xs, const double *p, double *q, double *partial, int max_parts, int *nparts);
