// Bordered band basis (kernel group K16b, sx_border.hip): the solves of the sparse crossover's basis
//
//        B_aug = [ B11  B12 ]     B11  m1 x m1  band part (K16f, sx_bandlu.hip): one variable per band row, or a PLACEHOLDER
//                [ B21  B22 ]                   (unit vector) where the matching / the band LU left a position empty;
//                                 border rows   the dense (linking) rows of the LP, then one row "z_p = 0" per placeholder p;
//                                 border cols   the basic variables that found no band row (linking activities, logicals of
//                                               dense rows, columns the band LU set aside);
// with the Schur complement S = B22 - B21 B11^-1 B12 (nb x nb, dense: K16g, sx_denselu.hip).  The reference leaves every
// basis factorisation to its solvers (solver_caller/gurobi.py:202-210).  Position space: [0, m1) band positions, then the
// nb border columns; row space of a right-hand side: [0, m1) band rows, then the nb border rows (mp = m1 + nb either way).
#pragma once

#include "sx_internal.h"

// rows of a sparse matrix on the device, optionally only the listed ones (list == nullptr: row r is row r)
struct SxRowsDev {
    int64_t nrows = 0;
    const int64_t *ptr = nullptr;  // [nrows + 1]
    const int32_t *idx = nullptr;
    const double *val = nullptr;
    const int32_t *list = nullptr; // [nrows] the row each entry of ptr stands for
};

struct SxBorderOps {
    sx_ctx *ctx = nullptr;
    int64_t m1 = 0, nb = 0, mp = 0;
    sx_bandlu *lu = nullptr;   // B11 (may be null when m1 == 0)
    sx_denselu *dl = nullptr;  // S (null while it is being assembled, or when nb == 0)
    SxRowsDev b21_rows;        // per border row: (band position, value)
    SxRowsDev b21_cols;        // per band position that has one: (border row, value)
    SxRowsDev b12_rows;        // per band position that has one: (border column, value)
    SxRowsDev b12_cols;        // per border column: (band position, value)
    double *work = nullptr;    // m1 x work_cols doubles (second band solve of an FTRAN; B21^T y of a BTRAN)
    int64_t work_cols = 0;
    double tiny = 1e-60;       // sx_bandlu_solve_sparse_dev: windows below this are zeros
    // windows of the columns of the last FTRAN with sparse_rhs: B11^-1 a1 of an LP column is LOCAL (its entries, and what the
    // band's fill carries a few hundred rows on), rows [win_lo[s], win_hi[s]) hold all that exceeds SX_BORDER_WIN_EPS
    int32_t *win_lo = nullptr, *win_hi = nullptr;
    int64_t win_cap = 0;
    // V = B11^-1 B12 packed by those windows (built while the Schur complement is assembled: pack_v): the second band solve
    // of an FTRAN becomes x1 = w1 - V x2, a product with a few columns per row.  v_ready = 0: not available (the windows
    // did not fit v_cap doubles -- no locality): the solve form is used
    double *v_val = nullptr;
    size_t v_cap = 0, v_used = 0, v_limit = 0; // (doubles; v_limit: how far v_val may grow)
    std::vector<int32_t> v_lo, v_hi;   // [nb] window of border column j (host)
    std::vector<int64_t> v_off;        // [nb] where it starts in v_val
    int32_t *d_v_lo = nullptr, *d_v_hi = nullptr, *d_vb_col = nullptr;
    int64_t *d_v_off = nullptr, *d_vb_ptr = nullptr;
    int64_t v_nblk = 0;
    int v_ready = 0, v_failed = 0;
    ~SxBorderOps();
    SxBorderOps() = default;
    SxBorderOps(const SxBorderOps &) = delete;
    SxBorderOps &operator=(const SxBorderOps &) = delete;

    // W (mp x ncols, leading dimension mp), columns in row space -> B_aug^-1 W in position space.  sparse_rhs: the columns
    // are columns of an LP (a handful of entries each).  upto_schur: stop after the border rows hold
    // a2 - B21 B11^-1 a1 (the columns of S when W holds the border columns; dl is not touched)
    int ftran(double *W, int64_t ncols, bool sparse_rhs, bool upto_schur = false);
    // v (mp, position space: costs of the basic variables) -> B_aug^-T v (row space: duals), one vector, in place
    int btran(double *v);
    // after ftran(W, ncols, true, true) of the border columns dest[0 .. ncols) (host): their band parts -> V (windows appended)
    int pack_v(const double *W, int64_t ncols, const int32_t *dest);
    // all border columns packed: build the row-block lists.  zero[j] != 0: column j was replaced (its V column is zero)
    int finish_v(const std::vector<int32_t> &zero);
};
constexpr double SX_BORDER_WIN_EPS = 1e-40;
