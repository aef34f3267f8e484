// Bordered band basis (kernel group K16b): FTRAN / BTRAN through  B_aug = [B11 B12; B21 B22]  with B11 factored by the band
// LU (K16f), the Schur complement S = B22 - B21 B11^-1 B12 by the dense LU (K16g) and the two sparse off-diagonal blocks
// applied by gather kernels -- see sx_border.h for the layout.  Stands where the reference's solvers factor and solve with
// whatever basis they meet (solver_caller/gurobi.py:111-115, 202-210).
//
//   FTRAN  w1 = B11^-1 a1;  t2 = a2 - B21 w1;  x2 = S^-1 t2;  x1 = w1 - B11^-1 (B12 x2)
//   BTRAN  z = B11^-T v1;   y2 = S^-T (v2 - B12^T z);          y1 = z - B11^-T (B21^T y2)
// fp64, no atomics, a fixed order of additions per entry: deterministic.
#include "sx_border.h"

#include <algorithm>

namespace {

constexpr int BD_WG = 256;

// W[m1 + k, s] -= sum_e val[e] W[idx[e], s]: a wave per (border row k, column s), the entries dealt over its lanes
__global__ __launch_bounds__(BD_WG) void k_bd_rows_apply(SxRowsDev R, int64_t ncols, double *__restrict__ W, int64_t mp, int64_t m1) {
    const int64_t wv = (static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wv >= R.nrows * ncols) return;
    const int64_t k = wv % R.nrows, s = wv / R.nrows;
    double *col = W + static_cast<size_t>(s) * mp;
    double acc = 0.0;
    for (int64_t e = R.ptr[k] + lane; e < R.ptr[k + 1]; e += 64) acc += R.val[e] * col[R.idx[e]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) col[m1 + k] -= acc;
}

// Rw[p, s] = sum_e val[e] X2[idx[e], s] for the listed band positions p (Rw zero elsewhere)
__global__ __launch_bounds__(BD_WG) void k_bd_b12_apply(SxRowsDev R, int64_t ncols, const double *__restrict__ X2, int64_t ldx,
                                                        double *__restrict__ Rw, int64_t ldr) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x;
    if (t >= R.nrows * ncols) return;
    const int64_t k = t % R.nrows, s = t / R.nrows;
    const double *x = X2 + static_cast<size_t>(s) * ldx;
    double acc = 0.0;
    for (int64_t e = R.ptr[k]; e < R.ptr[k + 1]; ++e) acc += R.val[e] * x[R.idx[e]];
    Rw[static_cast<size_t>(s) * ldr + R.list[k]] = acc;
}

__global__ __launch_bounds__(BD_WG) void k_bd_sub(int64_t rows, int64_t ncols, double *__restrict__ W, int64_t ldw, const double *__restrict__ Rw,
                                                  int64_t ldr) {
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x; t < rows * ncols; t += static_cast<int64_t>(gridDim.x) * BD_WG) {
        const int64_t s = t / rows, p = t - s * rows;
        const double r = Rw[static_cast<size_t>(s) * ldr + p];
        if (r != 0.0) W[static_cast<size_t>(s) * ldw + p] -= r;
    }
}

// v2[j] -= sum_e val[e] z[idx[e]] (border column j);  with list: out[list[k]] = sum_e val[e] y[idx[e]]
__global__ __launch_bounds__(BD_WG) void k_bd_gather_vec(SxRowsDev R, const double *__restrict__ in, double *__restrict__ out, int subtract) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x;
    if (k >= R.nrows) return;
    double acc = 0.0;
    for (int64_t e = R.ptr[k]; e < R.ptr[k + 1]; ++e) acc += R.val[e] * in[R.idx[e]];
    const int64_t o = R.list ? R.list[k] : k;
    if (subtract) out[o] -= acc;
    else out[o] = acc;
}

inline unsigned bd_grid(int64_t n) { return static_cast<unsigned>(n > 0 ? (n + BD_WG - 1) / BD_WG : 1); }

} // namespace

int SxBorderOps::ftran(double *W, int64_t ncols, bool sparse_rhs, bool upto_schur) {
    if (ncols <= 0) return SX_OK;
    hipStream_t s = ctx->stream;
    if (lu) {
        if (sparse_rhs && ncols > 1) SX_TRY(sx_bandlu_solve_sparse_dev(lu, ncols, W, mp, tiny));
        else SX_TRY(sx_bandlu_solve_dev(lu, 0, ncols, W, mp));
    }
    if (nb == 0) return SX_OK;
    if (b21_rows.nrows > 0 && b21_rows.idx)
        hipLaunchKernelGGL(k_bd_rows_apply, dim3(bd_grid(b21_rows.nrows * ncols * 64)), dim3(BD_WG), 0, s, b21_rows, ncols, W, mp, m1);
    SX_HIP(hipGetLastError());
    if (upto_schur) return SX_OK;
    SX_REQUIRE(dl != nullptr, "bordered basis: the Schur complement is not factored");
    SX_TRY(sx_denselu_solve_dev(dl, 0, ncols, W + m1, mp));
    if (lu && b12_rows.nrows > 0) {
        SX_REQUIRE(work && work_cols > 0, "bordered basis: no work block");
        for (int64_t c0 = 0; c0 < ncols; c0 += work_cols) {
            const int64_t kc = std::min<int64_t>(work_cols, ncols - c0);
            double *Wc = W + static_cast<size_t>(c0) * mp;
            SX_HIP(hipMemsetAsync(work, 0, sizeof(double) * static_cast<size_t>(m1) * kc, s));
            hipLaunchKernelGGL(k_bd_b12_apply, dim3(bd_grid(b12_rows.nrows * kc)), dim3(BD_WG), 0, s, b12_rows, kc, Wc + m1, mp, work, m1);
            SX_TRY(sx_bandlu_solve_dev(lu, 0, kc, work, m1));
            hipLaunchKernelGGL(k_bd_sub, dim3(static_cast<unsigned>(std::min<int64_t>(bd_grid(m1 * kc), 1 << 20))), dim3(BD_WG), 0, s, m1, kc, Wc, mp, work, m1);
        }
        SX_HIP(hipGetLastError());
    }
    return SX_OK;
}

int SxBorderOps::btran(double *v) {
    hipStream_t s = ctx->stream;
    if (lu) SX_TRY(sx_bandlu_solve_dev(lu, 1, 1, v, mp));
    if (nb == 0) return SX_OK;
    SX_REQUIRE(dl != nullptr, "bordered basis: the Schur complement is not factored");
    if (lu && b12_cols.nrows > 0) hipLaunchKernelGGL(k_bd_gather_vec, dim3(bd_grid(b12_cols.nrows)), dim3(BD_WG), 0, s, b12_cols, v, v + m1, 1);
    SX_TRY(sx_denselu_solve_dev(dl, 1, 1, v + m1, mp));
    if (lu && b21_cols.nrows > 0) {
        SX_REQUIRE(work && work_cols > 0, "bordered basis: no work block");
        SX_HIP(hipMemsetAsync(work, 0, sizeof(double) * static_cast<size_t>(m1), s));
        hipLaunchKernelGGL(k_bd_gather_vec, dim3(bd_grid(b21_cols.nrows)), dim3(BD_WG), 0, s, b21_cols, v + m1, work, 0);
        SX_TRY(sx_bandlu_solve_dev(lu, 1, 1, work, m1));
        hipLaunchKernelGGL(k_bd_sub, dim3(bd_grid(m1)), dim3(BD_WG), 0, s, m1, 1, v, mp, work, m1);
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}
