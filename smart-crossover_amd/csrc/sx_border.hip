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

// first and one past the last row of column s that holds more than eps in magnitude (lo = rows, hi = 0: none); a workgroup per column
__global__ __launch_bounds__(BD_WG) void k_bd_col_window(int64_t rows, const double *__restrict__ W, int64_t ldw, double eps, int32_t *__restrict__ lo,
                                                         int32_t *__restrict__ hi) {
    __shared__ int slo[BD_WG], shi[BD_WG];
    const double *col = W + static_cast<size_t>(blockIdx.x) * ldw;
    int l = static_cast<int>(rows), h = 0;
    for (int64_t p = threadIdx.x; p < rows; p += BD_WG)
        if (fabs(col[p]) > eps) {
            l = l < static_cast<int>(p) ? l : static_cast<int>(p);
            h = static_cast<int>(p) + 1; // (ascending per lane)
        }
    slo[threadIdx.x] = l;
    shi[threadIdx.x] = h;
    __syncthreads();
    for (int o = BD_WG / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            slo[threadIdx.x] = slo[threadIdx.x] < slo[threadIdx.x + o] ? slo[threadIdx.x] : slo[threadIdx.x + o];
            shi[threadIdx.x] = shi[threadIdx.x] > shi[threadIdx.x + o] ? shi[threadIdx.x] : shi[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lo[blockIdx.x] = slo[0];
        hi[blockIdx.x] = shi[0];
    }
}

// k_bd_rows_apply for columns whose band part lives in [lo[s], hi[s]): a row's entries are in ascending position order, so
// the ones inside the window are found by a bisection (most (row, column) pairs meet nowhere)
__global__ __launch_bounds__(BD_WG) void k_bd_rows_apply_win(SxRowsDev R, int64_t ncols, double *__restrict__ W, int64_t mp, int64_t m1,
                                                             const int32_t *__restrict__ lo, const int32_t *__restrict__ hi) {
    const int64_t wv = (static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wv >= R.nrows * ncols) return;
    const int64_t k = wv % R.nrows, s = wv / R.nrows;
    const int l = lo[s], h = hi[s];
    if (l >= h) return;
    int64_t a = R.ptr[k], b = R.ptr[k + 1];
    if (a >= b || R.idx[b - 1] < l || R.idx[a] >= h) return;
    while (a < b) { // first entry with position >= l
        const int64_t mid = (a + b) >> 1;
        if (R.idx[mid] < l) a = mid + 1;
        else b = mid;
    }
    const int64_t e1 = R.ptr[k + 1];
    double *col = W + static_cast<size_t>(s) * mp;
    double acc = 0.0;
    for (int64_t e = a + lane; e < e1; e += 64) {
        const int p = R.idx[e];
        if (p >= h) break;
        acc += R.val[e] * col[p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0 && acc != 0.0) col[m1 + k] -= acc;
}

// the same with a LANE per (border row, column): the few entries a row has inside a column's window (none, for most pairs)
// are added one after the other -- 64 times fewer waves than a wave per pair when the windows are short
__global__ __launch_bounds__(BD_WG) void k_bd_rows_apply_win_lane(SxRowsDev R, int64_t ncols, double *__restrict__ W, int64_t mp, int64_t m1,
                                                                  const int32_t *__restrict__ lo, const int32_t *__restrict__ hi) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x;
    if (t >= R.nrows * ncols) return;
    const int64_t k = t % R.nrows, s = t / R.nrows;
    const int l = lo[s], h = hi[s];
    if (l >= h) return;
    int64_t a = R.ptr[k], b = R.ptr[k + 1];
    if (a >= b || R.idx[b - 1] < l || R.idx[a] >= h) return;
    const int64_t e1 = b;
    while (a < b) { // first entry with position >= l
        const int64_t mid = (a + b) >> 1;
        if (R.idx[mid] < l) a = mid + 1;
        else b = mid;
    }
    double *col = W + static_cast<size_t>(s) * mp;
    double acc = 0.0;
    for (int64_t e = a; e < e1; ++e) {
        const int p = R.idx[e];
        if (p >= h) break;
        acc += R.val[e] * col[p];
    }
    if (acc != 0.0) col[m1 + k] -= acc;
}

// V[off[c0 + s] ...] = W[lo .. hi, s]
__global__ __launch_bounds__(BD_WG) void k_bd_pack(const double *__restrict__ W, int64_t ldw, const int32_t *__restrict__ lo, const int32_t *__restrict__ hi,
                                                   const int64_t *__restrict__ off, double *__restrict__ V) {
    const int s = blockIdx.x;
    const int l = lo[s], h = hi[s];
    const double *col = W + static_cast<size_t>(s) * ldw;
    double *dst = V + off[s];
    for (int p = l + threadIdx.x; p < h; p += BD_WG) dst[p - l] = col[p];
}

// W[p, s] -= sum_j V[p, j] X2[j, s] over the border columns j whose window meets the block of 256 rows (blk_ptr / blk_col);
// a lane per row, 8 columns s per workgroup (blockIdx.y), the x2 entries of the block's columns through LDS
constexpr int BD_VS = 8;
__global__ __launch_bounds__(BD_WG) void k_bd_v_apply(int64_t m1, int64_t ncols, const int64_t *__restrict__ blk_ptr, const int32_t *__restrict__ blk_col,
                                                      const int32_t *__restrict__ vlo, const int32_t *__restrict__ vhi, const int64_t *__restrict__ voff,
                                                      const double *__restrict__ V, const double *__restrict__ X2, int64_t ldx, double *__restrict__ W,
                                                      int64_t ldw) {
    __shared__ double sx[64][BD_VS];
    __shared__ int sl[64], sh_[64];
    __shared__ int64_t so[64];
    const int64_t p = static_cast<int64_t>(blockIdx.x) * BD_WG + threadIdx.x;
    const int64_t s0 = static_cast<int64_t>(blockIdx.y) * BD_VS;
    const int ns = static_cast<int>((ncols - s0 < BD_VS) ? ncols - s0 : BD_VS);
    double acc[BD_VS];
#pragma unroll
    for (int q = 0; q < BD_VS; ++q) acc[q] = 0.0;
    const int64_t e0 = blk_ptr[blockIdx.x], e1 = blk_ptr[blockIdx.x + 1];
    for (int64_t base = e0; base < e1; base += 64) {
        const int cnt = static_cast<int>((e1 - base < 64) ? e1 - base : 64);
        __syncthreads();
        if (threadIdx.x < cnt) {
            const int j = blk_col[base + threadIdx.x];
            sl[threadIdx.x] = vlo[j];
            sh_[threadIdx.x] = vhi[j];
            so[threadIdx.x] = voff[j];
        }
        for (int e = threadIdx.x; e < cnt * BD_VS; e += BD_WG) {
            const int c = e / BD_VS, q = e % BD_VS;
            sx[c][q] = (q < ns) ? X2[blk_col[base + c] + static_cast<size_t>(s0 + q) * ldx] : 0.0;
        }
        __syncthreads();
        if (p < m1)
            for (int c = 0; c < cnt; ++c) {
                if (p < sl[c] || p >= sh_[c]) continue;
                const double v = V[so[c] + (p - sl[c])];
#pragma unroll
                for (int q = 0; q < BD_VS; ++q) acc[q] += v * sx[c][q];
            }
    }
    if (p < m1)
        for (int q = 0; q < ns; ++q)
            if (acc[q] != 0.0) W[p + static_cast<size_t>(s0 + q) * ldw] -= acc[q];
}

inline unsigned bd_grid(int64_t n) { return static_cast<unsigned>(n > 0 ? (n + BD_WG - 1) / BD_WG : 1); }

} // namespace

SxBorderOps::~SxBorderOps() {
    for (void *q : {(void *)win_lo, (void *)v_val, (void *)d_v_lo, (void *)d_v_off, (void *)d_vb_ptr, (void *)d_vb_col}) (void)sx_dfree(q);
}

int SxBorderOps::ftran(double *W, int64_t ncols, bool sparse_rhs, bool upto_schur) {
    if (ncols <= 0) return SX_OK;
    hipStream_t s = ctx->stream;
    const bool windows = sparse_rhs && lu && nb > 0;
    if (lu) {
        if (sparse_rhs && ncols > 1) SX_TRY(sx_bandlu_solve_sparse_dev(lu, ncols, W, mp, tiny));
        else SX_TRY(sx_bandlu_solve_dev(lu, 0, ncols, W, mp));
    }
    if (nb == 0) return SX_OK;
    if (windows) {
        if (ncols > win_cap) {
            SX_HIP(hipStreamSynchronize(s));
            (void)sx_dfree(win_lo);
            win_lo = win_hi = nullptr;
            win_cap = 0;
            SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&win_lo), sizeof(int32_t) * 2 * static_cast<size_t>(ncols)));
            win_hi = win_lo + ncols;
            win_cap = ncols;
        }
        hipLaunchKernelGGL(k_bd_col_window, dim3(static_cast<unsigned>(ncols)), dim3(BD_WG), 0, s, m1, W, mp, SX_BORDER_WIN_EPS, win_lo, win_lo + win_cap);
        win_hi = win_lo + win_cap;
    }
    if (b21_rows.nrows > 0 && b21_rows.ptr) {
        if (windows && ncols >= 64)
            hipLaunchKernelGGL(k_bd_rows_apply_win_lane, dim3(bd_grid(b21_rows.nrows * ncols)), dim3(BD_WG), 0, s, b21_rows, ncols, W, mp, m1, win_lo, win_hi);
        else if (windows)
            hipLaunchKernelGGL(k_bd_rows_apply_win, dim3(bd_grid(b21_rows.nrows * ncols * 64)), dim3(BD_WG), 0, s, b21_rows, ncols, W, mp, m1, win_lo, win_hi);
        else hipLaunchKernelGGL(k_bd_rows_apply, dim3(bd_grid(b21_rows.nrows * ncols * 64)), dim3(BD_WG), 0, s, b21_rows, ncols, W, mp, m1);
    }
    SX_HIP(hipGetLastError());
    if (upto_schur) return SX_OK;
    SX_REQUIRE(dl != nullptr, "bordered basis: the Schur complement is not factored");
    SX_TRY(sx_denselu_solve_dev(dl, 0, ncols, W + m1, mp));
    if (lu && b12_rows.nrows > 0) {
        if (v_ready) { // x1 = w1 - V x2
            hipLaunchKernelGGL(k_bd_v_apply, dim3(static_cast<unsigned>(v_nblk), static_cast<unsigned>((ncols + BD_VS - 1) / BD_VS)), dim3(BD_WG), 0, s, m1, ncols,
                               d_vb_ptr, d_vb_col, d_v_lo, d_v_hi, d_v_off, v_val, W + m1, mp, W, mp);
            SX_HIP(hipGetLastError());
            return SX_OK;
        }
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

int SxBorderOps::pack_v(const double *W, int64_t ncols, const int32_t *dest) {
    if (v_failed || !lu || nb == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    if (v_lo.empty()) {
        v_lo.assign(static_cast<size_t>(nb), 0);
        v_hi.assign(static_cast<size_t>(nb), 0);
        v_off.assign(static_cast<size_t>(nb), 0);
    }
    SX_REQUIRE(dest && ncols <= nb && ncols <= win_cap, "bordered basis: bad block of border columns");
    std::vector<int32_t> lo(static_cast<size_t>(ncols)), hi(static_cast<size_t>(ncols));
    SX_HIP(hipMemcpyAsync(lo.data(), win_lo, sizeof(int32_t) * static_cast<size_t>(ncols), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(hi.data(), win_hi, sizeof(int32_t) * static_cast<size_t>(ncols), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    size_t need = v_used;
    std::vector<int64_t> off(static_cast<size_t>(ncols));
    for (int64_t t = 0; t < ncols; ++t) {
        const int32_t j = dest[t];
        SX_REQUIRE(j >= 0 && j < nb, "bordered basis: bad border column");
        if (hi[t] <= lo[t]) lo[t] = hi[t] = 0;
        v_lo[j] = lo[t];
        v_hi[j] = hi[t];
        v_off[j] = off[t] = static_cast<int64_t>(need);
        need += static_cast<size_t>(hi[t] - lo[t]);
    }
    if (need > v_cap) {
        // no room: grow up to the limit the caller allows (v_limit doubles), else give the packed form up
        size_t want = std::max<size_t>(need + need / 2, 1 << 20);
        if (want > v_limit) want = v_limit;
        if (need > want) {
            v_failed = 1;
            return SX_OK;
        }
        double *nv = nullptr;
        if (sx_dmalloc(reinterpret_cast<void **>(&nv), sizeof(double) * want) != hipSuccess) {
            v_failed = 1;
            return SX_OK;
        }
        if (v_used) SX_HIP(hipMemcpyAsync(nv, v_val, sizeof(double) * v_used, hipMemcpyDeviceToDevice, s));
        SX_HIP(hipStreamSynchronize(s));
        (void)sx_dfree(v_val);
        v_val = nv;
        v_cap = want;
    }
    int64_t *d_off = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d_off), sizeof(int64_t) * static_cast<size_t>(ncols)));
    SX_HIP(hipMemcpyAsync(d_off, off.data(), sizeof(int64_t) * static_cast<size_t>(ncols), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bd_pack, dim3(static_cast<unsigned>(ncols)), dim3(BD_WG), 0, s, W, mp, win_lo, win_hi, d_off, v_val);
    SX_HIP(hipStreamSynchronize(s));
    (void)sx_dfree(d_off);
    v_used = need;
    return SX_OK;
}

int SxBorderOps::finish_v(const std::vector<int32_t> &zero) {
    v_ready = 0;
    if (v_failed || !lu || nb == 0 || v_lo.empty()) return SX_OK;
    hipStream_t s = ctx->stream;
    for (int64_t j = 0; j < nb; ++j)
        if (!zero.empty() && zero[j]) v_lo[j] = v_hi[j] = 0;
    v_nblk = (m1 + BD_WG - 1) / BD_WG;
    std::vector<int64_t> ptr(static_cast<size_t>(v_nblk) + 1, 0);
    for (int64_t j = 0; j < nb; ++j)
        if (v_hi[j] > v_lo[j])
            for (int64_t bk = v_lo[j] / BD_WG; bk <= (v_hi[j] - 1) / BD_WG; ++bk) ++ptr[bk + 1];
    for (int64_t bk = 0; bk < v_nblk; ++bk) ptr[bk + 1] += ptr[bk];
    std::vector<int32_t> col(static_cast<size_t>(ptr[v_nblk]));
    std::vector<int64_t> at(ptr.begin(), ptr.end() - 1);
    for (int64_t j = 0; j < nb; ++j) // (ascending j within a block: the order of the additions)
        if (v_hi[j] > v_lo[j])
            for (int64_t bk = v_lo[j] / BD_WG; bk <= (v_hi[j] - 1) / BD_WG; ++bk) col[at[bk]++] = static_cast<int32_t>(j);
    for (void *q : {(void *)d_v_lo, (void *)d_v_off, (void *)d_vb_ptr, (void *)d_vb_col}) (void)sx_dfree(q);
    d_v_lo = d_v_hi = d_vb_col = nullptr;
    d_v_off = d_vb_ptr = nullptr;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d_v_lo), sizeof(int32_t) * 2 * static_cast<size_t>(nb)));
    d_v_hi = d_v_lo + nb;
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d_v_off), sizeof(int64_t) * static_cast<size_t>(nb)));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d_vb_ptr), sizeof(int64_t) * ptr.size()));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&d_vb_col), sizeof(int32_t) * std::max<size_t>(col.size(), 1)));
    SX_HIP(hipMemcpyAsync(d_v_lo, v_lo.data(), sizeof(int32_t) * static_cast<size_t>(nb), hipMemcpyHostToDevice, s));
    SX_HIP(hipMemcpyAsync(d_v_hi, v_hi.data(), sizeof(int32_t) * static_cast<size_t>(nb), hipMemcpyHostToDevice, s));
    SX_HIP(hipMemcpyAsync(d_v_off, v_off.data(), sizeof(int64_t) * static_cast<size_t>(nb), hipMemcpyHostToDevice, s));
    SX_HIP(hipMemcpyAsync(d_vb_ptr, ptr.data(), sizeof(int64_t) * ptr.size(), hipMemcpyHostToDevice, s));
    if (!col.empty()) SX_HIP(hipMemcpyAsync(d_vb_col, col.data(), sizeof(int32_t) * col.size(), hipMemcpyHostToDevice, s));
    SX_HIP(hipStreamSynchronize(s));
    v_ready = 1;
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
