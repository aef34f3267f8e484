// Dense LU with partial pivoting on the device (kernel group K16g): the factorisation of the SCHUR COMPLEMENT of the
// bordered basis of the sparse crossover (sx_border.hip) -- the linking rows of a staged LP, the rows the band matching
// left over and the separators between the band's blocks, a few thousand to ~16,000 of them.  The reference leaves
// every basis factorisation to Gurobi / CPLEX / Mosek (solver_caller/gurobi.py:202-210 model.optimize()).
//
//   storage   column major, n x n, leading dimension ld (n rounded up to 16 doubles);
//   factor    right-looking, outer blocks of 64 columns, inner panels of 8:
//               k_dl_panel    ONE workgroup of 1,024 lanes, a row per lane and slot (up to 16 slots: 16,384 rows); per column:
//                             arg-max as one 64-bit key per row (value bits | 16383 - row) reduced by DPP moves and one LDS
//                             atomic per wave, the two rows of the swap exchanged through LDS by their owner lanes, multipliers,
//                             rank-one update of the panel's other columns.  A lane is the only one that ever touches its rows,
//                             so the panel lives in L2 without a fence;
//               k_dl_swap     the panel's row swaps in every other column; for the columns of the outer block to the panel's
//                             right also their 8 rows of U (forward substitution with the panel's unit triangle);
//               k_dl_inblock  rows below the panel in those columns: A22 -= L21 U12 (8 multiply-adds per entry);
//               per outer block: k_dl_tri (U12 = L11^-1 A12, a lane per column, the 64 x 64 triangle broadcast from LDS) and
//               k_dl_gemm (A22 -= L21 U12 on the fp64 matrix cores: v_mfma_f64_16x16x4_f64, 64 x 64 tiles of C per workgroup,
//               operands staged through LDS in two halves of K; the tile is computed TRANSPOSED so that a lane's four
//               results sit in four columns of C at one row -- 16 lanes write 128 contiguous bytes).
//             A column without a usable pivot is REPLACED by the unit vector of the row on its diagonal (as in sx_bandlu.hip):
//             the caller learns which and puts that row's logical into the basis;
//   solves    row permutation by a gather, then per outer block a 64 x 64 triangle (k_dl_tri) and one k_dl_gemm for the
//             rows the block reaches; both orientations.
// fp64.  No atomics on data, fixed arithmetic order per entry: deterministic.
#include "sx_internal.h"
#include "sx_wave.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

struct sx_denselu {
    sx_ctx *ctx = nullptr;
    int64_t n = 0, ld = 0;
    double *a = nullptr;
    int32_t *ipiv = nullptr; // [n] row swapped with j at step j
    int32_t *rep = nullptr;  // [n] 1: column j was replaced by a unit vector
    int32_t *perm = nullptr; // [n] original row at position i once all swaps are made
    double *tmp = nullptr;   // copy of the right-hand sides for the permutation
    size_t tmp_cap = 0;
    bool factored = false;
};

namespace {

constexpr int DL_NB = 64;   // outer block
constexpr int DL_PW = 8;    // inner panel
constexpr int DL_PT = 1024; // lanes of the panel kernel

// ------------------------------------------------------------------------------------------- panel
template <int RPT>
__global__ __launch_bounds__(DL_PT) void k_dl_panel(double *__restrict__ A, int64_t ld, int n, int jb, int ncol, double tol,
                                                    int32_t *__restrict__ ipiv, int32_t *__restrict__ rep) {
    __shared__ double sA[DL_PW], sB[DL_PW];
    __shared__ unsigned long long skey[DL_PW];
    const int tid = threadIdx.x;
    const int R = n - jb;
    double *P = A + static_cast<size_t>(jb) * ld + jb; // P[r + q ld] = entry (jb + r, jb + q)
    if (tid < DL_PW) skey[tid] = 0;
    __syncthreads();
    for (int c = 0; c < ncol; ++c) {
        double *col = P + static_cast<size_t>(c) * ld;
        double v[RPT];
        unsigned long long key = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * DL_PT;
            v[k] = 0.0;
            if (r >= c && r < R) {
                v[k] = col[r];
                const unsigned long long kk = (static_cast<unsigned long long>(__double_as_longlong(fabs(v[k]))) & ~0x3FFFull) |
                                              static_cast<unsigned long long>(16383 - r);
                key = kk > key ? kk : key;
            }
        }
        key = sx_wave_max_u64(key);
        if ((tid & 63) == 0) atomicMax(&skey[c], key);
        __syncthreads();
        key = skey[c];
        const int p = 16383 - static_cast<int>(key & 0x3FFFull);
        const bool bad = !(__longlong_as_double(static_cast<long long>(key & ~0x3FFFull)) > tol); // (NaN counts as unusable)
        if (bad) { // no usable pivot: the column becomes the unit vector of the row on its diagonal
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * DL_PT;
                if (r >= c && r < R) col[r] = (r == c) ? 1.0 : 0.0;
            }
            double *gcol = A + static_cast<size_t>(jb + c) * ld;
            for (int i = tid; i < jb + c; i += DL_PT) gcol[i] = 0.0;
            if (tid == 0) {
                ipiv[jb + c] = jb + c;
                rep[jb + c] = 1;
            }
            continue; // (uniform)
        }
        if (tid == 0) {
            ipiv[jb + c] = jb + p;
            rep[jb + c] = 0;
        }
        // ---- rows c and p change places in the panel's columns: each row is read and written by its owner lane only
        if (tid == (p & (DL_PT - 1))) {
            for (int q = 0; q < ncol; ++q) sB[q] = P[p + static_cast<size_t>(q) * ld];
        }
        if (p != c && tid == (c & (DL_PT - 1))) {
            for (int q = 0; q < ncol; ++q) sA[q] = P[c + static_cast<size_t>(q) * ld];
        }
        __syncthreads();
        if (p != c) {
            if (tid == (c & (DL_PT - 1))) {
                for (int q = 0; q < ncol; ++q) P[c + static_cast<size_t>(q) * ld] = sB[q];
            }
            if (tid == (p & (DL_PT - 1))) {
                for (int q = 0; q < ncol; ++q) P[p + static_cast<size_t>(q) * ld] = sA[q];
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * DL_PT;
                if (r == c) v[k] = sB[c];
                if (r == p) v[k] = sA[c];
            }
        }
        const double piv = sB[c];
        // ---- multipliers
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * DL_PT;
            if (r > c && r < R) {
                v[k] = v[k] / piv;
                col[r] = v[k];
            } else {
                v[k] = 0.0;
            }
        }
        // ---- rank-one update of the panel's columns to the right
        for (int q = c + 1; q < ncol; ++q) {
            const double u = sB[q]; // (row c after the swap)
            if (u == 0.0) continue; // (uniform)
            double *cq = P + static_cast<size_t>(q) * ld;
            double t[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * DL_PT;
                t[k] = (r > c && r < R) ? cq[r] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = tid + k * DL_PT;
                if (r > c && r < R && v[k] != 0.0) cq[r] = t[k] - v[k] * u;
            }
        }
        __syncthreads(); // sA / sB are rewritten by the next column
    }
}

// the panel's row swaps in every column outside the panel; the columns of the outer block to the panel's right also get
// their rows of U: x <- L11^-1 x with the panel's unit lower triangle
__global__ __launch_bounds__(256) void k_dl_swap(double *__restrict__ A, int64_t ld, int n, int jb, int ncol, int blk_end,
                                                 const int32_t *__restrict__ ipiv) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n || (j >= jb && j < jb + ncol)) return;
    double *col = A + static_cast<size_t>(j) * ld;
    for (int c = 0; c < ncol; ++c) {
        const int p = ipiv[jb + c];
        if (p != jb + c) {
            const double a = col[jb + c];
            col[jb + c] = col[p];
            col[p] = a;
        }
    }
    if (j >= jb + ncol && j < blk_end) {
        double x[DL_PW];
#pragma unroll
        for (int c = 0; c < DL_PW; ++c) x[c] = (c < ncol) ? col[jb + c] : 0.0;
#pragma unroll
        for (int c = 0; c < DL_PW; ++c) {
            if (c < ncol) {
                const double *L = A + static_cast<size_t>(jb + c) * ld + jb;
#pragma unroll
                for (int r = c + 1; r < DL_PW; ++r)
                    if (r < ncol) x[r] = x[r] - L[r] * x[c];
            }
        }
#pragma unroll
        for (int c = 0; c < DL_PW; ++c)
            if (c < ncol) col[jb + c] = x[c];
    }
}

// rows below the panel, columns of the outer block to its right: A[r, j] -= sum_c L[r, jb + c] U[jb + c, j]
__global__ __launch_bounds__(256) void k_dl_inblock(double *__restrict__ A, int64_t ld, int n, int jb, int ncol, int blk_end) {
    __shared__ double sU[DL_PW][DL_NB];
    const int nq = blk_end - (jb + ncol);
    for (int e = threadIdx.x; e < DL_PW * nq; e += 256) {
        const int c = e % DL_PW, q = e / DL_PW;
        sU[c][q] = (c < ncol) ? A[static_cast<size_t>(jb + ncol + q) * ld + jb + c] : 0.0;
    }
    __syncthreads();
    const int r = jb + ncol + blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double l[DL_PW];
#pragma unroll
    for (int c = 0; c < DL_PW; ++c) l[c] = (c < ncol) ? A[static_cast<size_t>(jb + c) * ld + r] : 0.0;
    for (int q = 0; q < nq; ++q) {
        double *t = A + static_cast<size_t>(jb + ncol + q) * ld + r;
        double acc = *t;
#pragma unroll
        for (int c = 0; c < DL_PW; ++c) acc = acc - l[c] * sU[c][q];
        *t = acc;
    }
}

// ------------------------------------------------------------------------------------------- 64 x 64 triangles
// X[0:nb, t] <- T^-1 X[0:nb, t] for ncols columns; T = the triangle of the nb x nb block at `blk` that the sweep uses:
// FWD & !TRANS: L (unit), !FWD & !TRANS: U, FWD & TRANS: U^T, !FWD & TRANS: L^T (unit).  A wave per column, a lane per
// unknown with its row of T in registers: x_c is final when its turn comes and reaches the other lanes through v_readlane
// (no LDS round trip in the chain); the waves of a workgroup stride over the columns.
template <bool FWD, bool TRANS, bool UNIT>
__global__ __launch_bounds__(256) void k_dl_tri(const double *__restrict__ blk, int64_t ld, int nb, double *__restrict__ X, int64_t ldx,
                                                int64_t ncols) {
    __shared__ double sM[DL_NB][DL_NB + 1];
    for (int e = threadIdx.x; e < DL_NB * DL_NB; e += 256) {
        const int r = e % DL_NB, c = e / DL_NB;
        sM[r][c] = (r < nb && c < nb) ? blk[r + static_cast<size_t>(c) * ld] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double m[DL_NB];
#pragma unroll
    for (int c = 0; c < DL_NB; ++c) m[c] = TRANS ? sM[c][lane] : sM[lane][c];
    double dinv = 1.0;
    if (!UNIT) dinv = 1.0 / sM[lane][lane];
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * 4 + wave; t < ncols; t += static_cast<int64_t>(gridDim.x) * 4) {
        double *xc = X + static_cast<size_t>(t) * ldx;
        double x = (lane < nb) ? xc[lane] : 0.0;
#pragma unroll
        for (int s = 0; s < DL_NB; ++s) {
            const int c = FWD ? s : DL_NB - 1 - s;
            if (!UNIT) {
                if (lane == c) x = x * dinv;
            }
            const double v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), c), __builtin_amdgcn_readlane(__double2loint(x), c));
            if (FWD ? lane > c : lane < c) x = x - m[c] * v;
        }
        if (lane < nb) xc[lane] = x;
    }
}

// ------------------------------------------------------------------------------------------- C -= A B on the matrix cores
// C[i, j] -= sum_{k < K} Am[i, k] B[k, j], K <= 64;  Am[i, k] = A[i + k lda] (TRANSA: A[k + i lda]);  B[k + j ldb];  C[i + j ldc].
// A workgroup owns a 64 x 64 tile of C; wave w the 32 x 32 quarter (w >> 1: columns, w & 1: rows) as 2 x 2 MFMA tiles.  The MFMA
// computes the tile TRANSPOSED: its A operand is B^T (row = column j of C), its B operand Am^T (col = row i of C), so that
// D[row = j, col = i] puts a lane's four results at one row i of four columns j.
typedef double dl_v4d __attribute__((ext_vector_type(4)));
constexpr int DL_KH = 32;              // K half staged at a time
constexpr int DL_SA = 80;              // sA[k][i]: rows of 80 doubles (4 consecutive k: offsets 0, 16, 32, 48 mod 64)
constexpr int DL_ST = DL_KH + 4;       // sBt[j][k] / sAt[i][k]: rows of 36 doubles (16 consecutive j land on distinct slots mod 64)
template <bool TRANSA>
__global__ __launch_bounds__(256) void k_dl_gemm(int M, int N, int K, const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                 int64_t ldb, double *__restrict__ C, int64_t ldc) {
    __shared__ double sA[TRANSA ? DL_NB * DL_ST : DL_KH * DL_SA];
    __shared__ double sBt[DL_NB * DL_ST];
    const int i0 = blockIdx.x * DL_NB, j0 = blockIdx.y * DL_NB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wj = wave >> 1, wi = wave & 1;
    dl_v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = dl_v4d{0.0, 0.0, 0.0, 0.0};
    for (int kh = 0; kh < K; kh += DL_KH) {
        if (kh) __syncthreads();
        if (!TRANSA) {
            for (int e = tid; e < DL_KH * DL_NB; e += 256) { // i fastest: coalesced, conflict-free
                const int i = e % DL_NB, k = e / DL_NB;
                sA[k * DL_SA + i] = (i0 + i < M && kh + k < K) ? A[static_cast<size_t>(kh + k) * lda + i0 + i] : 0.0;
            }
        } else {
            for (int e = tid; e < DL_KH * DL_NB; e += 256) { // k fastest
                const int k = e % DL_KH, i = e / DL_KH;
                sA[i * DL_ST + k] = (i0 + i < M && kh + k < K) ? A[static_cast<size_t>(i0 + i) * lda + kh + k] : 0.0;
            }
        }
        for (int e = tid; e < DL_KH * DL_NB; e += 256) { // k fastest
            const int k = e % DL_KH, j = e / DL_KH;
            sBt[j * DL_ST + k] = (j0 + j < N && kh + k < K) ? B[static_cast<size_t>(j0 + j) * ldb + kh + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < DL_KH; k0 += 4) {
            const double a0 = sBt[(32 * wj + l15) * DL_ST + k0 + kq], a1 = sBt[(32 * wj + 16 + l15) * DL_ST + k0 + kq];
            double b0, b1;
            if (!TRANSA) {
                b0 = sA[(k0 + kq) * DL_SA + 32 * wi + l15];
                b1 = sA[(k0 + kq) * DL_SA + 32 * wi + 16 + l15];
            } else {
                b0 = sA[(32 * wi + l15) * DL_ST + k0 + kq];
                b1 = sA[(32 * wi + 16 + l15) * DL_ST + k0 + kq];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // D layout of the f64 form: col = lane & 15 (-> row i of C), row = (lane >> 4) + 4 reg (-> column j of C)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = j0 + 32 * wj + 16 * tj + kq + 4 * reg, i = i0 + 32 * wi + 16 * ti + l15;
                if (i < M && j < N) {
                    double *c = C + static_cast<size_t>(j) * ldc + i;
                    *c = *c - acc[tj][ti][reg];
                }
            }
}

// dst[i, t] = src[perm[i], t]  (GATHER)  /  dst[perm[i], t] = src[i, t]  (!GATHER)
template <bool GATHER>
__global__ __launch_bounds__(256) void k_dl_permute(int64_t n, int64_t ncols, const int32_t *__restrict__ perm, const double *__restrict__ src,
                                                    int64_t lds_, double *__restrict__ dst, int64_t ldd) {
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < n * ncols; e += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t t = e / n, i = e - t * n;
        if (GATHER) dst[i + t * ldd] = src[perm[i] + t * lds_];
        else dst[perm[i] + t * ldd] = src[i + t * lds_];
    }
}

inline unsigned dl_trigrid(int64_t ncols) { return static_cast<unsigned>(std::min<int64_t>((ncols + 3) / 4, 2048)); }
inline unsigned dl_grid(int64_t work, int per) { return static_cast<unsigned>(work > 0 ? (work + per - 1) / per : 1); }

template <bool TRANSA>
void dl_gemm(hipStream_t s, int64_t M, int64_t N, int K, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    hipLaunchKernelGGL((k_dl_gemm<TRANSA>), dim3(dl_grid(M, DL_NB), dl_grid(N, DL_NB)), dim3(256), 0, s, static_cast<int>(M), static_cast<int>(N), K, A,
                       lda, B, ldb, C, ldc);
}

} // namespace

SX_API int sx_denselu_create_dev(sx_ctx *ctx, int64_t n, sx_denselu **out) {
    SX_ENTER(ctx);
    SX_REQUIRE(out && n > 0 && n <= 16384, "dense LU: n must be in [1, 16384]");
    sx_denselu *h = new (std::nothrow) sx_denselu();
    SX_REQUIRE(h != nullptr, "out of host memory");
    h->ctx = ctx;
    h->n = n;
    h->ld = (n + 15) / 16 * 16;
    struct Guard {
        sx_denselu *h;
        ~Guard() {
            if (h) {
                (void)sx_dfree(h->a);
                (void)sx_dfree(h->ipiv);
                delete h;
            }
        }
    } guard{h};
    const size_t bytes = sizeof(double) * static_cast<size_t>(h->ld) * static_cast<size_t>(n);
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->a), bytes));
    SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->ipiv), sizeof(int32_t) * 3 * static_cast<size_t>(n)));
    h->rep = h->ipiv + n;
    h->perm = h->rep + n;
    SX_HIP(hipMemsetAsync(h->a, 0, bytes, ctx->stream));
    SX_HIP(hipMemsetAsync(h->ipiv, 0, sizeof(int32_t) * 3 * static_cast<size_t>(n), ctx->stream));
    guard.h = nullptr;
    *out = h;
    return SX_OK;
}

// (internal, sx_internal.h) where sx_border.hip writes the Schur complement
int sx_denselu_matrix(sx_denselu *h, double **a_dev, int64_t *ld) {
    SX_REQUIRE(h && a_dev && ld, "NULL argument");
    *a_dev = h->a;
    *ld = h->ld;
    return SX_OK;
}

SX_API int sx_denselu_set_dev(sx_denselu *h, const double *src, int64_t lds) {
    SX_REQUIRE(h != nullptr && src != nullptr && lds >= h->n, "bad argument");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(!h->factored, "already factored");
    SX_HIP(hipMemcpy2DAsync(h->a, sizeof(double) * h->ld, src, sizeof(double) * lds, sizeof(double) * h->n, static_cast<size_t>(h->n),
                            hipMemcpyDeviceToDevice, ctx->stream));
    return SX_OK;
}

SX_API int sx_denselu_destroy(sx_denselu *h) {
    if (!h) return SX_OK;
    sx_device_guard guard(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    (void)sx_dfree(h->a);
    (void)sx_dfree(h->ipiv);
    (void)sx_dfree(h->tmp);
    delete h;
    return SX_OK;
}

SX_API int sx_denselu_factor_dev(sx_denselu *h, double pivot_tol, int64_t *n_replaced_out, int32_t *replaced_host, int32_t *rowperm_host) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(!h->factored, "already factored");
    hipStream_t s = ctx->stream;
    const int n = static_cast<int>(h->n);
    const int64_t ld = h->ld;
    double *A = h->a;
    for (int j0 = 0; j0 < n; j0 += DL_NB) {
        const int nbk = std::min(DL_NB, n - j0), blk_end = j0 + nbk;
        for (int jb = j0; jb < blk_end; jb += DL_PW) {
            const int ncol = std::min(DL_PW, blk_end - jb);
            const int R = n - jb;
            if (R <= 2 * DL_PT) hipLaunchKernelGGL((k_dl_panel<2>), dim3(1), dim3(DL_PT), 0, s, A, ld, n, jb, ncol, pivot_tol, h->ipiv, h->rep);
            else if (R <= 4 * DL_PT) hipLaunchKernelGGL((k_dl_panel<4>), dim3(1), dim3(DL_PT), 0, s, A, ld, n, jb, ncol, pivot_tol, h->ipiv, h->rep);
            else if (R <= 8 * DL_PT) hipLaunchKernelGGL((k_dl_panel<8>), dim3(1), dim3(DL_PT), 0, s, A, ld, n, jb, ncol, pivot_tol, h->ipiv, h->rep);
            else hipLaunchKernelGGL((k_dl_panel<16>), dim3(1), dim3(DL_PT), 0, s, A, ld, n, jb, ncol, pivot_tol, h->ipiv, h->rep);
            hipLaunchKernelGGL(k_dl_swap, dim3(dl_grid(n, 256)), dim3(256), 0, s, A, ld, n, jb, ncol, blk_end, h->ipiv);
            if (jb + ncol < blk_end && jb + ncol < n)
                hipLaunchKernelGGL(k_dl_inblock, dim3(dl_grid(n - jb - ncol, 256)), dim3(256), 0, s, A, ld, n, jb, ncol, blk_end);
        }
        if (blk_end < n) {
            const int64_t rest = n - blk_end;
            hipLaunchKernelGGL((k_dl_tri<true, false, true>), dim3(dl_trigrid(rest)), dim3(256), 0, s, A + j0 + static_cast<size_t>(j0) * ld, ld, nbk,
                               A + j0 + static_cast<size_t>(blk_end) * ld, ld, rest);
            dl_gemm<false>(s, rest, rest, nbk, A + blk_end + static_cast<size_t>(j0) * ld, ld, A + j0 + static_cast<size_t>(blk_end) * ld, ld,
                           A + blk_end + static_cast<size_t>(blk_end) * ld, ld);
        }
    }
    SX_HIP(hipGetLastError());
    std::vector<int32_t> piv(static_cast<size_t>(n)), rep(static_cast<size_t>(n)), perm(static_cast<size_t>(n));
    SX_HIP(hipMemcpyAsync(piv.data(), h->ipiv, sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
    SX_HIP(hipMemcpyAsync(rep.data(), h->rep, sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    std::iota(perm.begin(), perm.end(), 0);
    for (int j = 0; j < n; ++j)
        if (piv[j] != j) std::swap(perm[j], perm[piv[j]]);
    SX_HIP(hipMemcpyAsync(h->perm, perm.data(), sizeof(int32_t) * static_cast<size_t>(n), hipMemcpyHostToDevice, s));
    SX_HIP(hipStreamSynchronize(s));
    h->factored = true;
    int64_t cnt = 0;
    for (int j = 0; j < n; ++j) cnt += rep[j] != 0;
    if (n_replaced_out) *n_replaced_out = cnt;
    if (replaced_host) std::memcpy(replaced_host, rep.data(), sizeof(int32_t) * static_cast<size_t>(n));
    if (rowperm_host) std::memcpy(rowperm_host, perm.data(), sizeof(int32_t) * static_cast<size_t>(n));
    return SX_OK;
}

// In-place solve of nrhs right-hand sides X (column major, ldx >= n): trans = 0: A x = b, 1: A^T x = b (A = the matrix with
// its replaced columns).  Stream-ordered.
SX_API int sx_denselu_solve_dev(sx_denselu *h, int trans, int64_t nrhs, double *X, int64_t ldx) {
    SX_REQUIRE(h != nullptr, "handle is NULL");
    sx_ctx *ctx = h->ctx;
    SX_ENTER(ctx);
    SX_REQUIRE(h->factored, "factor first");
    SX_REQUIRE(X && ldx >= h->n && nrhs >= 0, "bad right-hand side block");
    if (nrhs == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    const int64_t n = h->n, ld = h->ld;
    const double *A = h->a;
    const size_t need = static_cast<size_t>(n) * static_cast<size_t>(nrhs);
    if (need > h->tmp_cap) {
        SX_HIP(hipStreamSynchronize(s));
        (void)sx_dfree(h->tmp);
        h->tmp = nullptr;
        h->tmp_cap = 0;
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&h->tmp), sizeof(double) * need));
        h->tmp_cap = need;
    }
    const unsigned pg = static_cast<unsigned>(std::min<int64_t>((static_cast<int64_t>(need) + 255) / 256, 1 << 16));
    const unsigned tg = dl_trigrid(nrhs);
    if (!trans) {
        SX_HIP(hipMemcpy2DAsync(h->tmp, sizeof(double) * n, X, sizeof(double) * ldx, sizeof(double) * n, static_cast<size_t>(nrhs), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL((k_dl_permute<true>), dim3(pg), dim3(256), 0, s, n, nrhs, h->perm, h->tmp, n, X, ldx);
        for (int64_t j0 = 0; j0 < n; j0 += DL_NB) { // L forward
            const int nbk = static_cast<int>(std::min<int64_t>(DL_NB, n - j0));
            hipLaunchKernelGGL((k_dl_tri<true, false, true>), dim3(tg), dim3(256), 0, s, A + j0 + static_cast<size_t>(j0) * ld, ld, nbk, X + j0, ldx, nrhs);
            dl_gemm<false>(s, n - j0 - nbk, nrhs, nbk, A + j0 + nbk + static_cast<size_t>(j0) * ld, ld, X + j0, ldx, X + j0 + nbk, ldx);
        }
        for (int64_t j0 = (n - 1) / DL_NB * DL_NB; j0 >= 0; j0 -= DL_NB) { // U backward
            const int nbk = static_cast<int>(std::min<int64_t>(DL_NB, n - j0));
            hipLaunchKernelGGL((k_dl_tri<false, false, false>), dim3(tg), dim3(256), 0, s, A + j0 + static_cast<size_t>(j0) * ld, ld, nbk, X + j0, ldx, nrhs);
            dl_gemm<false>(s, j0, nrhs, nbk, A + static_cast<size_t>(j0) * ld, ld, X + j0, ldx, X, ldx);
        }
    } else {
        for (int64_t j0 = 0; j0 < n; j0 += DL_NB) { // U^T forward: z[after] -= U[blk, after]^T z[blk]
            const int nbk = static_cast<int>(std::min<int64_t>(DL_NB, n - j0));
            hipLaunchKernelGGL((k_dl_tri<true, true, false>), dim3(tg), dim3(256), 0, s, A + j0 + static_cast<size_t>(j0) * ld, ld, nbk, X + j0, ldx, nrhs);
            dl_gemm<true>(s, n - j0 - nbk, nrhs, nbk, A + j0 + static_cast<size_t>(j0 + nbk) * ld, ld, X + j0, ldx, X + j0 + nbk, ldx);
        }
        for (int64_t j0 = (n - 1) / DL_NB * DL_NB; j0 >= 0; j0 -= DL_NB) { // L^T backward: w[before] -= L[blk, before]^T w[blk]
            const int nbk = static_cast<int>(std::min<int64_t>(DL_NB, n - j0));
            hipLaunchKernelGGL((k_dl_tri<false, true, true>), dim3(tg), dim3(256), 0, s, A + j0 + static_cast<size_t>(j0) * ld, ld, nbk, X + j0, ldx, nrhs);
            dl_gemm<true>(s, j0, nrhs, nbk, A + j0, ld, X + j0, ldx, X, ldx);
        }
        SX_HIP(hipMemcpy2DAsync(h->tmp, sizeof(double) * n, X, sizeof(double) * ldx, sizeof(double) * n, static_cast<size_t>(nrhs), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL((k_dl_permute<false>), dim3(pg), dim3(256), 0, s, n, nrhs, h->perm, h->tmp, n, X, ldx);
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}
