// Probe for profiles/r03/experiments/k16s_fold.md: read-modify-write of an m x n tableau of doubles the way k_tb_fold walks
// it -- a workgroup owns 256 rows and visits the columns one after the other -- in (a) the column-major layout of the
// crossover (consecutive columns 8 m bytes apart) and (b) a row-blocked layout (the 256 rows of a block contiguous for
// all columns).  Same bytes, same arithmetic; only the address pattern differs.
// build: hipcc --offload-arch=gfx950 -O3 -o stride_probe tools/probes/stride_probe.hip ; run: ./stride_probe [m] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_colmajor(long m, long n, double *__restrict__ T, long tile) {
    // grid (row blocks strided, column tiles): as the kept fold kernel
    const long j0 = (long)blockIdx.y * tile;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < m; p += (long)gridDim.x * 256)
        for (long j = j0; j < j0 + tile && j < n; ++j) {
            double *t = T + j * m + p;
            *t = *t * 1.0000001 + 1.0;
        }
}
__global__ __launch_bounds__(256) void k_blocked(long m, long n, double *__restrict__ T, long tile) {
    const long j0 = (long)blockIdx.y * tile;
    for (long b = blockIdx.x; b * 256 < m; b += gridDim.x)
        for (long j = j0; j < j0 + tile && j < n; ++j) {
            double *t = T + (b * n + j) * 256 + threadIdx.x;
            *t = *t * 1.0000001 + 1.0;
        }
}
int main(int argc, char **argv) {
    const long m = argc > 1 ? atol(argv[1]) : 1000000, n = argc > 2 ? atol(argv[2]) : 8192;
    const long mp = (m + 255) / 256 * 256;
    double *T;
    CK(hipMalloc(&T, sizeof(double) * mp * n));
    CK(hipMemset(T, 0, sizeof(double) * mp * n));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long tile = 64;
    const dim3 grid(128, (unsigned)((n + tile - 1) / tile));
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_colmajor, grid, dim3(256), 0, 0, mp, n, T, tile);
            else hipLaunchKernelGGL(k_blocked, grid, dim3(256), 0, 0, mp, n, T, tile);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s m=%ld n=%ld: %.2f ms, %.2f TB/s (read + write)\n", which ? "row-blocked " : "column-major", m, n, ms,
                   2.0 * 8.0 * mp * n / ms * 1e-9);
        }
    return 0;
}
