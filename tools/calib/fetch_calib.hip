// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the walk kernels use
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Each kernel reads N bytes of a 1 GiB buffer exactly once:
//   k_read16  16 B per lane (the entry stream: 4 x int32 / 2 x double per load)
//   k_read8    8 B per lane, consecutive lanes (per-segment vectors: c, x, l, u, rowptr ...)
//   k_window   256 lanes fill a 32 KiB block with 16 loads of 8 B per lane (sx_window_fill's pattern)
//   k_read2    2 B per lane (the uint16 row starts of the blocked layout)
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k_read16(const int4 *p, size_t n, int *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    int acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { int4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) *out = acc;
}
__global__ void k_read8(const double *p, size_t n, int *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0.12345) *out = 1;
}
__global__ void k_window(const double *p, size_t nblocks, int *out) {
    __shared__ double win[4096];
    double acc = 0;
    for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const double *src = p + b * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) win[r * 256 + threadIdx.x] = src[r * 256 + threadIdx.x];
        __syncthreads();
        acc += win[(threadIdx.x * 17) & 4095];
        __syncthreads();
    }
    if (acc == 0.12345) *out = 1;
}
__global__ void k_read2(const uint16_t *p, size_t n, int *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    int acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == -1) *out = acc;
}

int main() {
    const size_t bytes = size_t(1) << 30;
    void *buf; int *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const int4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, (const double *)buf, bytes / 8, out);
        hipLaunchKernelGGL(k_window, dim3(1024), dim3(256), 0, 0, (const double *)buf, bytes / 32768, out);
        hipLaunchKernelGGL(k_read2, dim3(4096), dim3(256), 0, 0, (const uint16_t *)buf, bytes / 2, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("each kernel read %zu bytes\n", bytes);
    return 0;
}
