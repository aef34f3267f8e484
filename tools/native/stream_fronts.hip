// How fast does the chip read a 1.44 GB array when each of G resident workgroups streams a PRIVATE contiguous range (K2's super-tiles:
// 768 read fronts advancing 24 KB at a time) against the same bytes laid out so that step k of all workgroups is one contiguous region?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/native/stream_fronts tools/native/stream_fronts.hip ; run: ./stream_fronts
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));

// tile t reads `steps` chunks of `chunk_bytes`; chunk (t, k) lives at base + off(t, k)
template <int MODE> // 0: private ranges (tile-major), 1: step-major inside groups of G tiles
__global__ __launch_bounds__(512, 6) void k_read(const char *__restrict__ base, int ntiles, int steps, int chunk_bytes, int G, int *sink) {
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    int acc = 0;
    for (int k = 0; k < steps; ++k) {
        size_t off;
        if (MODE == 0) off = (static_cast<size_t>(t) * steps + k) * chunk_bytes;
        else {
            const int g = t / G, j = t % G, ng = (ntiles - g * G < G) ? ntiles - g * G : G;
            off = (static_cast<size_t>(g) * G * steps + static_cast<size_t>(k) * ng + j) * chunk_bytes;
        }
        const v4i *p = reinterpret_cast<const v4i *>(base + off);
        for (int i = threadIdx.x; i < chunk_bytes / 16; i += 512) {
            const v4i v = __builtin_nontemporal_load(p + i);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        __syncthreads(); // (a step of the walk ends with a barrier)
    }
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    const int ntiles = 2096, steps = 29, chunk = 24576, G = 768;
    const size_t bytes = static_cast<size_t>(ntiles) * steps * chunk;
    char *d;
    int *sink;
    hipMalloc(&d, bytes);
    hipMalloc(&sink, 4);
    hipMemset(d, 1, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(ntiles), dim3(512), 0, 0, d, ntiles, steps, chunk, G, sink);
            else hipLaunchKernelGGL(k_read<1>, dim3(ntiles), dim3(512), 0, 0, d, ntiles, steps, chunk, G, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("%s: %.3f ms, %.2f TB/s (%.2f GB, %d workgroups of 512 lanes, %d steps of %d KB)\n", mode ? "step-major in groups of 768" : "private contiguous ranges ", best,
               bytes / best / 1e9, bytes / 1e9, ntiles, steps, chunk / 1024);
    }
    return 0;
}
