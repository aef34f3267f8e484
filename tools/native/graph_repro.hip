// Minimal reproducer for profiles/r04/hipgraph_frames.md: a captured graph of N trivial kernels, instantiated once and launched
// R times; run it under `rocprofv3 --kernel-trace --stats -- ./graph_repro N R [K [S]]` (K = 0 / 256 / 1024 / 3072 bytes of kernel arguments by value, S = 1: a stream sync after every launch) to see where the tool faults.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/native/graph_repro tools/native/graph_repro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void k_inc(int *p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1;
}
template <int K>
struct Pad {
    int *p;
    char pad[K];
};
template <int K>
__global__ void k_big(Pad<K> a) { // K bytes of kernel arguments by value (the library's walks pass 150-250 bytes)
    if (threadIdx.x == 0 && blockIdx.x == 0) *a.p += 1 + (a.pad[K - 1] & 0);
}
template <int K>
void launch_big(hipStream_t s, int *d) {
    Pad<K> a{};
    a.p = d;
    hipLaunchKernelGGL(k_big<K>, dim3(64), dim3(256), 0, s, a);
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 4, kb = argc > 3 ? atoi(argv[3]) : 0, sync_each = argc > 4 ? atoi(argv[4]) : 0;
    int *d = nullptr;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipMalloc(&d, sizeof(int)));
    CK(hipMemsetAsync(d, 0, sizeof(int), s));
    CK(hipStreamSynchronize(s));
    hipGraph_t g;
    hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) {
        if (kb >= 3072) launch_big<3072>(s, d);
        else if (kb >= 1024) launch_big<1024>(s, d);
        else if (kb >= 256) launch_big<256>(s, d);
        else hipLaunchKernelGGL(k_inc, dim3(64), dim3(256), 0, s, d);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int r = 0; r < reps; ++r) {
        CK(hipGraphLaunch(ex, s));
        if (sync_each) CK(hipStreamSynchronize(s)); // (is it the launches outstanding, or all launches so far?)
    }
    CK(hipStreamSynchronize(s));
    int h = 0;
    CK(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
    printf("nodes %d, launches %d, %d bytes of arguments: counter %d (want %d)\n", n, reps, kb, h, n * reps);
    return h == n * reps ? 0 : 2;
}
