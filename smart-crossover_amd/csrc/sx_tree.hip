// K13: maximum-weight spanning tree of the bipartite supplier x demander graph of an OT problem, the
// first step of TNET's tree basis identification (reference: network_methods/tree_BI.py:32-59, which
// negates the flow weights and calls scipy.sparse.csgraph.minimum_spanning_tree; an arc with weight 0
// is not an edge there, nor here).
//
// Boruvka on the device.  Edges are totally ordered by (weight descending, linear arc index i*D + j
// ascending); every round each component takes its first outgoing edge in that order:
//   maxw   per component, atomicMax over the order-preserving bit pattern of the positive weights
//   mine   per component, atomicMin of the arc index among the edges that carry that weight
//   hook   a component root links to the root at the other end of its edge and marks the edge; because
//          the order is strict and total the only cycles are pairs that chose the same edge, and there
//          the smaller root stays a root
//   flat   every node jumps to the root of its component
// The number of components at least halves per round, so ceil(log2(S + D)) rounds are enqueued without
// any host synchronisation; rounds after the last merge find no outgoing edge and change nothing.  Max
// and min are order independent, so the tree does not depend on scheduling.  With distinct positive
// weights the maximum spanning tree is unique and equals the reference's; among equal weights the
// reference's choice follows an unstable argsort, here the smaller arc index wins.
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

constexpr unsigned long long NO_EDGE = ~0ull;

__global__ __launch_bounds__(SX_WG) void k_tree_init(int V, int *__restrict__ comp) {
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V) comp[v] = v;
}

__global__ __launch_bounds__(SX_WG) void k_tree_reset(int V, unsigned long long *__restrict__ best_key,
                                                      unsigned long long *__restrict__ best_edge) {
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V) {
        best_key[v] = 0ull;
        best_edge[v] = NO_EDGE;
    }
}

// positive doubles compare like their bit patterns; w <= 0 and NaN are "no edge"
__device__ __forceinline__ bool edge_key(double w, unsigned long long &key) {
    if (!(w > 0.0)) return false;
    key = static_cast<unsigned long long>(__double_as_longlong(w));
    return true;
}

__global__ __launch_bounds__(SX_WG) void k_tree_maxw(int64_t S, int64_t D, const double *__restrict__ w,
                                                     const int *__restrict__ comp,
                                                     unsigned long long *__restrict__ best_key) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        unsigned long long key;
        if (!edge_key(w[e], key)) continue;
        const int64_t i = e / D;
        const int ci = comp[i], cj = comp[S + (e - i * D)];
        if (ci == cj) continue;
        // the plain reads only skip atomics that could not raise the maximum (it never decreases)
        if (key > best_key[ci]) atomicMax(&best_key[ci], key);
        if (key > best_key[cj]) atomicMax(&best_key[cj], key);
    }
}

__global__ __launch_bounds__(SX_WG) void k_tree_mine(int64_t S, int64_t D, const double *__restrict__ w,
                                                     const int *__restrict__ comp,
                                                     const unsigned long long *__restrict__ best_key,
                                                     unsigned long long *__restrict__ best_edge) {
    const int64_t n = S * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x; e < n;
         e += static_cast<int64_t>(gridDim.x) * SX_WG) {
        unsigned long long key;
        if (!edge_key(w[e], key)) continue;
        const int64_t i = e / D;
        const int ci = comp[i], cj = comp[S + (e - i * D)];
        if (ci == cj) continue;
        const unsigned long long ue = static_cast<unsigned long long>(e);
        if (key == best_key[ci] && ue < best_edge[ci]) atomicMin(&best_edge[ci], ue);
        if (key == best_key[cj] && ue < best_edge[cj]) atomicMin(&best_edge[cj], ue);
    }
}

__global__ __launch_bounds__(SX_WG) void k_tree_hook(int V, int64_t S, int64_t D, const int *__restrict__ comp,
                                                     const unsigned long long *__restrict__ best_edge,
                                                     int *__restrict__ link, uint8_t *__restrict__ in_tree) {
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v >= V) return;
    int to = v;
    if (comp[v] == v && best_edge[v] != NO_EDGE) { // a root with an outgoing edge
        const int64_t e = static_cast<int64_t>(best_edge[v]);
        const int64_t i = e / D;
        const int a = comp[i], b = comp[S + (e - i * D)];
        const int other = (a == v) ? b : a;
        in_tree[e] = 1;
        const bool mutual = best_edge[other] == best_edge[v];
        to = (mutual && v < other) ? v : other;
    }
    link[v] = to;
}

__global__ __launch_bounds__(SX_WG) void k_tree_flat(int V, const int *__restrict__ link, int *__restrict__ comp) {
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v >= V) return;
    int r = comp[v]; // a root of the previous round; link[] is a forest over those roots
    for (int guard = 0; guard < V; ++guard) {
        const int up = link[r];
        if (up == r) break;
        r = up;
    }
    comp[v] = r;
}

} // namespace

SX_API int sx_spanning_tree_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *w, uint8_t *in_tree) {
    SX_ENTER(ctx);
    SX_REQUIRE(S >= 0 && D >= 0, "negative size");
    SX_REQUIRE(S + D < (int64_t(1) << 30), "too many nodes");
    if (S == 0 || D == 0) return SX_OK;
    SX_REQUIRE(w && in_tree, "NULL argument");
    const int V = static_cast<int>(S + D);
    const int64_t n = S * D;
    // workspace: best_key[V] u64 | best_edge[V] u64 | comp[V] i32 | link[V] i32
    SX_TRY(sx_reserve(ctx, static_cast<size_t>(V) * 24));
    unsigned long long *best_key = static_cast<unsigned long long *>(ctx->ws);
    unsigned long long *best_edge = best_key + V;
    int *comp = reinterpret_cast<int *>(best_edge + V);
    int *link = comp + V;
    hipStream_t s = ctx->stream;
    SX_HIP(hipMemsetAsync(in_tree, 0, static_cast<size_t>(n), s));
    const unsigned gv = static_cast<unsigned>((V + SX_WG - 1) / SX_WG);
    int64_t ge64 = (n + SX_WG - 1) / SX_WG;
    const unsigned ge = static_cast<unsigned>(ge64 < 4096 ? ge64 : 4096);
    hipLaunchKernelGGL(k_tree_init, dim3(gv), dim3(SX_WG), 0, s, V, comp);
    int rounds = 1;
    while ((1 << rounds) < V) ++rounds;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k_tree_reset, dim3(gv), dim3(SX_WG), 0, s, V, best_key, best_edge);
        hipLaunchKernelGGL(k_tree_maxw, dim3(ge), dim3(SX_WG), 0, s, S, D, w, comp, best_key);
        hipLaunchKernelGGL(k_tree_mine, dim3(ge), dim3(SX_WG), 0, s, S, D, w, comp, best_key, best_edge);
        hipLaunchKernelGGL(k_tree_hook, dim3(gv), dim3(SX_WG), 0, s, V, S, D, comp, best_edge, link, in_tree);
        hipLaunchKernelGGL(k_tree_flat, dim3(gv), dim3(SX_WG), 0, s, V, link, comp);
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}
