// K13: maximum-weight spanning tree of the bipartite supplier x demander graph of an OT problem, the
// first step of TNET's tree basis identification (reference: network_methods/tree_BI.py:32-59, which
// negates the flow weights and calls scipy.sparse.csgraph.minimum_spanning_tree; an arc with weight 0
// is not an edge there, nor here).
//
// Boruvka on the device.  Edges are totally ordered by (weight descending, linear arc index i*D + j
// ascending); every round each component takes its first outgoing edge in that order:
//   rows   per supplier node, its first outgoing edge: one wave scans the node's row of weights
//   cols   per demander node, likewise: one lane walks down the node's column (coalesced across lanes)
//   maxw   per component, atomicMax over its nodes' candidates (order-preserving bit pattern of the
//   mine   positive weight), then atomicMin of the arc index among the candidates with that weight --
//          S + D atomics per round instead of one per edge
//   hook   a component root links to the root at the other end of its edge and marks the edge; because
//          the order is strict and total the only cycles are pairs that chose the same edge, and there
//          the smaller root stays a root
//   flat   every node jumps to the root of its component
// The number of components at least halves per round, so ceil(log2(S + D)) rounds are enqueued without
// any host synchronisation; a round in which no root finds an outgoing edge raises a device flag that
// turns the kernels of all later rounds into immediate returns.  Max and min are order independent, so
// the tree does not depend on scheduling.  With distinct positive
// weights the maximum spanning tree is unique and equals the reference's; among equal weights the
// reference's choice follows an unstable argsort, here the smaller arc index wins.
#include "sx_internal.h"
#include "sx_segwalk.h"

namespace {

constexpr unsigned long long NO_EDGE = ~0ull;

// state[0] = number of roots that hooked in the current round, state[1] = 1 once a round hooked nothing
__global__ __launch_bounds__(SX_WG) void k_tree_init(int V, int *__restrict__ comp, int *__restrict__ state) {
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V) comp[v] = v;
    if (v == 0) state[0] = state[1] = 0;
}

__global__ __launch_bounds__(SX_WG) void k_tree_reset(int V, unsigned long long *__restrict__ best_key,
                                                      unsigned long long *__restrict__ best_edge,
                                                      int *__restrict__ state) {
    if (state[1]) return;
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V) {
        best_key[v] = 0ull;
        best_edge[v] = NO_EDGE;
    }
    if (v == 0) state[0] = 0; // read again only by this round's k_tree_flat, after k_tree_hook
}

// positive doubles compare like their bit patterns; w <= 0 and NaN are "no edge"
__device__ __forceinline__ bool edge_key(double w, unsigned long long &key) {
    if (!(w > 0.0)) return false;
    key = static_cast<unsigned long long>(__double_as_longlong(w));
    return true;
}

// better(a, b): candidate a = (key, edge) precedes b in the order (key descending, edge ascending)
__device__ __forceinline__ bool cand_better(unsigned long long ka, unsigned long long ea, unsigned long long kb,
                                            unsigned long long eb) {
    return ka > kb || (ka == kb && ea < eb);
}

// supplier node i: first edge (i, j) in the order whose other end lies in another component; one wave per row
__global__ __launch_bounds__(SX_WG) void k_tree_rows(int64_t S, int64_t D, const double *__restrict__ w,
                                                     const int *__restrict__ comp,
                                                     unsigned long long *__restrict__ node_key,
                                                     unsigned long long *__restrict__ node_edge,
                                                     const int *__restrict__ state) {
    if (state[1]) return;
    const int lane = threadIdx.x & 63;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * (SX_WG / 64) + (threadIdx.x >> 6);
    if (i >= S) return;
    const int ci = comp[i];
    unsigned long long bk = 0ull, be = NO_EDGE;
    for (int64_t j = lane; j < D; j += 64) {
        unsigned long long key;
        const int64_t e = i * D + j;
        if (edge_key(w[e], key) && comp[S + j] != ci && cand_better(key, e, bk, be)) {
            bk = key;
            be = static_cast<unsigned long long>(e);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long k2 = __shfl_down(bk, o, 64), e2 = __shfl_down(be, o, 64);
        if (cand_better(k2, e2, bk, be)) {
            bk = k2;
            be = e2;
        }
    }
    if (lane == 0) {
        node_key[i] = bk;
        node_edge[i] = be;
    }
}

// demander nodes: lane = column j, the rows are cut into gridDim.y * 4 interleaved slices (one per wave) so
// that no lane walks the whole column; slice candidates go to part_key / part_edge [slice][D]
__global__ __launch_bounds__(SX_WG) void k_tree_cols(int64_t S, int64_t D, const double *__restrict__ w,
                                                     const int *__restrict__ comp,
                                                     unsigned long long *__restrict__ part_key,
                                                     unsigned long long *__restrict__ part_edge,
                                                     const int *__restrict__ state) {
    if (state[1]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + lane;
    const int64_t slice = static_cast<int64_t>(blockIdx.y) * (SX_WG / 64) + wave;
    const int64_t nslices = static_cast<int64_t>(gridDim.y) * (SX_WG / 64);
    if (j >= D) return;
    const int cj = comp[S + j];
    unsigned long long bk = 0ull, be = NO_EDGE;
    for (int64_t i = slice; i < S; i += nslices) { // ascending rows: the first maximum of the slice wins
        unsigned long long key;
        if (edge_key(w[i * D + j], key) && comp[i] != cj && key > bk) {
            bk = key;
            be = static_cast<unsigned long long>(i * D + j);
        }
    }
    part_key[slice * D + j] = bk;
    part_edge[slice * D + j] = be;
}

__global__ __launch_bounds__(SX_WG) void k_tree_cols_reduce(int64_t S, int64_t D, int64_t nslices,
                                                            const unsigned long long *__restrict__ part_key,
                                                            const unsigned long long *__restrict__ part_edge,
                                                            unsigned long long *__restrict__ node_key,
                                                            unsigned long long *__restrict__ node_edge,
                                                            const int *__restrict__ state) {
    if (state[1]) return;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * SX_WG + threadIdx.x;
    if (j >= D) return;
    unsigned long long bk = 0ull, be = NO_EDGE;
    for (int64_t q = 0; q < nslices; ++q) {
        const unsigned long long k2 = part_key[q * D + j], e2 = part_edge[q * D + j];
        if (cand_better(k2, e2, bk, be)) {
            bk = k2;
            be = e2;
        }
    }
    node_key[S + j] = bk;
    node_edge[S + j] = be;
}

__global__ __launch_bounds__(SX_WG) void k_tree_maxw(int V, const int *__restrict__ comp,
                                                     const unsigned long long *__restrict__ node_key,
                                                     unsigned long long *__restrict__ best_key,
                                                     const int *__restrict__ state) {
    if (state[1]) return;
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V && node_key[v] != 0ull) atomicMax(&best_key[comp[v]], node_key[v]);
}

__global__ __launch_bounds__(SX_WG) void k_tree_mine(int V, const int *__restrict__ comp,
                                                     const unsigned long long *__restrict__ node_key,
                                                     const unsigned long long *__restrict__ node_edge,
                                                     const unsigned long long *__restrict__ best_key,
                                                     unsigned long long *__restrict__ best_edge,
                                                     const int *__restrict__ state) {
    if (state[1]) return;
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v < V && node_key[v] != 0ull && node_key[v] == best_key[comp[v]]) atomicMin(&best_edge[comp[v]], node_edge[v]);
}

__global__ __launch_bounds__(SX_WG) void k_tree_hook(int V, int64_t S, int64_t D, const int *__restrict__ comp,
                                                     const unsigned long long *__restrict__ best_edge,
                                                     int *__restrict__ link, uint8_t *__restrict__ in_tree,
                                                     int *__restrict__ state) {
    if (state[1]) return;
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v >= V) return;
    int to = v;
    if (comp[v] == v && best_edge[v] != NO_EDGE) { // a root with an outgoing edge
        const int64_t e = static_cast<int64_t>(best_edge[v]);
        const int64_t i = e / D;
        const int a = comp[i], b = comp[S + (e - i * D)];
        const int other = (a == v) ? b : a;
        in_tree[e] = 1;
        atomicAdd(&state[0], 1);
        const bool mutual = best_edge[other] == best_edge[v];
        to = (mutual && v < other) ? v : other;
    }
    link[v] = to;
}

__global__ __launch_bounds__(SX_WG) void k_tree_flat(int V, const int *__restrict__ link, int *__restrict__ comp,
                                                     int *__restrict__ state) {
    if (state[1]) return;
    const int v = blockIdx.x * SX_WG + threadIdx.x;
    if (v >= V) return;
    if (state[0] == 0) { // nothing hooked in this round: the forest is final (every thread stores the same value)
        state[1] = 1;
        return;
    }
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
    // row slices of the column pass: about 32 rows per lane, at most 64 slices (a multiple of 4)
    int64_t ysl = (S + 127) / 128;
    ysl = ysl < 1 ? 1 : (ysl > 16 ? 16 : ysl);
    const int64_t nslices = ysl * (SX_WG / 64);
    // workspace: best_key | best_edge | node_key | node_edge [V] u64 each | comp[V] i32 | link[V] i32 | state[2] i32
    //            | part_key | part_edge [nslices * D] u64 each
    const size_t fixed = (static_cast<size_t>(V) * 40 + 16 + 255) & ~static_cast<size_t>(255);
    SX_TRY(sx_reserve(ctx, fixed + 16 * static_cast<size_t>(nslices) * static_cast<size_t>(D)));
    unsigned long long *best_key = static_cast<unsigned long long *>(ctx->ws);
    unsigned long long *best_edge = best_key + V;
    unsigned long long *node_key = best_edge + V;
    unsigned long long *node_edge = node_key + V;
    int *comp = reinterpret_cast<int *>(node_edge + V);
    int *link = comp + V;
    int *state = link + V;
    unsigned long long *part_key = reinterpret_cast<unsigned long long *>(static_cast<char *>(ctx->ws) + fixed);
    unsigned long long *part_edge = part_key + nslices * D;
    hipStream_t s = ctx->stream;
    SX_HIP(hipMemsetAsync(in_tree, 0, static_cast<size_t>(n), s));
    const unsigned gv = static_cast<unsigned>((V + SX_WG - 1) / SX_WG);
    const unsigned gr = static_cast<unsigned>((S + SX_WG / 64 - 1) / (SX_WG / 64)); // one wave per supplier
    const dim3 gc(static_cast<unsigned>((D + 63) / 64), static_cast<unsigned>(ysl));  // 64 demanders x 4 row slices
    const unsigned gd = static_cast<unsigned>((D + SX_WG - 1) / SX_WG);
    hipLaunchKernelGGL(k_tree_init, dim3(gv), dim3(SX_WG), 0, s, V, comp, state);
    int rounds = 1;
    while ((1 << rounds) < V) ++rounds;
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(k_tree_reset, dim3(gv), dim3(SX_WG), 0, s, V, best_key, best_edge, state);
        hipLaunchKernelGGL(k_tree_rows, dim3(gr), dim3(SX_WG), 0, s, S, D, w, comp, node_key, node_edge, state);
        hipLaunchKernelGGL(k_tree_cols, gc, dim3(SX_WG), 0, s, S, D, w, comp, part_key, part_edge, state);
        hipLaunchKernelGGL(k_tree_cols_reduce, dim3(gd), dim3(SX_WG), 0, s, S, D, nslices, part_key, part_edge,
                           node_key, node_edge, state);
        hipLaunchKernelGGL(k_tree_maxw, dim3(gv), dim3(SX_WG), 0, s, V, comp, node_key, best_key, state);
        hipLaunchKernelGGL(k_tree_mine, dim3(gv), dim3(SX_WG), 0, s, V, comp, node_key, node_edge, best_key, best_edge,
                           state);
        hipLaunchKernelGGL(k_tree_hook, dim3(gv), dim3(SX_WG), 0, s, V, S, D, comp, best_edge, link, in_tree, state);
        hipLaunchKernelGGL(k_tree_flat, dim3(gv), dim3(SX_WG), 0, s, V, link, comp, state);
    }
    SX_HIP(hipGetLastError());
    return SX_OK;
}
