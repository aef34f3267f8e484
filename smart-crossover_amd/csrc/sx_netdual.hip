// K16d: DUAL network simplex on a spanning-tree basis, the whole GPU on one pivot -- the re-solves of the
// network crossover's column generation (network_methods/net_manager.py:211-222 solve_subproblem ->
// solve_mcf(..., warm_start_basis); network_methods/algorithms.py:109-140).
//
// Why dual: a round of the column generation adds thousands of arcs, all sitting at a bound, to a problem whose
// tree was optimal.  The tree stays dual feasible once every new arc with a wrong-signed reduced cost is moved to
// its other bound (capacities are finite), so the dual method starts at once, and its bound-flipping ratio test
// moves many arcs per iteration; the primal method (sx_netsimplex.hip) has to bring them in one by one and takes
// 3-6x the iterations on those rounds (profiles/r02/network_simplex.md).  oracle/net_simplex.py states the
// algorithm; on integral data device and oracle make the same pivots.
//
// Why the whole GPU: with the tree in PREORDER (pos, size, order) every step of an iteration is a flat pass --
//   leaving    tree arc with the largest violation^2 / |subtree| (|subtree| = squared norm of the arc's row of
//              the basis inverse: exact dual steepest edge for free), all nodes, two-level reduction
//   cut        arcs with exactly one end in the subtree S = [pos, pos + size): the adjacency (CSR rows) of S's
//              nodes when S is small, all arcs otherwise; candidates -> list
//   ratio      ascending |reduced cost|: arcs are passed (flipped) while the flips leave the leaving arc
//              infeasible, the next one enters (one workgroup, a few reductions)
//   update     potentials of S, flows of the tree arcs whose subtree separates the ends of a moved arc, subtree
//              sizes above the two attachment points: all nodes, no climbing
//   re-hang    S re-rooted at the entering arc: its path to the old root in order (ordered compaction by
//              position), then every element of the affected range of order[] is placed independently
// -- and at config 4 (131,073 nodes, 1e6 arcs) one workgroup needs 340 us for such passes.  Here a cooperative
// grid of workgroups shares them, six grid barriers per iteration.  Deterministic: no floating-point atomics,
// fixed reduction orders, ties by index.
#include "sx_internal.h"

#include <hip/hip_cooperative_groups.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace cg = cooperative_groups;

namespace {

constexpr int ND_T = 1024;       // lanes per workgroup
constexpr int ND_W = ND_T / 64;  // waves per workgroup
constexpr int ND_GMAX = 256;     // workgroups of the cooperative grid (one per CU at most)
constexpr int ND_PATH_LDS = 1024; // path entries cached in LDS for the new positions
constexpr int ND_SMALL = 512;     // subtrees up to this many nodes: workgroup 0 scans the cut alone
constexpr int ND_LCAP = 2048;     // candidates of the ratio test kept in LDS
#ifndef SX_ND_FINE_TICKS
#define SX_ND_FINE_TICKS 0 // 1: SX_NS_PROFILE also splits the pass into loads / compute (costs a drain per round)
#endif
constexpr int ND_PUSH_LDS = 64;   // moved arcs of a decision cached in LDS for the pass
constexpr int ST_TREE = 0, ST_LOWER = 1, ST_UPPER = -1;
constexpr int UNK = -2;

struct NdDec { // an iteration's decision: written by workgroup 0, applied by all in the next pass
    int has;        // 0 before the first iteration
    int enter;      // entering arc, -1: none (primal infeasible)
    int v_in;       // its end outside the subtree S
    int b_pos;      // pos[v_in]
    int a_pos, n_sub; // S = old positions [a_pos, a_pos + n_sub)
    int K;          // nodes on the path u_in .. v (snode / sarc / spos / ssize, u_in first)
    int npush;      // moved arcs: the passed ones and, last, the entering arc (push_pt / push_ph / push_d)
    int to_lower;   // the leaving arc lands on 0 (else on its capacity)
    int lo, hi, newstart; // preorder positions [lo, hi) change; S starts at newstart
    double dy;      // shift of the potentials of S
    double enter_flow, enter_cap; // the entering arc: it hangs u_in from now on
};

struct NdShared {
    long long iters;
    long long status; // 0 optimal, 1 primal infeasible, 3 iteration limit, 5 not applicable
    long long flips;
    double obj;
    double max_violation;
    int not_network;
    int ntree, nroot, root;
    int any_capacity; // a non-tree arc with a finite capacity exists
    int any_nontree;
    int cand_count;
    int pad_;
    NdDec dec;
    // SX_NS_PROFILE: 10 ns ticks per phase (workgroup 0) and sums over the iterations
    long long t_phase[8];
    long long sum_cand, sum_sub, sum_path, sum_range, n_small;
};

struct NdProblem {
    int V;
    long long E;
    int G; // workgroups
    const int64_t *rowptr; // CSR of A: row v -> its arcs
    const int32_t *rowarc;
    int32_t *rowother;     // per CSR entry: the arc's other end o, or ~o when this row is the arc's head
    int2 *noderec;         // per node: {first CSR entry, degree}
    int4 *ordq;            // by preorder position: {node, first CSR entry, degree, -}: order[] of the iterations
    int32_t *tail, *head;
    const double *cost, *cap;
    double *flow;          // [E] non-tree arcs' values (tree arcs: filled in at the end from nflow)
    double *nflow, *ncap;  // [V] flow and capacity of the arc that hangs a node (coalesced in the pass)
    double *sflow, *scap;  // [V] those of the path nodes before the re-hang
    int8_t *state;
    int4 *nd;       // {parent, pred arc, pos, size}
    double *y;
    int32_t *order, *tmp;
    int32_t *first_child, *next_sib; // set-up
    double *e, *bsum;                // set-up: b - A x_N in preorder, block sums
    // per iteration
    double *part_s;
    int *part_n;
    int *part_cnt;
    int32_t *cand_j;
    double *cand_r, *cand_c;
    int32_t *push_pt, *push_ph;
    double *push_d;
    int32_t *snode, *sarc, *spos, *ssize;
    int32_t *pathidx; // [V] 1 + index on the path of the pending decision, 0 elsewhere
    double *acc[2];
    int32_t *anc[2];
    NdShared *sh;
    unsigned long long *bar; // grid barrier: arrivals
    int nap_short, nap_long; // naps between polls: after a pass / while workgroup 0 decides
    int lazy;                // 1: barriers without cache invalidation where the next phase allows it (nd_barrier)
};

// ------------------------------------------------------------------ arcs from the columns of A
__global__ __launch_bounds__(256) void k_nd_endpoints(int64_t E, const int64_t *__restrict__ colptr,
                                                      const int32_t *__restrict__ rowidx,
                                                      const double *__restrict__ val, const double *__restrict__ l,
                                                      const double *__restrict__ u, const int8_t *__restrict__ vbasis,
                                                      int32_t *__restrict__ tail, int32_t *__restrict__ head,
                                                      int8_t *__restrict__ state, NdShared *sh) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= E) return;
    const int64_t p0 = colptr[j];
    bool ok = colptr[j + 1] - p0 == 2 && l[j] == 0.0 && u[j] >= 0.0;
    int t = 0, h = 0;
    if (ok) {
        const double a = val[p0], b = val[p0 + 1];
        if (a == 1.0 && b == -1.0) {
            t = rowidx[p0];
            h = rowidx[p0 + 1];
        } else if (a == -1.0 && b == 1.0) {
            t = rowidx[p0 + 1];
            h = rowidx[p0];
        } else {
            ok = false;
        }
    }
    const int code = vbasis[j];
    const int st = code == 0 ? ST_TREE : code == -2 ? ST_UPPER : ST_LOWER;
    if (code != 0 && code != -1 && code != -2) ok = false;
    if (st == ST_UPPER && isinf(u[j])) ok = false;
    tail[j] = t;
    head[j] = h;
    state[j] = static_cast<int8_t>(st);
    if (!ok) sh->not_network = 1;
    if (st == ST_TREE) atomicAdd(&sh->ntree, 1);
    else {
        sh->any_nontree = 1; // benign races: every writer stores 1
        if (!isinf(u[j])) sh->any_capacity = 1;
    }
}

__global__ __launch_bounds__(256) void k_nd_root(int64_t V, const int8_t *__restrict__ cbasis, NdShared *sh) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= V) return;
    if (cbasis[i] == 0) {
        atomicAdd(&sh->nroot, 1);
        sh->root = static_cast<int>(i);
    }
}

// ------------------------------------------------------------------ set-up: tree arrays, preorder, potentials
// one workgroup: parents by hooking sweeps from the root, children in ascending node order, one depth-first pass
__global__ __launch_bounds__(ND_T) void k_nd_tree(NdProblem P) {
    __shared__ int s_flag, s_count;
    const int tid = threadIdx.x;
    const int V = P.V;
    const long long E = P.E;
    NdShared *sh = P.sh;
    int4 *nd = P.nd;
    volatile int4 *ndv = nd;
    const int root = sh->root;
    for (int v = tid; v < V; v += ND_T) {
        nd[v] = make_int4(v == root ? -1 : UNK, -1, 0, 1);
        P.first_child[v] = -1;
        P.next_sib[v] = -1;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (long long e = tid; e < E; e += ND_T)
        if (P.state[e] == ST_TREE) {
            const int slot = __hip_atomic_fetch_add(&s_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < V) P.tmp[slot] = static_cast<int32_t>(e);
        }
    __syncthreads();
    const int ntree = s_count;
    for (int sweep = 0; sweep <= V; ++sweep) {
        __syncthreads();
        if (tid == 0) s_flag = 0;
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < ntree; i += ND_T) {
            const int e = P.tmp[i];
            const int p = P.tail[e], q = P.head[e];
            const int pp = ndv[p].x, pq = ndv[q].x;
            if (pp != UNK && pq == UNK) {
                ndv[q].x = p;
                ndv[q].y = e;
                changed = 1;
            } else if (pq != UNK && pp == UNK) {
                ndv[p].x = q;
                ndv[p].y = e;
                changed = 1;
            }
        }
        if (changed) s_flag = 1;
        __syncthreads();
        if (!s_flag) break;
    }
    {
        int bad = 0;
        for (int v = tid; v < V; v += ND_T)
            if (nd[v].x == UNK) bad = 1;
        if (tid == 0) s_flag = 0;
        __syncthreads();
        if (bad) s_flag = 1;
        __syncthreads();
        if (s_flag || ntree != V - 1) {
            if (tid == 0) sh->status = 5;
            return;
        }
    }
    if (tid == 0) {
        for (int v = V - 1; v >= 0; --v) {
            if (v == root) continue;
            const int p = nd[v].x;
            P.next_sib[v] = P.first_child[p];
            P.first_child[p] = v;
        }
        int t = 1, v = root;
        P.y[root] = 0.0;
        nd[root].z = 0;
        P.order[0] = root;
        bool down = true;
        for (long long steps = 0; steps < 4ll * V + 8; ++steps) {
            if (down) {
                const int c = P.first_child[v];
                if (c >= 0) {
                    const int a = nd[c].y;
                    P.y[c] = (P.tail[a] == c) ? P.cost[a] + P.y[v] : P.y[v] - P.cost[a];
                    nd[c].z = t;
                    P.order[t] = c;
                    ++t;
                    v = c;
                    continue;
                }
                down = false;
            }
            nd[v].w = t - nd[v].z;
            if (v == root) break;
            const int par = nd[v].x;
            const int sb = P.next_sib[v];
            if (sb >= 0) {
                const int a2 = nd[sb].y;
                P.y[sb] = (P.tail[a2] == sb) ? P.cost[a2] + P.y[par] : P.y[par] - P.cost[a2];
                nd[sb].z = t;
                P.order[t] = sb;
                ++t;
                v = sb;
                down = true;
            } else {
                v = par;
            }
        }
    }
}

// adjacency side arrays: the other end of every CSR entry (one wave per row), the rows' extents
__global__ __launch_bounds__(256) void k_nd_adjacency(NdProblem P) {
    const int lane = threadIdx.x & 63;
    const long long w = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) >> 6;
    if (w >= P.V) return;
    const int64_t p0 = P.rowptr[w], p1 = P.rowptr[w + 1];
    if (lane == 0) P.noderec[w] = make_int2(static_cast<int>(p0), static_cast<int>(p1 - p0));
    for (int64_t p = p0 + lane; p < p1; p += 64) {
        const int j = P.rowarc[p];
        const int t = P.tail[j], h = P.head[j];
        P.rowother[p] = t == w ? h : ~t;
    }
}

__global__ __launch_bounds__(256) void k_nd_ordrec(NdProblem P) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.V) {
        const int w = P.order[t];
        const int2 rec = P.noderec[w];
        P.ordq[t] = make_int4(w, rec.x, rec.y, 0);
    }
}

// the tree arrays kept from the previous solve describe THIS basis iff every node's arc is coded basic and joins it to
// its parent (V - 1 distinct tree arcs out of the V - 1 the basis has)
// The kept potentials belong to the costs of the solve that left them: a tree arc whose reduced cost under
// THIS call's costs is not zero (to rounding: the potentials were shifted subtree by subtree) means the costs
// changed -- re-scaled, perturbed, another instance on the same graph -- and the tree set-up has to run again.
__global__ __launch_bounds__(256) void k_nd_check_kept(NdProblem P, const double *__restrict__ y_kept, int *__restrict__ differs) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= P.V) return;
    const int4 r = P.nd[w];
    bool ok;
    if (r.x < 0) {
        ok = w == P.sh->root;
    } else {
        ok = r.y >= 0 && r.y < P.E && r.x < P.V && P.state[r.y] == ST_TREE;
        if (ok) {
            const int t = P.tail[r.y], h = P.head[r.y];
            ok = (t == w && h == r.x) || (h == w && t == r.x);
            if (ok) {
                const double c = P.cost[r.y], yt = y_kept[t], yh = y_kept[h];
                const double rc = (c - yt) + yh;
                ok = fabs(rc) <= 1e-10 * (1.0 + fabs(c) + fabs(yt) + fabs(yh)); // (NaN fails the test)
            }
        }
    }
    if (!ok) *differs = 1;
}

__global__ __launch_bounds__(256) void k_nd_keep_order(NdProblem P) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < P.V) P.order[t] = P.ordq[t].x;
}

// dual feasibility by moving arcs to their other bound; x_N for the right-hand side
__global__ __launch_bounds__(256) void k_nd_flip(NdProblem P, double *__restrict__ xn) {
    const long long j = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= P.E) return;
    int st = P.state[j];
    double x = 0.0;
    if (st != ST_TREE) {
        const double rc = (P.cost[j] - P.y[P.tail[j]]) + P.y[P.head[j]];
        const bool wrong = st == ST_LOWER ? rc < 0.0 : rc > 0.0;
        if (wrong) {
            if (isinf(P.cap[j])) {
                P.sh->not_network = 1; // cannot be made dual feasible: the primal method's job
            } else {
                st = -st;
                P.state[j] = static_cast<int8_t>(st);
                atomicAdd(reinterpret_cast<unsigned long long *>(&P.sh->flips), 1ull);
            }
        }
        if (st == ST_UPPER) x = P.cap[j];
    }
    xn[j] = x;
    P.flow[j] = x;
}

// e[t] = (b - A x_N)[order[t]] and the sums of its blocks of 256
__global__ __launch_bounds__(256) void k_nd_excess(NdProblem P, const double *__restrict__ beff) {
    __shared__ double s[256];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const double v = t < P.V ? beff[P.order[t]] : 0.0;
    if (t < P.V) P.e[t] = v;
    s[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) { // left to right: a fixed order
        double a = 0.0;
        for (int k = 0; k < 256; ++k) a = a + s[k];
        P.bsum[blockIdx.x] = a;
    }
}

// tree flows: the arc of node w carries +- the sum of e over w's preorder interval (left to right, whole
// blocks through their sums)
__global__ __launch_bounds__(256) void k_nd_initflows(NdProblem P) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= P.V) return;
    const int4 r = P.nd[w];
    if (r.x < 0) {
        P.nflow[w] = 0.0;
        P.ncap[w] = INFINITY; // the root hangs on nothing
        return;
    }
    int t = r.z;
    const int end = r.z + r.w;
    double s = 0.0;
    while (t < end && (t & 255)) s = s + P.e[t++];
    while (t + 256 <= end) {
        s = s + P.bsum[t >> 8];
        t += 256;
    }
    while (t < end) s = s + P.e[t++];
    P.nflow[w] = (P.tail[r.y] == w) ? s : -s;
    P.ncap[w] = P.cap[r.y];
}

// ------------------------------------------------------------------ grid barrier
// Arrival counter that only grows (epoch e is complete at (e + 1) * G arrivals); the waiting workgroups poll it
// with naps in between: while workgroup 0 decides alone, 128 workgroups polling the same line without a pause
// slow its dependent loads by a factor of two (profiles/r02/netdual.md).  Release / acquire at agent scope make
// the passes' plain stores visible across the XCDs' L2s, as the cooperative-groups barrier does.
// `acquire` = false: the workgroup goes on WITHOUT invalidating its caches.  An acquire at agent scope empties the
// CU's L1 and the XCD's L2 of everything that is not dirty, and then every load of the next phase misses; a
// workgroup whose next phase reads only (a) data it wrote itself -- the pass: every node's records belong to one
// lane for the whole solve -- and (b) a handful of values written elsewhere, which it fetches with nd_ld() (loads
// at agent scope, served from memory), does not need that.  The release (arrivals) is always there.
__device__ __forceinline__ void nd_barrier(unsigned long long *counter, unsigned long long &epoch, int G, int naps,
                                           bool acquire = true) {
    __syncthreads();
    if (G == 1) return; // one workgroup: its own barrier orders everything, and the caches stay warm
    ++epoch;
    if (threadIdx.x == 0) {
        const unsigned long long target = epoch * static_cast<unsigned long long>(G);
        __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (acquire) {
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
                for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(16); // 16 x 64 clocks
        } else {
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
                for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
}

// a value another workgroup wrote before the last barrier, read without an acquire of this workgroup's own
template <class T>
__device__ __forceinline__ T nd_ld(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int4 nd_ld4(const int4 *p) {
    const long long *q = reinterpret_cast<const long long *>(p);
    const long long a = nd_ld(q), b = nd_ld(q + 1);
    return make_int4(static_cast<int>(a), static_cast<int>(a >> 32), static_cast<int>(b), static_cast<int>(b >> 32));
}

// ------------------------------------------------------------------ workgroup reductions
struct NdLds {
    double d[ND_W];
    double d2[ND_W];
    int i[ND_W];
    int scan[ND_W];
    int spos[ND_PATH_LDS], ssize[ND_PATH_LDS];
    int s_node[ND_SMALL], s_p0[ND_SMALL], s_off[ND_SMALL]; // workgroup 0: the nodes of a small S and their rows
    double c_r[ND_LCAP], c_c[ND_LCAP];                     // workgroup 0: candidates of the ratio test
    int c_j[ND_LCAP];
    int ccount;
    int p_pt[ND_PUSH_LDS], p_ph[ND_PUSH_LDS]; // the decision's moved arcs
    double p_d[ND_PUSH_LDS];
    // result of the ratio test (wave 0 -> workgroup)
    int res_enter, res_npush;
    double res_theta, res_cap, res_remaining;
};

// larger score wins, then the smaller index; result in every lane
__device__ __forceinline__ void nd_argmax(NdLds &L, double &s, int &n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double s2 = __shfl_xor(s, o, 64);
        const int n2 = __shfl_xor(n, o, 64);
        if (s2 > s || (s2 == s && n2 < n)) {
            s = s2;
            n = n2;
        }
    }
    __syncthreads();
    if (lane == 0) {
        L.d[wave] = s;
        L.i[wave] = n;
    }
    __syncthreads();
    s = L.d[0];
    n = L.i[0];
#pragma unroll
    for (int w = 1; w < ND_W; ++w) {
        const double s2 = L.d[w];
        const int n2 = L.i[w];
        if (s2 > s || (s2 == s && n2 < n)) {
            s = s2;
            n = n2;
        }
    }
}

// smaller (r, j) wins; c travels with it; j < 0 = nothing
__device__ __forceinline__ void nd_argmin(NdLds &L, double &r, int &j, double &c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto better = [](double r2, int j2, double r1, int j1) { return j2 >= 0 && (j1 < 0 || r2 < r1 || (r2 == r1 && j2 < j1)); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double r2 = __shfl_xor(r, o, 64), c2 = __shfl_xor(c, o, 64);
        const int j2 = __shfl_xor(j, o, 64);
        if (better(r2, j2, r, j)) {
            r = r2;
            j = j2;
            c = c2;
        }
    }
    __syncthreads();
    if (lane == 0) {
        L.d[wave] = r;
        L.d2[wave] = c;
        L.i[wave] = j;
    }
    __syncthreads();
    r = L.d[0];
    c = L.d2[0];
    j = L.i[0];
#pragma unroll
    for (int w = 1; w < ND_W; ++w)
        if (better(L.d[w], L.i[w], r, j)) {
            r = L.d[w];
            c = L.d2[w];
            j = L.i[w];
        }
}

__device__ __forceinline__ int nd_sum(NdLds &L, int v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) L.i[wave] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < ND_W; ++w) t += L.i[w];
    return t;
}

// ------------------------------------------------------------------ the solver
// Two grid barriers per iteration:
//   pass   every node for itself, all workgroups: the previous iteration's decision is applied IN PLACE -- flow of
//          the node's tree arc (the subtree separates the ends of a moved arc), potential (inside S), subtree size
//          (ancestors of the two attachment points), the re-rooted path nodes' parent / arc / size, the node's
//          new preorder position and its slot of order[] -- nobody reads another node's data in this pass, so no
//          temporary copy and no barrier in between; then the node's violation^2 / size, two-level argmax
//   ----   barrier
//   decide workgroup 0 (the others wait): the cut from the adjacency of S's nodes, ratio test with bound
//          flipping, the path u_in .. v in order; publishes the decision.  A subtree of more than ND_SMALL nodes
//          (1 % of the iterations) has its cut scanned by all workgroups first, one more barrier
//   ----   barrier
__global__ __launch_bounds__(ND_T) void k_nd_solve(NdProblem P, long long max_iters, double feas_tol) {
    __shared__ NdLds L;
    unsigned long long epoch = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = P.G, g = blockIdx.x;
    const int V = P.V;
    const long long E = P.E;
    const long long gtid = static_cast<long long>(g) * ND_T + tid, gsize = static_cast<long long>(G) * ND_T;
    const int gwave = g * ND_W + wave, nwaves = G * ND_W;
    NdShared *sh = P.sh;
    int4 *nd = P.nd;
    const int root = sh->root;
    long long iters = 0, flips = 0;
    long long status = -1;
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sum_cand = 0, sum_sub = 0, sum_path = 0, sum_range = 0, n_small = 0;
    long long tc = wall_clock64();
    auto tick = [&](int k) {
        const long long now = wall_clock64();
        tph[k] += now - tc;
        tc = now;
    };
    int prev_K = 0; // workgroup 0: path marks to clear

    while (true) {
        // ================================================== pass: apply the decision, score the tree arcs
        NdDec D;
        {
            const NdDec *d = &sh->dec;
            D.has = nd_ld(&d->has);
            D.enter = nd_ld(&d->enter);
            D.v_in = nd_ld(&d->v_in);
            D.b_pos = nd_ld(&d->b_pos);
            D.a_pos = nd_ld(&d->a_pos);
            D.n_sub = nd_ld(&d->n_sub);
            D.K = nd_ld(&d->K);
            D.npush = nd_ld(&d->npush);
            D.to_lower = nd_ld(&d->to_lower);
            D.lo = nd_ld(&d->lo);
            D.hi = nd_ld(&d->hi);
            D.newstart = nd_ld(&d->newstart);
            D.dy = nd_ld(&d->dy);
            D.enter_flow = nd_ld(&d->enter_flow);
            D.enter_cap = nd_ld(&d->enter_cap);
        }
        if (D.has && D.enter < 0) {
            status = 1;
            break;
        }
        const bool cached = D.has && D.K <= ND_PATH_LDS;
        const bool pcached = D.has && D.npush <= ND_PUSH_LDS;
        double bs = 0.0;
        int bn = 0x7fffffff;
        for (long long rb = 0; rb < V; rb += 2 * gsize) { // the same trip count in every lane: barriers inside
            const long long base = rb + gtid;
            // two nodes per lane and round; everything that depends only on the node is requested before the
            // decision is looked at, the flow and capacity of its arc right behind -- two levels of loads
            long long wk[2] = {base, base + gsize};
            int4 rk[2];
            double fk[2], ck[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { // three coalesced loads per node, no gathers
                const long long w = wk[k] < V ? wk[k] : 0;
                rk[k] = nd[w];
                fk[k] = P.nflow[w];
                ck[k] = P.ncap[w];
            }
            if (SX_ND_FINE_TICKS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tick(7);
            }
            if (rb == 0 && (cached || pcached)) { // (first round only) the decision's lists -> LDS
                __syncthreads();
                if (cached)
                    for (int i = tid; i < D.K; i += ND_T) {
                        L.spos[i] = nd_ld(&P.spos[i]);
                        L.ssize[i] = nd_ld(&P.ssize[i]);
                    }
                if (pcached && tid < D.npush) {
                    L.p_pt[tid] = nd_ld(&P.push_pt[tid]);
                    L.p_ph[tid] = nd_ld(&P.push_ph[tid]);
                    L.p_d[tid] = nd_ld(&P.push_d[tid]);
                }
                __syncthreads();
            }
            auto s_pos = [&](int i) { return cached ? L.spos[i] : nd_ld(&P.spos[i]); };
            auto s_size = [&](int i) { return cached ? L.ssize[i] : nd_ld(&P.ssize[i]); };
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const long long w = wk[k];
                if (w >= V) continue;
                int4 r = rk[k];
                double f = fk[k], c = ck[k];
                if (D.has) {
                    const bool inside = r.z >= D.a_pos && r.z < D.a_pos + D.n_sub;
                    const int pi = inside ? nd_ld(&P.pathidx[w]) : 0; // 1 + index on the path u_in .. v
                    if (w != root) {
                        // the arc that hangs w after the re-hang, and the subtree it closed before: a path node
                        // takes over the arc of the path node below it
                        int arc = r.y, iz = r.z, iw = r.w, owner = static_cast<int>(w);
                        bool changed = false;
                        if (pi > 0) { // (the leaving arc, v's old one, got its bound from workgroup 0)
                            changed = true;
                            if (pi == 1) {
                                arc = -1; // the entering arc
                                f = D.enter_flow;
                                c = D.enter_cap;
                            } else {
                                arc = nd_ld(&P.sarc[pi - 2]);
                                iz = s_pos(pi - 2);
                                iw = s_size(pi - 2);
                                owner = nd_ld(&P.snode[pi - 2]);
                                f = nd_ld(&P.sflow[pi - 2]);
                                c = nd_ld(&P.scap[pi - 2]);
                            }
                        }
                        if (arc >= 0) {
                            double d = 0.0;
                            bool any = false;
                            for (int i = 0; i < D.npush; ++i) { // the subtree separates the ends of a moved arc
                                const int pt = pcached ? L.p_pt[i] : nd_ld(&P.push_pt[i]), ph = pcached ? L.p_ph[i] : nd_ld(&P.push_ph[i]);
                                const bool ti = pt >= iz && pt < iz + iw, hi = ph >= iz && ph < iz + iw;
                                if (ti != hi) {
                                    const double x = pcached ? L.p_d[i] : nd_ld(&P.push_d[i]);
                                    d = d + (hi ? x : -x);
                                    any = true;
                                }
                            }
                            if (any) {
                                f = f + ((P.tail[arc] == owner) ? d : -d);
                                changed = true;
                            }
                        }
                        if (changed) {
                            P.nflow[w] = f;
                            if (pi > 0) P.ncap[w] = c;
                        }
                    }
                    int4 q = r;
                    if (pi > 0) {
                        q.x = pi == 1 ? D.v_in : nd_ld(&P.snode[pi - 2]);
                        q.y = pi == 1 ? D.enter : nd_ld(&P.sarc[pi - 2]);
                        q.w = pi == 1 ? D.n_sub : D.n_sub - s_size(pi - 2);
                    } else if (!inside) {
                        const bool anc_v = r.z <= D.a_pos && D.a_pos < r.z + r.w;
                        const bool anc_in = r.z <= D.b_pos && D.b_pos < r.z + r.w;
                        if (anc_v != anc_in) q.w = r.w + (anc_in ? D.n_sub : -D.n_sub);
                    }
                    if (inside) P.y[w] = P.y[w] + D.dy;
                    if (r.z >= D.lo && r.z < D.hi) { // S moves right behind v_in, re-rooted at u_in
                        const int t = r.z;
                        int nt;
                        if (inside) {
                            int l2 = 0, h2 = D.K - 1; // smallest i with t inside the old segment of path node i
                            while (l2 < h2) {
                                const int mid = (l2 + h2) >> 1;
                                const int q0 = s_pos(mid);
                                if (t >= q0 && t < q0 + s_size(mid)) h2 = mid;
                                else l2 = mid + 1;
                            }
                            const int i = l2;
                            int rel, off = 0;
                            if (i == 0) {
                                rel = t - s_pos(0);
                            } else {
                                const int hp = s_pos(i - 1), hs = s_size(i - 1); // the hole: the old segment of path node i - 1
                                off = hs;
                                rel = t < hp ? t - s_pos(i) : (hp - s_pos(i)) + (t - (hp + hs));
                            }
                            nt = D.newstart + off + rel;
                        } else {
                            nt = D.b_pos < D.a_pos ? t + D.n_sub : t - D.n_sub;
                        }
                        q.z = nt;
                        const int2 rec = P.noderec[w];
                        P.ordq[nt] = make_int4(static_cast<int>(w), rec.x, rec.y, 0);
                    }
                    if (q.x != r.x || q.y != r.y || q.z != r.z || q.w != r.w) nd[w] = q;
                    r = q;
                }
                if (w != root) {
                    const double lo = -f, hi = f - c;
                    const double viol = lo > hi ? lo : hi;
                    if (viol > feas_tol) {
                        const double sc = (viol * viol) / static_cast<double>(r.w);
                        if (sc > bs || (sc == bs && w < bn)) {
                            bs = sc;
                            bn = static_cast<int>(w);
                        }
                    }
                }
            }
        }
        if (SX_ND_FINE_TICKS) tick(5);
        nd_argmax(L, bs, bn);
        if (tid == 0) {
            P.part_s[g] = bs;
            P.part_n[g] = bn;
            if (g == 0) sh->cand_count = 0;
        }
        tick(0);
        nd_barrier(P.bar, epoch, G, P.nap_short, g == 0 || !P.lazy); // ---- B1: workgroup 0 reads everybody's pass
        tick(6);
        bs = 0.0;
        bn = 0x7fffffff;
        if (tid < G) {
            bs = nd_ld(&P.part_s[tid]);
            bn = nd_ld(&P.part_n[tid]);
        }
        nd_argmax(L, bs, bn);
        if (!(bs > 0.0)) {
            status = 0;
            break;
        }
        if (iters >= max_iters) {
            status = 3;
            break;
        }
        ++iters;
        const int v = bn;
        const int4 rv = nd_ld4(&nd[v]);
        const int a = rv.y, a_pos = rv.z, n_sub = rv.w;
        const double f_a = nd_ld(&P.nflow[v]), cap_a = nd_ld(&P.ncap[v]);
        const bool to_lower = -f_a >= f_a - cap_a;
        const double delta = to_lower ? -f_a : f_a - cap_a;
        const bool a_out = P.tail[a] == v;
        const int tau = to_lower ? (a_out ? -1 : 1) : (a_out ? 1 : -1);
        auto in_S = [&](int p) { return p >= a_pos && p < a_pos + n_sub; };
        const bool small = n_sub <= ND_SMALL;
        sum_sub += n_sub;
        n_small += small ? 1 : 0;
        // ================================================== the cut: candidates
        auto append = [&](bool elig, int j, double r, double c) {
            const unsigned long long m = __ballot(elig);
            if (m == 0ull) return;
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh->cand_count, __popcll(m));
            base = __shfl(base, 0, 64);
            if (elig) {
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                P.cand_j[slot] = j;
                P.cand_r[slot] = r;
                P.cand_c[slot] = c;
            }
        };
        auto consider = [&](int j, bool tail_in, int st, bool &elig, double &r, double &c) {
            // tail_in: the arc leaves S.  tau > 0: reduced costs of leaving arcs fall, of entering arcs rise
            elig = tau > 0 ? (tail_in ? st == ST_LOWER : st == ST_UPPER) : (tail_in ? st == ST_UPPER : st == ST_LOWER);
            if (elig) {
                const double rc = (P.cost[j] - P.y[P.tail[j]]) + P.y[P.head[j]];
                r = fabs(rc);
                c = P.cap[j];
            }
        };
        if (!small) { // a large subtree: all workgroups scan
            if (P.lazy && g != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // they read everybody's pass too
            if (static_cast<long long>(n_sub) * 4 <= V) { // the adjacency of S's nodes, one wave per node
                for (int t = a_pos + gwave; t < a_pos + n_sub; t += nwaves) {
                    const int w = P.ordq[t].x;
                    const int64_t p0 = P.rowptr[w], p1 = P.rowptr[w + 1];
                    for (int64_t p = p0; p < p1; p += 64) { // uniform trip count: the slots come from wave ballots
                        const int64_t q = p + lane;
                        bool elig = false;
                        int j = -1;
                        double r = 0.0, c = 0.0;
                        if (q < p1) {
                            j = P.rowarc[q];
                            const int st = P.state[j];
                            if (st != ST_TREE) {
                                const int tl = P.tail[j], hd = P.head[j];
                                const int o = tl == w ? hd : tl;
                                if (!in_S(nd[o].z)) consider(j, tl == w, st, elig, r, c);
                            }
                        }
                        append(elig, j, r, c);
                    }
                }
            } else { // all arcs
                for (long long j0 = static_cast<long long>(gwave) * 64; j0 < E; j0 += static_cast<long long>(nwaves) * 64) {
                    const long long j = j0 + lane;
                    bool elig = false;
                    double r = 0.0, c = 0.0;
                    if (j < E) {
                        const int st = P.state[j];
                        if (st != ST_TREE) {
                            const bool ti = in_S(nd[P.tail[j]].z), hi = in_S(nd[P.head[j]].z);
                            if (ti != hi) consider(static_cast<int>(j), ti, st, elig, r, c);
                        }
                    }
                    append(elig, static_cast<int>(j), r, c);
                }
            }
            tick(1);
            nd_barrier(P.bar, epoch, G, P.nap_short); // ---- B2 (large subtrees only)
            tick(6);
        }
        // ================================================== workgroup 0 decides
        if (g == 0) {
            // marks of the previous path
            for (int i = tid; i < prev_K; i += ND_T) P.pathidx[P.snode[i]] = 0;
            if (tid == 0) L.ccount = 0;
            if (small) {
                // the entries of the rows of S's nodes, flattened over the lanes: node t of S owns the entries
                // [off[t], off[t + 1]) of the flattened list (n_sub <= ND_SMALL <= ND_T: one node per lane)
                int w = -1, p0 = 0, deg = 0;
                if (tid < n_sub) {
                    const int4 rec = P.ordq[a_pos + tid];
                    w = rec.x;
                    p0 = rec.y;
                    deg = rec.z;
                }
                int incl = deg;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                __syncthreads();
                if (lane == 63) L.scan[wave] = incl;
                __syncthreads();
                int before = 0, T = 0;
#pragma unroll
                for (int q = 0; q < ND_W; ++q) {
                    before += q < wave ? L.scan[q] : 0;
                    T += L.scan[q];
                }
                if (tid < n_sub) {
                    L.s_node[tid] = w;
                    L.s_p0[tid] = p0;
                    L.s_off[tid] = before + incl - deg;
                }
                __syncthreads();
                auto append_lds = [&](bool elig, int j, double r, double c) {
                    const unsigned long long m = __ballot(elig);
                    if (m == 0ull) return;
                    int base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(&L.ccount, __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = __shfl(base, 0, 64);
                    if (elig) {
                        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < ND_LCAP) {
                            L.c_j[slot] = j;
                            L.c_r[slot] = r;
                            L.c_c[slot] = c;
                        }
                        P.cand_j[slot] = j; // the list in memory serves counts beyond ND_LCAP
                        P.cand_r[slot] = r;
                        P.cand_c[slot] = c;
                    }
                };
                constexpr int U = 4; // entries per lane and round: their loads are in flight together
                for (int e0 = 0; e0 < T; e0 += U * ND_T) { // uniform trip count: the slots come from wave ballots
                    // three levels of dependent loads: (entry -> arc, other end, y of this end) -> (state, cost,
                    // capacity, position and y of the other end) -> candidate
                    int jj[U], oo[U], st[U], po[U];
                    double yw[U], yo[U], cs[U], cp[U];
                    bool live[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const int e = e0 + k * ND_T + tid;
                        live[k] = e < T;
                        jj[k] = 0;
                        oo[k] = 0;
                        yw[k] = 0.0;
                        if (live[k]) {
                            int l2 = 0, h2 = n_sub - 1; // largest t with off[t] <= e
                            while (l2 < h2) {
                                const int mid = (l2 + h2 + 1) >> 1;
                                if (L.s_off[mid] <= e) l2 = mid;
                                else h2 = mid - 1;
                            }
                            const int p = L.s_p0[l2] + (e - L.s_off[l2]);
                            jj[k] = P.rowarc[p];
                            oo[k] = P.rowother[p];
                            yw[k] = P.y[L.s_node[l2]];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        st[k] = ST_TREE;
                        po[k] = 0;
                        yo[k] = cs[k] = cp[k] = 0.0;
                        if (live[k]) {
                            const int o = oo[k] < 0 ? ~oo[k] : oo[k];
                            st[k] = P.state[jj[k]];
                            po[k] = nd[o].z;
                            yo[k] = P.y[o];
                            cs[k] = P.cost[jj[k]];
                            cp[k] = P.cap[jj[k]];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        bool elig = false;
                        const bool tail_in = oo[k] >= 0; // this end (inside S) is the arc's tail
                        if (st[k] != ST_TREE && !in_S(po[k]))
                            elig = tau > 0 ? (tail_in ? st[k] == ST_LOWER : st[k] == ST_UPPER)
                                           : (tail_in ? st[k] == ST_UPPER : st[k] == ST_LOWER);
                        const double rc = tail_in ? (cs[k] - yw[k]) + yo[k] : (cs[k] - yo[k]) + yw[k];
                        append_lds(elig, jj[k], fabs(rc), cp[k]);
                    }
                }
                __syncthreads();
                if (tid == 0) sh->cand_count = L.ccount;
            }
            tick(2);
            // ---- ratio test with bound flipping: candidates in ascending (|reduced cost|, arc)
            __syncthreads();
            const int C = small ? L.ccount : sh->cand_count;
            sum_cand += C;
            if (!small && C <= ND_LCAP) { // the wide scan left its list in memory
                for (int i = tid; i < C; i += ND_T) {
                    L.c_j[i] = P.cand_j[i];
                    L.c_r[i] = P.cand_r[i];
                    L.c_c[i] = P.cand_c[i];
                }
                __syncthreads();
            }
            double remaining = delta, last_r = -1.0;
            int last_j = -1, npush = 0, enter = -1;
            double theta = 0.0, cap_e = 0.0;
            if (C <= ND_LCAP) { // one wave, candidates in LDS: a round is a strided scan and a wave reduction
                if (wave == 0) {
                    // every lane keeps the four smallest of its strided share in registers, sorted; a round pops
                    // the smallest head of the wave.  A lane that runs dry while it has more re-reads its share
                    double tr[4], tcap[4];
                    int tj[4];
                    auto refill = [&](double lr, int lj) { // the lane's four smallest candidates above (lr, lj)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            tj[k] = -1;
                            tr[k] = 0.0;
                            tcap[k] = 0.0;
                        }
                        int seen = 0;
                        for (int i = lane; i < C; i += 64) {
                            double r2 = L.c_r[i];
                            int j2 = L.c_j[i];
                            if (r2 > lr || (r2 == lr && j2 > lj)) {
                                double c2 = L.c_c[i];
                                ++seen;
#pragma unroll
                                for (int k = 0; k < 4; ++k) { // insertion: the larger one travels on
                                    const bool first = j2 >= 0 && (tj[k] < 0 || r2 < tr[k] || (r2 == tr[k] && j2 < tj[k]));
                                    if (first) {
                                        const double rr = tr[k], cc = tcap[k];
                                        const int jj2 = tj[k];
                                        tr[k] = r2;
                                        tcap[k] = c2;
                                        tj[k] = j2;
                                        r2 = rr;
                                        c2 = cc;
                                        j2 = jj2;
                                    }
                                }
                            }
                        }
                        return seen > 4;
                    };
                    bool more = refill(-1.0, -1);
                    while (true) {
                        double r = tr[0], c = tcap[0];
                        int j = tj[0];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const double r2 = __shfl_xor(r, o, 64), c2 = __shfl_xor(c, o, 64);
                            const int j2 = __shfl_xor(j, o, 64);
                            if (j2 >= 0 && (j < 0 || r2 < r || (r2 == r && j2 < j))) {
                                r = r2;
                                j = j2;
                                c = c2;
                            }
                        }
                        if (j < 0) break; // every candidate passed and the arc is still infeasible: no entering arc
                        if (c < remaining) {
                            if (lane == 0) P.tmp[npush] = j;
                            ++npush;
                            remaining = remaining - c;
                            last_r = r;
                            last_j = j;
                            const bool mine = tj[0] == j;
                            if (mine) { // pop
#pragma unroll
                                for (int k = 0; k < 3; ++k) {
                                    tr[k] = tr[k + 1];
                                    tcap[k] = tcap[k + 1];
                                    tj[k] = tj[k + 1];
                                }
                                tj[3] = -1;
                                if (tj[0] < 0 && more) more = refill(r, j);
                            }
                        } else {
                            enter = j;
                            theta = r;
                            cap_e = c;
                            break;
                        }
                    }
                    if (lane == 0) {
                        L.res_enter = enter;
                        L.res_npush = npush;
                        L.res_theta = theta;
                        L.res_cap = cap_e;
                        L.res_remaining = remaining;
                    }
                }
                __syncthreads();
                enter = L.res_enter;
                npush = L.res_npush;
                theta = L.res_theta;
                cap_e = L.res_cap;
                remaining = L.res_remaining;
            } else {
                double r0 = 0.0, c0 = 0.0; // this lane's first candidate, kept in registers over the rounds
                int j0 = -1;
                if (tid < C) {
                    r0 = P.cand_r[tid];
                    c0 = P.cand_c[tid];
                    j0 = P.cand_j[tid];
                }
                while (true) {
                    double r = 0.0, c = 0.0;
                    int j = -1;
                    if (j0 >= 0 && (r0 > last_r || (r0 == last_r && j0 > last_j))) {
                        r = r0;
                        c = c0;
                        j = j0;
                    }
                    for (int i = tid + ND_T; i < C; i += ND_T) {
                        const double r2 = P.cand_r[i];
                        const int j2 = P.cand_j[i];
                        if ((r2 > last_r || (r2 == last_r && j2 > last_j)) && (j < 0 || r2 < r || (r2 == r && j2 < j))) {
                            r = r2;
                            j = j2;
                            c = P.cand_c[i];
                        }
                    }
                    nd_argmin(L, r, j, c);
                    if (j < 0) break; // every candidate passed and the arc is still infeasible: no entering arc
                    if (c < remaining) {
                        if (tid == 0) P.tmp[npush] = j;
                        ++npush;
                        remaining = remaining - c;
                        last_r = r;
                        last_j = j;
                    } else {
                        enter = j;
                        theta = r;
                        cap_e = c;
                        break;
                    }
                }
            }
            __syncthreads();
            tick(3);
            NdDec N;
            memset(&N, 0, sizeof(N));
            N.has = 1;
            N.enter = enter;
            if (enter >= 0) {
                const int tl = P.tail[enter], hd = P.head[enter];
                const int pt = nd[tl].z, ph = nd[hd].z;
                const bool t_in = in_S(pt);
                const int u_in = t_in ? tl : hd, p_uin = t_in ? pt : ph, b_pos = t_in ? ph : pt;
                // ---- the path u_in .. v inside S in order: ancestors of u_in, by position (= from v downwards)
                int K = 0;
                for (int t0 = 0; t0 < n_sub; t0 += ND_T) {
                    const int t = t0 + tid;
                    bool on = false;
                    if (t < n_sub) {
                        const int w = small ? L.s_node[t] : P.ordq[a_pos + t].x;
                        const int4 r = nd[w];
                        on = r.z <= p_uin && p_uin < r.z + r.w;
                    }
                    K += on ? 1 : 0;
                }
                K = nd_sum(L, K);
                int before = 0;
                for (int t0 = 0; t0 < n_sub; t0 += ND_T) {
                    const int t = t0 + tid;
                    bool on = false;
                    int w = 0;
                    int4 r = make_int4(0, 0, 0, 0);
                    if (t < n_sub) {
                        w = small ? L.s_node[t] : P.ordq[a_pos + t].x;
                        r = nd[w];
                        on = r.z <= p_uin && p_uin < r.z + r.w;
                    }
                    const unsigned long long m = __ballot(on);
                    __syncthreads();
                    if (lane == 0) L.scan[wave] = __popcll(m);
                    __syncthreads();
                    int wbefore = 0, total = 0;
#pragma unroll
                    for (int q = 0; q < ND_W; ++q) {
                        wbefore += q < wave ? L.scan[q] : 0;
                        total += L.scan[q];
                    }
                    if (on) {
                        const int i = K - 1 - (before + wbefore + __popcll(m & ((1ull << lane) - 1ull)));
                        P.snode[i] = w;
                        P.sarc[i] = r.y;
                        P.spos[i] = r.z;
                        P.ssize[i] = r.w;
                        P.sflow[i] = P.nflow[w];
                        P.scap[i] = P.ncap[w];
                        P.pathidx[w] = i + 1;
                    }
                    before += total;
                }
                // ---- moved arcs: the passed ones change bound, the entering arc takes what is left
                for (int i = tid; i < npush; i += ND_T) {
                    const int j = P.tmp[i];
                    const int st = P.state[j];
                    const double c = P.cap[j];
                    P.state[j] = static_cast<int8_t>(-st);
                    P.flow[j] = st == ST_LOWER ? c : 0.0;
                    P.push_pt[i] = nd[P.tail[j]].z;
                    P.push_ph[i] = nd[P.head[j]].z;
                    P.push_d[i] = st == ST_LOWER ? c : -c;
                }
                double enter_flow = 0.0;
                if (tid == 0) {
                    const int st = P.state[enter];
                    const double d = st == ST_LOWER ? remaining : -remaining;
                    enter_flow = (st == ST_LOWER ? 0.0 : cap_e) + d;
                    P.state[enter] = ST_TREE;
                    P.flow[a] = to_lower ? 0.0 : cap_a; // the leaving arc lands exactly on its bound
                    P.push_pt[npush] = pt;
                    P.push_ph[npush] = ph;
                    P.push_d[npush] = d;
                    P.state[a] = static_cast<int8_t>(to_lower ? ST_LOWER : ST_UPPER);
                }
                N.v_in = t_in ? hd : tl;
                N.b_pos = b_pos;
                N.npush = npush + 1;
                N.K = K;
                N.a_pos = a_pos;
                N.n_sub = n_sub;
                N.to_lower = to_lower ? 1 : 0;
                N.dy = tau > 0 ? theta : -theta;
                N.enter_flow = enter_flow; // (lane 0's value is the one that is stored)
                N.enter_cap = cap_e;
                N.lo = a_pos < b_pos + 1 ? a_pos : b_pos + 1;
                N.hi = a_pos + n_sub > b_pos + 1 ? a_pos + n_sub : b_pos + 1;
                N.newstart = b_pos < a_pos ? b_pos + 1 : b_pos + 1 - n_sub;
                prev_K = K;
                flips += npush;
                sum_path += K;
                sum_range += N.hi - N.lo;
                (void)u_in;
            }
            if (tid == 0) sh->dec = N;
            tick(4);
        }
        nd_barrier(P.bar, epoch, G, P.nap_long, !P.lazy); // ---- B3: the pass reads its own nodes + nd_ld()
        tick(6);
    }

    // ====================================================== results
    nd_barrier(P.bar, epoch, G, P.nap_short);
    // potentials once more from the tree itself (pointer jumping): the duals handed back are the tree's own
    for (long long w = gtid; w < V; w += gsize) {
        const int4 r = nd[w];
        double c = 0.0;
        int up = static_cast<int>(w);
        if (r.x >= 0) {
            c = (P.tail[r.y] == w) ? P.cost[r.y] : -P.cost[r.y];
            up = r.x;
        }
        P.acc[0][w] = c;
        P.anc[0][w] = up;
    }
    nd_barrier(P.bar, epoch, G, P.nap_short);
    int cur = 0;
    for (int round = 0; round < 32; ++round) {
        int moved = 0;
        for (long long w = gtid; w < V; w += gsize) {
            const int p = P.anc[cur][w];
            const int p2 = P.anc[cur][p];
            P.acc[1 - cur][w] = P.acc[cur][w] + P.acc[cur][p];
            P.anc[1 - cur][w] = p2;
            if (p2 != p) moved = 1;
        }
        moved = nd_sum(L, moved);
        if (tid == 0) P.part_cnt[g] = moved;
        nd_barrier(P.bar, epoch, G, P.nap_short);
        int any = tid < G ? P.part_cnt[tid] : 0;
        any = nd_sum(L, any);
        cur = 1 - cur;
        nd_barrier(P.bar, epoch, G, P.nap_short); // part_cnt is rewritten in the next round
        if (!any) break;
    }
    for (long long w = gtid; w < V; w += gsize) {
        P.y[w] = P.acc[cur][w];
        const int4 r = nd[w];
        if (r.x >= 0) P.flow[r.y] = P.nflow[w]; // the tree arcs' flows were kept by node
    }
    nd_barrier(P.bar, epoch, G, P.nap_short);
    // objective and largest bound violation: fixed order (lane-strided partials, workgroup order)
    double part = 0.0, worst = 0.0;
    for (long long e = gtid; e < E; e += gsize) {
        const double f = P.flow[e];
        part += P.cost[e] * f;
        const double viol = -f > f - P.cap[e] ? -f : f - P.cap[e];
        if (viol > worst) worst = viol;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_down(part, o, 64);
        const double w2 = __shfl_down(worst, o, 64);
        worst = w2 > worst ? w2 : worst;
    }
    __syncthreads();
    if (lane == 0) {
        L.d[wave] = part;
        L.d2[wave] = worst;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0, wm = 0.0;
        for (int w = 0; w < ND_W; ++w) {
            tot += L.d[w];
            wm = L.d2[w] > wm ? L.d2[w] : wm;
        }
        P.part_s[g] = tot;
        P.acc[1 - cur][g] = wm; // (free again)
    }
    nd_barrier(P.bar, epoch, G, P.nap_short);
    if (g == 0 && tid == 0) {
        double tot = 0.0, wm = 0.0;
        for (int k = 0; k < G; ++k) {
            tot += P.part_s[k];
            wm = P.acc[1 - cur][k] > wm ? P.acc[1 - cur][k] : wm;
        }
        sh->obj = tot;
        sh->max_violation = wm;
        sh->iters = iters;
        sh->flips += flips;
        sh->status = status;
        for (int k = 0; k < 8; ++k) sh->t_phase[k] = tph[k];
        sh->sum_cand = sum_cand;
        sh->sum_sub = sum_sub;
        sh->sum_path = sum_path;
        sh->sum_range = sum_range;
        sh->n_small = n_small;
    }
}

__global__ __launch_bounds__(256) void k_nd_outputs(int64_t V, int64_t E, const int8_t *__restrict__ state, int root,
                                                    int8_t *__restrict__ vbasis, int8_t *__restrict__ cbasis) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (vbasis && i < E) vbasis[i] = static_cast<int8_t>(state[i] == ST_TREE ? 0 : state[i] == ST_LOWER ? -1 : -2);
    if (cbasis && i < V) cbasis[i] = static_cast<int8_t>(i == root ? 0 : -1);
}

} // namespace

SX_API int sx_netdual_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                          const double *u, const int8_t *vbasis_in, const int8_t *cbasis_in, int64_t max_iter,
                          double feas_tol, double *x_out, double *y_out, int8_t *vbasis_out, int8_t *cbasis_out,
                          sx_simplex_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && b && c && l && u && vbasis_in && cbasis_in && result, "NULL argument");
    SX_REQUIRE(A->ctx->device == ctx->device, "matrix lives on another device");
    SX_REQUIRE(A->csc_ptr && A->csr_ptr, "matrix needs both layouts");
    memset(result, 0, sizeof(*result));
    result->status = 5;
    const int64_t V = A->m, E = A->n;
    if (V < 2 || E < 1 || V >= (static_cast<int64_t>(1) << 30) || E >= (static_cast<int64_t>(1) << 30) || A->nnz != 2 * E)
        return SX_OK; // not a network: the caller takes another method
    if (ctx->opt_netdual == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    // every temporary of the call in one block of the context (about 80 B per arc and 300 B per node)
    SX_TRY(sx_reserve3(ctx, static_cast<size_t>(96) * static_cast<size_t>(E) + static_cast<size_t>(384) * static_cast<size_t>(V) + (1u << 20)));
    sx_arena pool(ctx);
    NdProblem P;
    memset(&P, 0, sizeof(P));
    P.V = static_cast<int>(V);
    P.E = E;
    P.rowptr = A->csr_ptr;
    P.rowarc = A->csr_idx;
    P.cost = c;
    P.cap = u;
    double *xn, *beff;
    SX_TRY(pool.get(E, &P.tail));
    SX_TRY(pool.get(E, &P.head));
    SX_TRY(pool.get(E, &P.state));
    SX_TRY(pool.get(E, &xn));
    SX_TRY(pool.get(V, &beff));
    SX_TRY(pool.get(1, &P.sh));
    SX_HIP(hipMemsetAsync(P.sh, 0, sizeof(NdShared), s));
    hipLaunchKernelGGL(k_nd_endpoints, dim3(static_cast<unsigned>((E + 255) / 256)), dim3(256), 0, s, E, A->csc_ptr,
                       A->csc_idx, A->csc_val, l, u, vbasis_in, P.tail, P.head, P.state, P.sh);
    hipLaunchKernelGGL(k_nd_root, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, V, cbasis_in, P.sh);
    NdShared sh;
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (sh.not_network || sh.ntree != V - 1 || sh.nroot != 1) return SX_OK;
    // no capacitated arc outside the tree (the OT crossovers): nothing can be flipped, the start is dual feasible
    // only if it is optimal already -- the primal method's case, decided here before any set-up work
    if (sh.any_nontree && !sh.any_capacity && ctx->opt_netdual < 1) return SX_OK;
    if (x_out) P.flow = x_out;
    else SX_TRY(pool.get(E, &P.flow));
    if (y_out) P.y = y_out;
    else SX_TRY(pool.get(V, &P.y));
    // tree arrays live in the context: kept from solve to solve (a column generation's next round starts from the
    // tree this one ends with)
    bool kept = ctx->nd_tree && ctx->nd_tree_V == V && ctx->nd_tree_root == sh.root;
    if (!kept) {
        if (ctx->nd_tree) (void)sx_dfree(ctx->nd_tree);
        if (ctx->nd_order) (void)sx_dfree(ctx->nd_order);
        if (ctx->nd_y) (void)sx_dfree(ctx->nd_y);
        ctx->nd_tree = ctx->nd_order = nullptr;
        ctx->nd_y = nullptr;
        ctx->nd_tree_V = 0;
        SX_HIP(sx_dmalloc(&ctx->nd_tree, sizeof(int4) * static_cast<size_t>(V)));
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&ctx->nd_order), sizeof(int32_t) * static_cast<size_t>(V)));
        SX_HIP(sx_dmalloc(reinterpret_cast<void **>(&ctx->nd_y), sizeof(double) * static_cast<size_t>(V)));
    }
    P.nd = static_cast<int4 *>(ctx->nd_tree);
    P.order = ctx->nd_order;
    SX_TRY(pool.get(V > E ? V : E, &P.tmp)); // preorder move (V) and passed arcs of the ratio test (E)
    SX_TRY(pool.get(V, &P.first_child));
    SX_TRY(pool.get(V, &P.next_sib));
    SX_TRY(pool.get(V, &P.e));
    SX_TRY(pool.get((V + 255) / 256, &P.bsum));
    SX_TRY(pool.get(ND_GMAX, &P.part_s));
    SX_TRY(pool.get(ND_GMAX, &P.part_n));
    SX_TRY(pool.get(ND_GMAX, &P.part_cnt));
    SX_TRY(pool.get(E, &P.cand_j));
    SX_TRY(pool.get(E, &P.cand_r));
    SX_TRY(pool.get(E, &P.cand_c));
    SX_TRY(pool.get(E + 1, &P.push_pt));
    SX_TRY(pool.get(E + 1, &P.push_ph));
    SX_TRY(pool.get(E + 1, &P.push_d));
    SX_TRY(pool.get(V, &P.snode));
    SX_TRY(pool.get(V, &P.sarc));
    SX_TRY(pool.get(V, &P.spos));
    SX_TRY(pool.get(V, &P.ssize));
    SX_TRY(pool.get(V, &P.pathidx));
    SX_TRY(pool.get(2 * E, &P.rowother));
    SX_TRY(pool.get(V, &P.noderec));
    SX_TRY(pool.get(V, &P.ordq));
    SX_TRY(pool.get(V, &P.nflow));
    SX_TRY(pool.get(V, &P.ncap));
    SX_TRY(pool.get(V, &P.sflow));
    SX_TRY(pool.get(V, &P.scap));
    SX_HIP(hipMemsetAsync(P.pathidx, 0, sizeof(int32_t) * static_cast<size_t>(V), s));
    for (int k = 0; k < 2; ++k) {
        SX_TRY(pool.get(V > ND_GMAX ? V : ND_GMAX, &P.acc[k]));
        SX_TRY(pool.get(V, &P.anc[k]));
    }
    struct Events { // destroyed on every return path
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events() {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
        hipEvent_t &operator[](int k) { return e[k]; }
    } ev;
    for (int k = 0; k < 3; ++k) SX_HIP(hipEventCreate(&ev[k]));
    SX_HIP(hipEventRecord(ev[0], s));
    // ---- set-up: tree, potentials, dual feasibility by flips, tree flows
    if (kept) { // is it this basis?
        int *d_differs;
        SX_TRY(pool.get(1, &d_differs));
        SX_HIP(hipMemsetAsync(d_differs, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_nd_check_kept, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, P, ctx->nd_y, d_differs);
        int differs = 0;
        SX_HIP(hipMemcpyAsync(&differs, d_differs, sizeof(int), hipMemcpyDeviceToHost, s));
        SX_HIP(hipStreamSynchronize(s));
        kept = !differs;
    }
    ctx->nd_tree_V = 0; // (valid again once this solve has run)
    if (kept) SX_HIP(hipMemcpyAsync(P.y, ctx->nd_y, sizeof(double) * static_cast<size_t>(V), hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(k_nd_tree, dim3(1), dim3(ND_T), 0, s, P);
    hipLaunchKernelGGL(k_nd_adjacency, dim3(static_cast<unsigned>((V * 64 + 255) / 256)), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_nd_flip, dim3(static_cast<unsigned>((E + 255) / 256)), dim3(256), 0, s, P, xn);
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (sh.status == 5 || sh.not_network) return SX_OK; // not a tree / an uncapacitated arc would have to flip
    hipLaunchKernelGGL(k_nd_ordrec, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, P); // (order[] is valid now)
    SX_TRY(sx_score_rows_dev(ctx, A, xn, b, nullptr, 0.0, beff, nullptr)); // b - A x_N, sums in stored order
    hipLaunchKernelGGL(k_nd_excess, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, P, beff);
    hipLaunchKernelGGL(k_nd_initflows, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, P);
    // ---- the cooperative grid: as many workgroups as the passes can use, all resident
    int per_cu = 0;
    SX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_nd_solve, ND_T, 0));
    int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    // one node per lane in the pass up to 64 workgroups, two beyond: more workgroups cost more at the two barriers
    // (and their polling slows workgroup 0) than the second node costs in the pass (profiles/r02/netdual_grid_sweep.txt)
    int64_t want = (V + ND_T - 1) / ND_T;
    if (want > 64) want = std::max<int64_t>(64, (V + 2 * ND_T - 1) / (2 * ND_T));
    int G = static_cast<int>(want < 1 ? 1 : want);
    if (ctx->opt_nd_grid > 0) G = ctx->opt_nd_grid;
    if (G > ND_GMAX) G = ND_GMAX;
    if (G > per_cu * cus) G = per_cu * cus;
    if (G < 1) G = 1;
    P.G = G;
    SX_TRY(pool.get(1, &P.bar));
    SX_HIP(hipMemsetAsync(P.bar, 0, sizeof(unsigned long long), s));
    {
        static const char *e1 = getenv("SX_ND_NAP_SHORT"), *e2 = getenv("SX_ND_NAP_LONG"); // experiments
        P.nap_short = e1 ? atoi(e1) : 0;
        P.nap_long = e2 ? atoi(e2) : 2;
        static const char *e3 = getenv("SX_ND_LAZY");
        P.lazy = e3 ? atoi(e3) : 1;
    }
    // default limit: several times what the column-generation rounds measured need (config 4: 0.2 M of 1.2 M);
    // hitting it returns status 3 and the caller goes on with the primal method from the basis it gave
    long long limit = max_iter > 0 ? max_iter : 4 * V + E / 2 + 100000;
    void *args[] = {&P, &limit, &feas_tol};
    SX_HIP(hipEventRecord(ev[1], s));
    SX_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_nd_solve), dim3(static_cast<unsigned>(G)), dim3(ND_T),
                                      args, 0, s));
    SX_HIP(hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(k_nd_keep_order, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, P);
    SX_HIP(hipMemcpyAsync(ctx->nd_y, P.y, sizeof(double) * static_cast<size_t>(V), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_nd_outputs, dim3(static_cast<unsigned>(((E > V ? E : V) + 255) / 256)), dim3(256), 0, s, V, E,
                       P.state, sh.root, vbasis_out, cbasis_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    const double it_n = sh.iters > 0 ? static_cast<double>(sh.iters) : 1.0;
    float ms_setup = 0.f, ms_solve = 0.f;
    (void)hipEventElapsedTime(&ms_setup, ev[0], ev[1]);
    (void)hipEventElapsedTime(&ms_solve, ev[1], ev[2]);
    if (getenv("SX_NS_PROFILE"))
        fprintf(stderr, "[sx_netdual] V=%lld E=%lld grid=%d iterations=%lld flips=%lld status=%lld | set-up %.1f ms, solve %.1f ms | "
                        "us per iteration: pass %.2f wide cut %.2f cut %.2f ratio %.2f publish %.2f (pass compute %.2f) barriers %.2f (pass loads %.2f) | per iteration: "
                        "candidates %.1f, subtree %.1f (alone %.0f%%), path %.1f, positions moved %.1f\n",
                (long long)V, (long long)E, G, sh.iters, sh.flips, sh.status, ms_setup, ms_solve,
                sh.t_phase[0] * 0.01 / it_n, sh.t_phase[1] * 0.01 / it_n, sh.t_phase[2] * 0.01 / it_n, sh.t_phase[3] * 0.01 / it_n,
                sh.t_phase[4] * 0.01 / it_n, sh.t_phase[5] * 0.01 / it_n, sh.t_phase[6] * 0.01 / it_n, sh.t_phase[7] * 0.01 / it_n, sh.sum_cand / it_n,
                sh.sum_sub / it_n, 100.0 * sh.n_small / it_n, sh.sum_path / it_n, sh.sum_range / it_n);
    if (sh.status == 0 || sh.status == 1 || sh.status == 3) { // the arrays describe the tree handed back
        ctx->nd_tree_V = V;
        ctx->nd_tree_root = sh.root;
    }
    result->status = sh.status;
    result->iters = sh.iters;
    result->phase1_iters = sh.flips;        // arcs moved bound to bound (at the start and by the ratio test)
    result->warm_start_used = kept ? 2 : 1; // 2: the tree arrays of the previous solve were reused
    result->obj = sh.obj;
    result->max_violation = sh.max_violation;
    return SX_OK;
}
