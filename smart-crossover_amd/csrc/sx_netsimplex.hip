// K16n: primal network simplex on a spanning-tree basis -- the re-solves of the network crossover
// (network_methods/net_manager.py:211-222 solve_subproblem -> solve_mcf(..., warm_start_basis); column
// generation of network_methods/algorithms.py:109-140).  The sub-problems are pure networks: every column of
// A holds one +1 (the arc's tail row) and one -1 (its head row), 0 <= x <= u, A x = b, and a basis is a
// spanning tree.  A pivot touches one cycle of the tree, so there is no m x m inverse: the dense-inverse
// simplex of sx_simplex.hip needs 8 m^2 bytes (137 GB at config 4's 131,073 rows) and 3x the pivots.
//
// Shape of the computation.  A pivot is a chain of dependent steps (price -> cycle -> ratio test -> re-hang),
// each short, so the whole solve is ONE persistent workgroup of 1024 lanes that never returns to the host:
//   price    all lanes: a block of arcs, reduced cost c - y[tail] + y[head], best violation by wave shuffles
//            (block search: the cursor moves on, a full empty round proves optimality);
//   cycle    no climb: the tree is kept in PREORDER (pos[v], size[v], order[]), so "u is an ancestor of w" is
//            pos[u] <= pos[w] < pos[u] + size[u], and the two tree paths from the entering arc's ends to their
//            join are the nodes that are an ancestor of exactly one end -- every lane tests its share of the
//            nodes (a first version climbed with two lanes: 1 us per hop of dependent L2 loads, 90 us per pivot);
//   ratio    strict '<' on the side that loses flow first, '<=' on the other (the last blocking arc seen
//            from the join leaves: strongly feasible trees, no cycling from a strongly feasible start);
//   augment  all lanes over the two recorded paths;
//   re-hang  the cut subtree S is re-rooted at the entering arc's end: its new preorder is the old one cut
//            into 2k+1 pieces (k = path length) whose offsets telescope to old sizes, so every lane moves
//            its elements of the affected range [lo, hi) of order[] independently (binary search over the
//            path), shifts the potentials of S by the entering arc's reduced cost and rewrites pos[].
// No pointer chasing anywhere in a pivot: every step is a flat pass of all lanes over nodes, path entries or a
// range of the preorder array.
//
// Start: the given basis must be a spanning tree (vbasis 0 on V-1 arcs, cbasis 0 on the root row) whose tree
// flows respect the bounds; anything else returns status 5 and the caller takes the general simplex
// (sx_simplex.hip, which has a phase 1).  Set-up (parents by hooking sweeps, then one depth-first pass for
// preorder, sizes, potentials and tree flows) runs in the same kernel; children are visited in ascending node
// order, so the whole solve is deterministic.
#include "sx_internal.h"

#include <cmath>
#include <cstdlib>

namespace {

constexpr int NS_T = 1024;         // lanes of the persistent workgroup
constexpr int NS_W = NS_T / 64;    // its waves
constexpr int ST_TREE = 0, ST_LOWER = 1, ST_UPPER = -1;
constexpr int UNK = -2;

struct NsShared { // results and flags, global memory
    long long iters;
    long long status; // 0 optimal, 2 unbounded, 3 iteration limit, 4 tree arrays inconsistent (a bug), 5 not applicable
    double obj;
    double max_violation;
    int not_network; // a column that is not (+1, -1), a lower bound != 0, an arc at an infinite upper bound
    int ntree;       // arcs coded basic
    int nroot;       // rows coded basic
    int root;
    long long t_phase[8]; // shader clocks per phase, lane 0 (SX_NS_PROFILE=1 prints them)
    long long blocks;     // pricing blocks scanned
    long long path_sum;   // nodes on the cycles, all pivots
    long long range_sum;  // preorder positions rewritten, all pivots
};

struct NsProblem {
    int V;
    long long E;
    int32_t *tail, *head; // [E]
    const double *cost, *cap;
    double *flow;   // [E] (= the x output)
    int8_t *state;  // [E]
    const double *beff; // [V] b - A x_N
    int4 *nd;       // [V] {parent, pred, pos, size}
    double *y;      // [V]
    int32_t *order; // [V]
    int32_t *tmp;   // [V]
    int32_t *first_child, *next_sib; // [V] set-up only
    double *exc;    // [V] set-up only
    int32_t *pnode[2], *parc[2], *ppos[2], *psize[2]; // [V] each: the nodes of the two tree paths, as found
    int32_t *snode, *sarc, *spos, *ssize;             // [V] each: the path u_in .. u_out in order
    int8_t *pdec[2];
    double *acc[2]; // [V] each: potentials refresh
    int32_t *anc[2];
    NsShared *sh;
};

// ------------------------------------------------------------------ arcs from the columns of A
__global__ __launch_bounds__(256) void k_ns_endpoints(int64_t E, const int64_t *__restrict__ colptr,
                                                      const int32_t *__restrict__ rowidx,
                                                      const double *__restrict__ val, const double *__restrict__ l,
                                                      const double *__restrict__ u, const int8_t *__restrict__ vbasis,
                                                      int32_t *__restrict__ tail, int32_t *__restrict__ head,
                                                      int8_t *__restrict__ state, double *__restrict__ xn, NsShared *sh) {
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
    int st = code == 0 ? ST_TREE : code == -2 ? ST_UPPER : ST_LOWER;
    if (code != 0 && code != -1 && code != -2) ok = false;
    double x = 0.0;
    if (st == ST_UPPER) {
        x = u[j];
        if (isinf(x)) ok = false;
    }
    tail[j] = t;
    head[j] = h;
    state[j] = static_cast<int8_t>(st);
    xn[j] = x;
    if (!ok) sh->not_network = 1;
    if (st == ST_TREE) atomicAdd(&sh->ntree, 1);
}

__global__ __launch_bounds__(256) void k_ns_root(int64_t V, const int8_t *__restrict__ cbasis, NsShared *sh) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= V) return;
    if (cbasis[i] == 0) {
        atomicAdd(&sh->nroot, 1);
        sh->root = static_cast<int>(i); // only meaningful when nroot ends up 1
    }
}

// ------------------------------------------------------------------ the solver
struct NsBest {
    double v;
    long long e;
};

__device__ __forceinline__ void ns_better(double &v, long long &e, double v2, long long e2) {
    if (v2 > v || (v2 == v && e2 >= 0 && (e < 0 || e2 < e))) {
        v = v2;
        e = e2;
    }
}

// path lists in LDS (a tree path longer than this lives in the global lists only)
constexpr int NS_PCAP = 1024;
constexpr int NS_BITW = 8192; // 32-bit words of the position bitmap that LDS has room for (V <= 262144)
// SMALL: the tree arrays (16 B per node) and the potentials (8 B) live in LDS for the whole solve
constexpr int NS_SMALL_V = 4480;

struct NsLists {
    int node[2][NS_PCAP], arc[2][NS_PCAP], pos[2][NS_PCAP], size[2][NS_PCAP];
    int8_t dec[2][NS_PCAP];
    int snode[NS_PCAP], sarc[NS_PCAP], spos[NS_PCAP], ssize[NS_PCAP];
};

template <bool SMALL>
__global__ __launch_bounds__(NS_T) void k_ns_solve(NsProblem P, long long max_iters, double opt_tol, double feas_tol,
                                                   int block_k, int use_bitmap) {
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    __shared__ NsLists L;
    __shared__ double s_v[NS_W];
    __shared__ long long s_e[NS_W];
    __shared__ double s_red[NS_W];
    __shared__ int s_flag;
    __shared__ int s_count;
    __shared__ int s_k[2];
    __shared__ int s_scan[NS_W];
    __shared__ double s_bd[2][NS_W];
    __shared__ int s_bsz[2][NS_W], s_bslot[2][NS_W];
    __shared__ int s_in[4];     // entering arc: tail, head, state
    __shared__ double s_inrc;   // its reduced cost
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = P.V;
    const long long E = P.E;
    NsShared *sh = P.sh;
    int4 *nd;
    double *y;
    size_t dyn_off = 0;
    if constexpr (SMALL) {
        nd = reinterpret_cast<int4 *>(dyn_lds);
        y = reinterpret_cast<double *>(dyn_lds + sizeof(int4) * static_cast<size_t>(V));
        dyn_off = (sizeof(int4) + sizeof(double)) * static_cast<size_t>(V);
    } else {
        nd = P.nd;
        y = P.y;
    }
    // bitmap over the preorder positions + prefix of its words' popcounts (ranks of the path nodes)
    const int nwords = (V + 31) >> 5;
    const int wpl = (nwords + NS_T - 1) / NS_T;
    // (pointers into LDS are never tested against null: the flag says whether the room was allocated)
    typedef __attribute__((address_space(3))) unsigned int lds_u32;
    typedef __attribute__((address_space(3))) int lds_i32;
    lds_u32 *bm = (lds_u32 *)(dyn_lds + dyn_off);
    lds_i32 *wpre = (lds_i32 *)(dyn_lds + dyn_off + sizeof(unsigned int) * static_cast<size_t>(use_bitmap ? nwords : 0));
    volatile int4 *ndv = nd;

    // ================================================================= set-up
    const int root = sh->root;
    for (int v = tid; v < V; v += NS_T) {
        nd[v] = make_int4(v == root ? -1 : UNK, -1, 0, 1);
        P.first_child[v] = -1;
        P.next_sib[v] = -1;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    // tree arcs -> list (order irrelevant), in tmp
    for (long long e = tid; e < E; e += NS_T)
        if (P.state[e] == ST_TREE) {
            const int slot = __hip_atomic_fetch_add(&s_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot < V) P.tmp[slot] = static_cast<int32_t>(e);
        }
    __syncthreads();
    const int ntree = s_count;
    // parents by hooking sweeps from the root: an arc hooks its unknown end under its known end
    for (int sweep = 0; sweep <= V; ++sweep) {
        __syncthreads();
        if (tid == 0) s_flag = 0;
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < ntree; i += NS_T) {
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
    // every node reached?  (V - 1 arcs that reach all nodes are a spanning tree)
    {
        int bad = 0;
        for (int v = tid; v < V; v += NS_T)
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
    // children lists, ascending node order, then one depth-first pass (lane 0): preorder positions, sizes,
    // potentials (root 0), subtree sums of b_eff -> tree flows
    if (tid == 0) {
        for (int v = V - 1; v >= 0; --v) {
            if (v == root) continue;
            const int p = nd[v].x;
            P.next_sib[v] = P.first_child[p];
            P.first_child[p] = v;
        }
        int t = 0, v = root, infeasible = 0;
        double worst = 0.0;
        y[root] = 0.0;
        nd[root].z = 0;
        P.order[0] = root;
        P.exc[root] = P.beff[root];
        t = 1;
        bool down = true;
        for (long long steps = 0; steps < 4ll * V + 8; ++steps) {
            if (down) {
                const int c = P.first_child[v];
                if (c >= 0) { // enter the first child
                    const int a = nd[c].y;
                    y[c] = (P.tail[a] == c) ? P.cost[a] + y[v] : y[v] - P.cost[a];
                    nd[c].z = t;
                    P.order[t] = c;
                    P.exc[c] = P.beff[c];
                    ++t;
                    v = c;
                    continue;
                }
                down = false;
            }
            // v is finished
            nd[v].w = t - nd[v].z;
            if (v == root) break;
            const int a = nd[v].y, par = nd[v].x;
            const double ex = P.exc[v];
            double f = (P.tail[a] == v) ? ex : -ex;
            const double cp = P.cap[a];
            const double viol = f < 0.0 ? -f : (f > cp ? f - cp : 0.0);
            if (viol > worst) worst = viol;
            if (viol > feas_tol) infeasible = 1;
            f = f < 0.0 ? 0.0 : (f > cp ? cp : f);
            P.flow[a] = f;
            P.exc[par] = P.exc[par] + ex;
            const int sb = P.next_sib[v];
            if (sb >= 0) { // enter the next sibling
                const int a2 = nd[sb].y;
                y[sb] = (P.tail[a2] == sb) ? P.cost[a2] + y[par] : y[par] - P.cost[a2];
                nd[sb].z = t;
                P.order[t] = sb;
                P.exc[sb] = P.beff[sb];
                ++t;
                v = sb;
                down = true;
            } else {
                v = par;
            }
        }
        // the root's own excess must vanish (sum of b = 0 up to rounding)
        const double rex = fabs(P.exc[root]);
        if (rex > worst) worst = rex;
        if (rex > feas_tol * (1.0 + fabs(P.beff[root]))) infeasible = 1;
        sh->max_violation = worst;
        s_flag = infeasible;
    }
    __syncthreads();
    if (s_flag) {
        if (tid == 0) sh->status = 5;
        return;
    }

    // ================================================================= pivots
    const long long B = static_cast<long long>(NS_T) * block_k;
    long long cursor = 0, scanned = 0, iters = 0;
    long long status = -1;
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nblocks = 0, path_sum = 0, range_sum = 0;
    long long tc = clock64();
    auto tick = [&](int k) {
        const long long now = clock64();
        tph[k] += now - tc;
        tc = now;
    };
    while (status < 0) {
        ++nblocks;
        // ---- price one block; the lane that holds the winner publishes the arc's data
        double bv = 0.0, my_rc = 0.0;
        long long be = -1;
        int my_p = 0, my_q = 0, my_st = 0;
        for (int k = 0; k < block_k; ++k) {
            long long e = cursor + tid + static_cast<long long>(k) * NS_T;
            if (e >= E) e -= E;
            if (e < E && static_cast<long long>(tid) + static_cast<long long>(k) * NS_T < E) {
                const int st = P.state[e];
                const int p = P.tail[e], q = P.head[e]; // requested with the state, not after it
                const double ce = P.cost[e];
                if (st != ST_TREE) {
                    const double rc = (ce - y[p]) + y[q];
                    const double viol = st == ST_LOWER ? -rc : rc;
                    if (viol > opt_tol && p != q && (viol > bv || (viol == bv && (be < 0 || e < be)))) {
                        bv = viol;
                        be = e;
                        my_p = p;
                        my_q = q;
                        my_st = st;
                        my_rc = rc;
                    }
                }
            }
        }
        const long long my_e = be;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_down(bv, o, 64);
            const long long e2 = __shfl_down(be, o, 64);
            ns_better(bv, be, v2, e2);
        }
        __syncthreads(); // s_v / s_e free again
        if (lane == 0) {
            s_v[wave] = bv;
            s_e[wave] = be;
        }
        __syncthreads();
        bv = s_v[0];
        be = s_e[0];
#pragma unroll
        for (int w = 1; w < NS_W; ++w) ns_better(bv, be, s_v[w], s_e[w]);
        cursor += B;
        if (cursor >= E) cursor -= E * (cursor / E);
        if (be < 0) {
            scanned += B;
            if (scanned >= E + B) status = 0; // a whole round without a candidate
            continue;
        }
        scanned = 0;
        if (iters >= max_iters) {
            status = 3;
            break;
        }
        ++iters;
        if (my_e == be) {
            s_in[0] = my_p;
            s_in[1] = my_q;
            s_in[2] = my_st;
            s_inrc = my_rc;
        }
        if (tid < 2) s_k[tid] = 0;
        __syncthreads();
        tick(0);

        // ---- cycle: every lane tests its nodes against the two ends of the entering arc.  In preorder "v is an
        // ancestor of w" is pos[v] <= pos[w] < pos[v] + size[v], so the two tree paths to the join are the nodes
        // that are an ancestor of exactly one end -- found by all lanes at once, no climb.  Sizes grow strictly
        // along a path, so "first met from the end" is "smallest size".
        const long long ein = be;
        const int p_in = s_in[0], q_in = s_in[1], st_in = s_in[2];
        const double rc_in = s_inrc;
        const int first = st_in == ST_LOWER ? p_in : q_in; // the end that receives flow
        const int second = st_in == ST_LOWER ? q_in : p_in;
        const int pf = nd[first].z, ps2 = nd[second].z;
        const double cap_in = P.cap[ein];
        double bd[2] = {INFINITY, INFINITY};
        int bsz[2] = {0x7fffffff, -1}, bslot[2] = {-1, -1};
        for (int base = 0; base < V; base += NS_T) { // uniform trip count: the slots come from wave ballots
            const int v = base + tid;
            const int4 r = nd[v < V ? v : V - 1]; // lanes past the end re-read the last node and stay out
            const bool a1 = r.z <= pf && pf < r.z + r.w, a2 = r.z <= ps2 && ps2 < r.z + r.w;
            const bool on = v < V && a1 != a2;
            const int side = a1 ? 0 : 1;
            const unsigned long long m0 = __ballot(on && side == 0), m1 = __ballot(on && side == 1);
            int base0 = 0, base1 = 0;
            int cnt0 = __popcll(m0), cnt1 = __popcll(m1);
            asm volatile("" : "+v"(cnt0), "+v"(cnt1)); // the counts are wave-uniform: keep them in VGPRs for ds_add
            if (lane == 0) { // one LDS atomic per wave and side instead of one per path node
                if (cnt0) base0 = __hip_atomic_fetch_add(&s_k[0], cnt0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cnt1) base1 = __hip_atomic_fetch_add(&s_k[1], cnt1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            base0 = __builtin_amdgcn_readfirstlane(base0);
            base1 = __builtin_amdgcn_readfirstlane(base1);
            if (on) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const int slot = side == 0 ? base0 + __popcll(m0 & below) : base1 + __popcll(m1 & below);
                if (slot < NS_PCAP) { // global memory takes only what LDS has no room for
                    L.node[side][slot] = v;
                    L.arc[side][slot] = r.y;
                    L.pos[side][slot] = r.z;
                    L.size[side][slot] = r.w;
                } else {
                    P.pnode[side][slot] = v;
                    P.parc[side][slot] = r.y;
                    P.ppos[side][slot] = r.z;
                    P.psize[side][slot] = r.w;
                }
            }
        }
        __syncthreads();
        // the arcs of the two paths: residual in the direction of the cycle, ratio test -- one entry per lane, so
        // the arc data of the whole cycle arrive in one round trip
        {
            const int k0 = s_k[0], k1 = s_k[1];
            for (int i = tid; i < k0 + k1; i += NS_T) {
                const int side = i < k0 ? 0 : 1, j = i < k0 ? i : i - k0;
                const bool in_lds = j < NS_PCAP;
                const int v = in_lds ? L.node[side][j] : P.pnode[side][j];
                const int a = in_lds ? L.arc[side][j] : P.parc[side][j];
                const int sz = in_lds ? L.size[side][j] : P.psize[side][j];
                const bool dec = (P.tail[a] == v) == (side == 0);
                const double f = P.flow[a];
                const double d = dec ? f : P.cap[a] - f;
                if (in_lds) L.dec[side][j] = dec ? 1 : 0;
                else P.pdec[side][j] = dec ? 1 : 0;
                // ratio test: strict on the first side (first blocking arc from the end), the last one on the second
                if (side == 0 ? (d < bd[0] || (d == bd[0] && sz < bsz[0])) : (d < bd[1] || (d == bd[1] && sz > bsz[1]))) {
                    bd[side] = d;
                    bsz[side] = sz;
                    bslot[side] = j;
                }
            }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const double d2 = __shfl_down(bd[sd], o2, 64);
                const int z2 = __shfl_down(bsz[sd], o2, 64), l2 = __shfl_down(bslot[sd], o2, 64);
                if (l2 >= 0 && (bslot[sd] < 0 || d2 < bd[sd] || (d2 == bd[sd] && (sd == 0 ? z2 < bsz[sd] : z2 > bsz[sd])))) {
                    bd[sd] = d2;
                    bsz[sd] = z2;
                    bslot[sd] = l2;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                s_bd[sd][wave] = bd[sd];
                s_bsz[sd][wave] = bsz[sd];
                s_bslot[sd][wave] = bslot[sd];
            }
        }
        __syncthreads();
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            bd[sd] = s_bd[sd][0];
            bsz[sd] = s_bsz[sd][0];
            bslot[sd] = s_bslot[sd][0];
            for (int w = 1; w < NS_W; ++w) {
                const double d2 = s_bd[sd][w];
                const int z2 = s_bsz[sd][w], l2 = s_bslot[sd][w];
                if (l2 >= 0 && (bslot[sd] < 0 || d2 < bd[sd] || (d2 == bd[sd] && (sd == 0 ? z2 < bsz[sd] : z2 > bsz[sd])))) {
                    bd[sd] = d2;
                    bsz[sd] = z2;
                    bslot[sd] = l2;
                }
            }
        }
        tick(1);
        // ---- ratio test
        double delta = cap_in;
        int result = 0;
        if (bslot[0] >= 0 && bd[0] < delta) {
            delta = bd[0];
            result = 1;
        }
        if (bslot[1] >= 0 && bd[1] <= delta) {
            delta = bd[1];
            result = 2;
        }
        if (isinf(delta)) {
            status = 2;
            break;
        }
        // ---- augment
        const int k0 = s_k[0], k1 = s_k[1];
        path_sum += k0 + k1;
        auto e_node = [&](int side, int j) { return j < NS_PCAP ? L.node[side][j] : P.pnode[side][j]; };
        auto e_arc = [&](int side, int j) { return j < NS_PCAP ? L.arc[side][j] : P.parc[side][j]; };
        auto e_pos = [&](int side, int j) { return j < NS_PCAP ? L.pos[side][j] : P.ppos[side][j]; };
        auto e_size = [&](int side, int j) { return j < NS_PCAP ? L.size[side][j] : P.psize[side][j]; };
        auto e_dec = [&](int side, int j) { return j < NS_PCAP ? L.dec[side][j] : P.pdec[side][j]; };
        const int s = result - 1, o = 1 - s; // (s = -1: the entering arc itself blocks)
        const int out_slot = result ? bslot[s] : -1;
        for (int i = tid; i < k0 + k1; i += NS_T) {
            const int side = i < k0 ? 0 : 1, j = i < k0 ? i : i - k0;
            const int a = e_arc(side, j);
            const bool dec = e_dec(side, j) != 0;
            if (side == s && j == out_slot) { // the leaving arc lands exactly on its bound
                P.flow[a] = dec ? 0.0 : P.cap[a];
                P.state[a] = static_cast<int8_t>(dec ? ST_LOWER : ST_UPPER);
            } else if (delta > 0.0) {
                P.flow[a] = dec ? P.flow[a] - delta : P.flow[a] + delta;
            }
        }
        if (result == 0) { // the entering arc runs to its other bound: no change of the tree
            if (tid == 0) {
                P.state[ein] = static_cast<int8_t>(-st_in);
                P.flow[ein] = st_in == ST_LOWER ? cap_in : 0.0;
            }
            __syncthreads();
            continue;
        }
        tick(2);
        const int ks = s_k[s], ko = s_k[o];
        const int u_in = s == 0 ? first : second, v_in = s == 0 ? second : first;
        const int n_sub = bsz[s], a_pos = e_pos(s, out_slot);
        const int b_pos = s == 0 ? ps2 : pf;
        const double dy = (p_in == u_in) ? rc_in : -rc_in;
        if (tid == 0) {
            P.state[ein] = ST_TREE;
            P.flow[ein] = st_in == ST_LOWER ? delta : cap_in - delta;
            s_count = 0;
        }
        __syncthreads();
        auto put_sorted = [&](int rank, int node, int arc, int ps, int sz) {
            if (rank < NS_PCAP) {
                L.snode[rank] = node;
                L.sarc[rank] = arc;
                L.spos[rank] = ps;
                L.ssize[rank] = sz;
            } else {
                P.snode[rank] = node;
                P.sarc[rank] = arc;
                P.spos[rank] = ps;
                P.ssize[rank] = sz;
            }
        };
        auto s_node = [&](int i) { return i < NS_PCAP ? L.snode[i] : P.snode[i]; };
        auto s_arc = [&](int i) { return i < NS_PCAP ? L.sarc[i] : P.sarc[i]; };
        auto s_pos = [&](int i) { return i < NS_PCAP ? L.spos[i] : P.spos[i]; };
        auto s_size = [&](int i) { return i < NS_PCAP ? L.ssize[i] : P.ssize[i]; };
        // ---- the path u_in .. u_out in order (the rest of side s, above u_out, loses the subtree; the other side
        // gains it).  Order = by position: a bitmap over the preorder positions takes the path's nodes, popcounts
        // give every node its rank (without the bitmap, V > 32 * NS_BITW: by counting, k^2 / lanes)
        if (use_bitmap) {
            for (int w = tid; w < nwords; w += NS_T) bm[w] = 0u;
            __syncthreads();
        }
        for (int i = tid; i < ks + ko; i += NS_T) {
            if (i < ks) {
                const int sz = e_size(s, i);
                if (sz > n_sub) {
                    nd[e_node(s, i)].w = sz - n_sub;
                } else if (use_bitmap) {
                    const int ps = e_pos(s, i);
                    __atomic_fetch_or(&bm[ps >> 5], 1u << (ps & 31), __ATOMIC_RELAXED);
                } else {
                    int rank = 0;
                    for (int j = 0; j < ks; ++j) rank += e_size(s, j) < sz ? 1 : 0;
                    put_sorted(rank, e_node(s, i), e_arc(s, i), e_pos(s, i), sz);
                    __hip_atomic_fetch_add(&s_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                const int j = i - ks;
                nd[e_node(o, j)].w = e_size(o, j) + n_sub;
            }
        }
        __syncthreads();
        if (use_bitmap) {
            // exclusive prefix of the words' popcounts (lane t owns words [t * wpl, (t + 1) * wpl))
            int local = 0;
            for (int q = 0; q < wpl; ++q) {
                const int w = tid * wpl + q;
                if (w < nwords) local += __popc(bm[w]);
            }
            int incl = local;
#pragma unroll
            for (int d2 = 1; d2 < 64; d2 <<= 1) {
                const int up = __shfl_up(incl, d2, 64);
                if (lane >= d2) incl += up;
            }
            if (lane == 63) s_scan[wave] = incl;
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < NS_W; ++w) {
                before += w < wave ? s_scan[w] : 0;
                total += s_scan[w];
            }
            int run = before + incl - local;
            for (int q = 0; q < wpl; ++q) {
                const int w = tid * wpl + q;
                if (w < nwords) {
                    wpre[w] = run;
                    run += __popc(bm[w]);
                }
            }
            if (tid == 0) s_count = total;
            __syncthreads();
            for (int i = tid; i < ks; i += NS_T) {
                const int sz = e_size(s, i);
                if (sz <= n_sub) {
                    const int ps = e_pos(s, i);
                    const int above = wpre[ps >> 5] + __popc(bm[ps >> 5] & ((1u << (ps & 31)) - 1u));
                    const int rank = total - 1 - above; // deepest node (u_in) first
                    put_sorted(rank, e_node(s, i), e_arc(s, i), ps, sz);
                }
            }
            __syncthreads();
        }
        tick(3);
        const int idx = s_count - 1; // position of u_out on the path
        // the path is re-rooted: every node hangs under its former child
        for (int i = tid; i <= idx; i += NS_T) {
            const int w = s_node(i);
            nd[w].x = i == 0 ? v_in : s_node(i - 1);
            nd[w].y = i == 0 ? static_cast<int>(ein) : s_arc(i - 1);
            nd[w].w = i == 0 ? n_sub : n_sub - s_size(i - 1);
        }
        // ---- preorder: S (old [a_pos, a_pos + n_sub)) moves right behind v_in, re-rooted at u_in
        const int lo = a_pos < b_pos + 1 ? a_pos : b_pos + 1;
        const int hi = a_pos + n_sub > b_pos + 1 ? a_pos + n_sub : b_pos + 1;
        const int newstart = b_pos < a_pos ? b_pos + 1 : b_pos + 1 - n_sub;
        range_sum += hi - lo;
        for (int t = lo + tid; t < hi; t += NS_T) {
            const int w = P.order[t];
            int nt;
            if (t >= a_pos && t < a_pos + n_sub) {
                int l2 = 0, h2 = idx; // smallest i with t inside the old segment of path node i
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
                    const int hp = s_pos(i - 1), hs = s_size(i - 1); // the hole: the old segment of path node i-1
                    off = hs;
                    rel = t < hp ? t - s_pos(i) : (hp - s_pos(i)) + (t - (hp + hs));
                }
                nt = newstart + off + rel;
                y[w] = y[w] + dy;
            } else {
                nt = b_pos < a_pos ? t + n_sub : t - n_sub;
            }
            P.tmp[nt - lo] = w;
        }
        __syncthreads();
        tick(4);
        for (int t = lo + tid; t < hi; t += NS_T) {
            const int w = P.tmp[t - lo];
            P.order[t] = w;
            nd[w].z = t;
        }
        __syncthreads();
        tick(5);
    }

    // ================================================================= results
    // potentials once more from the tree itself (pointer jumping), so that the duals handed back are the
    // tree's own and not the sum of a million shifts
    {
        for (int v = tid; v < V; v += NS_T) {
            const int4 r = nd[v];
            double c = 0.0;
            int up = v;
            if (r.x >= 0) {
                c = (P.tail[r.y] == v) ? P.cost[r.y] : -P.cost[r.y];
                up = r.x;
            }
            P.acc[0][v] = c;
            P.anc[0][v] = up;
        }
        __syncthreads();
        int cur = 0;
        for (int round = 0; round < 32; ++round) { // after round r: acc = the sum over 2^r arcs, anc = that ancestor
            if (tid == 0) s_flag = 0;
            __syncthreads();
            int moved = 0;
            for (int v = tid; v < V; v += NS_T) {
                const int a = P.anc[cur][v];
                const int a2 = P.anc[cur][a];
                P.acc[1 - cur][v] = P.acc[cur][v] + P.acc[cur][a]; // the root carries 0 and points to itself
                P.anc[1 - cur][v] = a2;
                if (a2 != a) moved = 1;
            }
            if (moved) s_flag = 1;
            __syncthreads();
            cur = 1 - cur;
            if (!s_flag) break;
        }
        for (int v = tid; v < V; v += NS_T) P.y[v] = P.acc[cur][v];
        __syncthreads();
    }
    // objective: fixed-order sum
    double part = 0.0;
    for (long long e = tid; e < E; e += NS_T) part += P.cost[e] * P.flow[e];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) part += __shfl_down(part, o2, 64);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < NS_W; ++w) tot += s_red[w];
        sh->obj = tot;
        sh->iters = iters;
        sh->status = status;
        for (int k = 0; k < 8; ++k) sh->t_phase[k] = tph[k];
        sh->blocks = nblocks;
        sh->path_sum = path_sum;
        sh->range_sum = range_sum;
    }
}

__global__ __launch_bounds__(256) void k_ns_outputs(int64_t V, int64_t E, const int8_t *__restrict__ state, int root,
                                                    int8_t *__restrict__ vbasis, int8_t *__restrict__ cbasis) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (vbasis && i < E) vbasis[i] = static_cast<int8_t>(state[i] == ST_TREE ? 0 : state[i] == ST_LOWER ? -1 : -2);
    if (cbasis && i < V) cbasis[i] = static_cast<int8_t>(i == root ? 0 : -1);
}

} // namespace

SX_API int sx_netsimplex_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                             const double *u, const int8_t *vbasis_in, const int8_t *cbasis_in, int64_t max_iter,
                             double feas_tol, double opt_tol, double *x_out, double *y_out, int8_t *vbasis_out,
                             int8_t *cbasis_out, sx_simplex_result *result) {
    SX_ENTER(ctx);
    SX_REQUIRE(A && b && c && l && u && vbasis_in && cbasis_in && result, "NULL argument");
    SX_REQUIRE(A->ctx->device == ctx->device, "matrix lives on another device");
    SX_REQUIRE(A->csc_ptr && A->csr_ptr, "matrix needs both layouts");
    memset(result, 0, sizeof(*result));
    result->status = 5;
    const int64_t V = A->m, E = A->n;
    if (V < 2 || E < 1 || V >= (static_cast<int64_t>(1) << 30) || E >= (static_cast<int64_t>(1) << 31) || A->nnz != 2 * E)
        return SX_OK; // not a network: the caller takes the general simplex
    if (ctx->opt_netsimplex == 0) return SX_OK;
    hipStream_t s = ctx->stream;
    // every temporary of the call in one block of the context (about 30 B per arc and 130 B per node)
    SX_TRY(sx_reserve3(ctx, static_cast<size_t>(40) * static_cast<size_t>(E) + static_cast<size_t>(192) * static_cast<size_t>(V) + (1u << 20)));
    sx_arena pool(ctx);
    NsProblem P;
    memset(&P, 0, sizeof(P));
    P.V = static_cast<int>(V);
    P.E = E;
    double *xn, *beff, *flow;
    int8_t *state;
    SX_TRY(pool.get(E, &P.tail));
    SX_TRY(pool.get(E, &P.head));
    SX_TRY(pool.get(E, &state));
    SX_TRY(pool.get(E, &xn));
    SX_TRY(pool.get(V, &beff));
    SX_TRY(pool.get(1, &P.sh));
    SX_HIP(hipMemsetAsync(P.sh, 0, sizeof(NsShared), s));
    hipLaunchKernelGGL(k_ns_endpoints, dim3(static_cast<unsigned>((E + 255) / 256)), dim3(256), 0, s, E, A->csc_ptr,
                       A->csc_idx, A->csc_val, l, u, vbasis_in, P.tail, P.head, state, xn, P.sh);
    hipLaunchKernelGGL(k_ns_root, dim3(static_cast<unsigned>((V + 255) / 256)), dim3(256), 0, s, V, cbasis_in, P.sh);
    NsShared sh;
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (sh.not_network || sh.ntree != V - 1 || sh.nroot != 1) return SX_OK;
    // b_eff = b - A x_N by the row walk (sums in stored order: deterministic)
    SX_TRY(sx_score_rows_dev(ctx, A, xn, b, nullptr, 0.0, beff, nullptr));
    flow = x_out ? x_out : xn; // the nonbasic flows are already in place when the flows live in xn
    if (x_out) SX_HIP(hipMemcpyAsync(x_out, xn, sizeof(double) * static_cast<size_t>(E), hipMemcpyDeviceToDevice, s));
    P.cost = c;
    P.cap = u;
    P.flow = flow;
    P.state = state;
    P.beff = beff;
    SX_TRY(pool.get(V, &P.nd));
    double *y = y_out;
    if (!y) SX_TRY(pool.get(V, &y));
    P.y = y;
    SX_TRY(pool.get(V, &P.order));
    SX_TRY(pool.get(V, &P.tmp));
    SX_TRY(pool.get(V, &P.first_child));
    SX_TRY(pool.get(V, &P.next_sib));
    SX_TRY(pool.get(V, &P.exc));
    SX_TRY(pool.get(V, &P.snode));
    SX_TRY(pool.get(V, &P.sarc));
    SX_TRY(pool.get(V, &P.spos));
    SX_TRY(pool.get(V, &P.ssize));
    for (int k = 0; k < 2; ++k) {
        SX_TRY(pool.get(V, &P.pnode[k]));
        SX_TRY(pool.get(V, &P.parc[k]));
        SX_TRY(pool.get(V, &P.ppos[k]));
        SX_TRY(pool.get(V, &P.psize[k]));
        SX_TRY(pool.get(V, &P.pdec[k]));
        SX_TRY(pool.get(V, &P.acc[k]));
        SX_TRY(pool.get(V, &P.anc[k]));
    }
    // arcs priced per block: all lanes, 1..8 arcs each (pricing is parallel, so blocks are larger than a
    // sequential code would pick: better entering arcs for the same latency)
    int block_k = static_cast<int>(E / (static_cast<int64_t>(NS_T) * 32));
    block_k = block_k < 1 ? 1 : block_k > 8 ? 8 : block_k;
    if (ctx->opt_ns_block > 0) block_k = ctx->opt_ns_block > 64 ? 64 : ctx->opt_ns_block;
    const long long limit = max_iter > 0 ? max_iter : 100 * (V + E);
    const int64_t nwords = (V + 31) / 32;
    const int use_bitmap = nwords <= NS_BITW ? 1 : 0;
    const size_t bm_bytes = use_bitmap ? static_cast<size_t>(nwords) * 8 : 0;
    if (V <= NS_SMALL_V && ctx->opt_ns_lds) { // tree and potentials in LDS
        const size_t dyn = (sizeof(int4) + sizeof(double)) * static_cast<size_t>(V) + bm_bytes;
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ns_solve<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn)));
        hipLaunchKernelGGL(k_ns_solve<true>, dim3(1), dim3(NS_T), dyn, s, P, limit, opt_tol, feas_tol, block_k, use_bitmap);
    } else {
        SX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ns_solve<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bm_bytes)));
        hipLaunchKernelGGL(k_ns_solve<false>, dim3(1), dim3(NS_T), bm_bytes, s, P, limit, opt_tol, feas_tol, block_k,
                           use_bitmap);
    }
    hipLaunchKernelGGL(k_ns_outputs, dim3(static_cast<unsigned>(((E > V ? E : V) + 255) / 256)), dim3(256), 0, s, V, E,
                       state, sh.root, vbasis_out, cbasis_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    if (getenv("SX_NS_PROFILE"))
        fprintf(stderr, "[sx_netsimplex] V=%lld E=%lld pivots=%lld blocks=%lld clocks: price %lld mark %lld augment %lld rank %lld "
                        "range1 %lld range2 %lld; cycle nodes %lld, positions moved %lld\n", (long long)V, (long long)E, sh.iters, sh.blocks,
                sh.t_phase[0], sh.t_phase[1], sh.t_phase[2], sh.t_phase[3], sh.t_phase[4], sh.t_phase[5], sh.path_sum, sh.range_sum);
    result->status = sh.status;
    result->iters = sh.iters;
    result->phase1_iters = 0;
    result->warm_start_used = 1;
    result->obj = sh.obj;
    result->max_violation = sh.max_violation;
    return SX_OK;
}
