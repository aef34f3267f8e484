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
//   cycle    two lanes, one per end of the entering arc, climb to the join node and record their paths;
//            the join is found without depths: the tree is kept in PREORDER (pos[v], size[v], order[]), so
//            "u is an ancestor of w" is pos[u] <= pos[w] < pos[u] + size[u];
//   ratio    strict '<' on the side that loses flow first, '<=' on the other (the last blocking arc seen
//            from the join leaves: strongly feasible trees, no cycling from a strongly feasible start);
//   augment  all lanes over the two recorded paths;
//   re-hang  the cut subtree S is re-rooted at the entering arc's end: its new preorder is the old one cut
//            into 2k+1 pieces (k = path length) whose offsets telescope to old sizes, so every lane moves
//            its elements of the affected range [lo, hi) of order[] independently (binary search over the
//            path), shifts the potentials of S by the entering arc's reduced cost and rewrites pos[].
// No thread/linked-list traversal anywhere: the sequential part of a pivot is the two climbs.
//
// Start: the given basis must be a spanning tree (vbasis 0 on V-1 arcs, cbasis 0 on the root row) whose tree
// flows respect the bounds; anything else returns status 5 and the caller takes the general simplex
// (sx_simplex.hip, which has a phase 1).  Set-up (parents by hooking sweeps, then one depth-first pass for
// preorder, sizes, potentials and tree flows) runs in the same kernel; children are visited in ascending node
// order, so the whole solve is deterministic.
#include "sx_internal.h"

#include <cmath>

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
    int32_t *pnode[2], *parc[2], *ppos[2], *psize[2]; // [V] each: recorded paths of the two climbs
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

__global__ __launch_bounds__(NS_T) void k_ns_solve(NsProblem P, long long max_iters, double opt_tol, double feas_tol,
                                                   int block_k) {
    __shared__ double s_v[NS_W];
    __shared__ long long s_e[NS_W];
    __shared__ double s_red[NS_W];
    __shared__ int s_flag;
    __shared__ int s_count;
    __shared__ int s_k[2], s_arg[2], s_join[2];
    __shared__ double s_delta[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = P.V;
    const long long E = P.E;
    NsShared *sh = P.sh;
    volatile int4 *ndv = P.nd;
    int4 *nd = P.nd;

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
            const int slot = atomicAdd(&s_count, 1);
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
        P.y[root] = 0.0;
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
                    P.y[c] = (P.tail[a] == c) ? P.cost[a] + P.y[v] : P.y[v] - P.cost[a];
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
            const int s = P.next_sib[v];
            if (s >= 0) { // enter the next sibling
                const int a2 = nd[s].y;
                P.y[s] = (P.tail[a2] == s) ? P.cost[a2] + P.y[par] : P.y[par] - P.cost[a2];
                nd[s].z = t;
                P.order[t] = s;
                P.exc[s] = P.beff[s];
                ++t;
                v = s;
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
    while (status < 0) {
        // ---- price one block
        double bv = 0.0;
        long long be = -1;
        for (int k = 0; k < block_k; ++k) {
            long long e = cursor + tid + static_cast<long long>(k) * NS_T;
            if (e >= E) e -= E;
            if (e < E && static_cast<long long>(tid) + static_cast<long long>(k) * NS_T < E) {
                const int st = P.state[e];
                if (st != ST_TREE) {
                    const int p = P.tail[e], q = P.head[e];
                    const double rc = (P.cost[e] - P.y[p]) + P.y[q];
                    const double viol = st == ST_LOWER ? -rc : rc;
                    if (viol > opt_tol && p != q) ns_better(bv, be, viol, e);
                }
            }
        }
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

        // ---- cycle: climb from both ends of the entering arc to the join
        const long long ein = be;
        const int p_in = P.tail[ein], q_in = P.head[ein];
        const int st_in = P.state[ein];
        const double rc_in = (P.cost[ein] - P.y[p_in]) + P.y[q_in];
        const int first = st_in == ST_LOWER ? p_in : q_in; // the end that receives flow
        const int second = st_in == ST_LOWER ? q_in : p_in;
        if (lane == 0 && wave < 2) {
            const int side = wave;
            const int start = side == 0 ? first : second, other = side == 0 ? second : first;
            const int po = nd[other].z;
            int k = 0, u = start, arg = -1;
            double delta = INFINITY;
            for (;;) {
                const int4 r = nd[u];
                if (r.z <= po && po < r.z + r.w) break; // u is an ancestor of (or is) the other end: the join
                if (k >= V || r.x < 0) { // cannot happen on a tree: leave instead of spinning
                    k = -1;
                    break;
                }
                const int a = r.y;
                const bool dec = (P.tail[a] == u) == (side == 0);
                const double f = P.flow[a];
                const double d = dec ? f : P.cap[a] - f;
                P.pnode[side][k] = u;
                P.parc[side][k] = a;
                P.pdec[side][k] = dec ? 1 : 0;
                P.ppos[side][k] = r.z;
                P.psize[side][k] = r.w;
                if (side == 0 ? (d < delta) : (d <= delta)) {
                    delta = d;
                    arg = k;
                }
                ++k;
                u = r.x;
            }
            s_k[side] = k;
            s_arg[side] = arg;
            s_delta[side] = delta;
            s_join[side] = u;
        }
        __syncthreads();
        if (s_k[0] < 0 || s_k[1] < 0 || s_join[0] != s_join[1]) {
            status = 4;
            break;
        }
        // ---- ratio test
        double delta = P.cap[ein];
        int result = 0;
        if (s_delta[0] < delta) {
            delta = s_delta[0];
            result = 1;
        }
        if (s_delta[1] <= delta) {
            delta = s_delta[1];
            result = 2;
        }
        if (isinf(delta)) {
            status = 2;
            break;
        }
        // ---- augment
        const int k0 = s_k[0], k1 = s_k[1];
        if (delta > 0.0) {
            for (int i = tid; i < k0 + k1; i += NS_T) {
                const int side = i < k0 ? 0 : 1, j = i < k0 ? i : i - k0;
                const int a = P.parc[side][j];
                P.flow[a] = P.pdec[side][j] ? P.flow[a] - delta : P.flow[a] + delta;
            }
        }
        if (result == 0) { // the entering arc runs to its other bound: no change of the tree
            if (tid == 0) {
                P.state[ein] = static_cast<int8_t>(-st_in);
                P.flow[ein] = st_in == ST_LOWER ? P.cap[ein] : 0.0;
            }
            __syncthreads();
            continue;
        }
        const int s = result - 1, o = 1 - s;
        const int idx = s_arg[s];
        const int ks = s_k[s], ko = s_k[o];
        const int u_in = s == 0 ? first : second, v_in = s == 0 ? second : first;
        const int n_sub = P.psize[s][idx], a_pos = P.ppos[s][idx];
        const int b_pos = nd[v_in].z;
        const double dy = (p_in == u_in) ? rc_in : -rc_in;
        __syncthreads(); // every lane has read the flows / records it needs
        if (tid == 0) {
            const int a_out = P.parc[s][idx];
            const bool dec = P.pdec[s][idx] != 0;
            P.state[a_out] = static_cast<int8_t>(dec ? ST_LOWER : ST_UPPER);
            P.flow[a_out] = dec ? 0.0 : P.cap[a_out];
            P.state[ein] = ST_TREE;
            P.flow[ein] = st_in == ST_LOWER ? delta : P.cap[ein] - delta;
        }
        // ---- sizes off the old branch, onto the new one; the path itself is re-rooted
        for (int i = tid; i < ks + ko; i += NS_T) {
            if (i < ks) {
                const int w = P.pnode[s][i];
                if (i > idx) {
                    nd[w].w = P.psize[s][i] - n_sub;
                } else {
                    nd[w].x = i == 0 ? v_in : P.pnode[s][i - 1];
                    nd[w].y = i == 0 ? static_cast<int>(ein) : P.parc[s][i - 1];
                    nd[w].w = i == 0 ? n_sub : n_sub - P.psize[s][i - 1];
                }
            } else {
                const int j = i - ks;
                nd[P.pnode[o][j]].w = P.psize[o][j] + n_sub;
            }
        }
        // ---- preorder: S (old [a_pos, a_pos + n_sub)) moves right behind v_in, re-rooted at u_in
        const int lo = a_pos < b_pos + 1 ? a_pos : b_pos + 1;
        const int hi = a_pos + n_sub > b_pos + 1 ? a_pos + n_sub : b_pos + 1;
        const int newstart = b_pos < a_pos ? b_pos + 1 : b_pos + 1 - n_sub;
        const int32_t *pp = P.ppos[s], *ps = P.psize[s];
        for (int t = lo + tid; t < hi; t += NS_T) {
            const int w = P.order[t];
            int nt;
            if (t >= a_pos && t < a_pos + n_sub) {
                int l2 = 0, h2 = idx; // smallest i with t inside the old segment of path node i
                while (l2 < h2) {
                    const int mid = (l2 + h2) >> 1;
                    const int q0 = pp[mid];
                    if (t >= q0 && t < q0 + ps[mid]) h2 = mid;
                    else l2 = mid + 1;
                }
                const int i = l2;
                int rel, off = 0;
                if (i == 0) {
                    rel = t - pp[0];
                } else {
                    const int hp = pp[i - 1], hs = ps[i - 1]; // the hole: the old segment of path node i-1
                    off = hs;
                    rel = t < hp ? t - pp[i] : (hp - pp[i]) + (t - (hp + hs));
                }
                nt = newstart + off + rel;
                P.y[w] = P.y[w] + dy;
            } else {
                nt = b_pos < a_pos ? t + n_sub : t - n_sub;
            }
            P.tmp[nt - lo] = w;
        }
        __syncthreads();
        for (int t = lo + tid; t < hi; t += NS_T) {
            const int w = P.tmp[t - lo];
            P.order[t] = w;
            nd[w].z = t;
        }
        __syncthreads();
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
    }
}

__global__ __launch_bounds__(256) void k_ns_outputs(int64_t V, int64_t E, const int8_t *__restrict__ state, int root,
                                                    int8_t *__restrict__ vbasis, int8_t *__restrict__ cbasis) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (vbasis && i < E) vbasis[i] = static_cast<int8_t>(state[i] == ST_TREE ? 0 : state[i] == ST_LOWER ? -1 : -2);
    if (cbasis && i < V) cbasis[i] = static_cast<int8_t>(i == root ? 0 : -1);
}

struct Pool { // device temporaries of one call
    std::vector<void *> p;
    ~Pool() {
        for (void *q : p) (void)hipFree(q);
    }
    template <class T>
    int get(size_t count, T **out) {
        void *d = nullptr;
        SX_HIP(hipMalloc(&d, sizeof(T) * (count ? count : 1)));
        p.push_back(d);
        *out = static_cast<T *>(d);
        return SX_OK;
    }
};

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
    hipStream_t s = ctx->stream;
    Pool pool;
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
    hipLaunchKernelGGL(k_ns_solve, dim3(1), dim3(NS_T), 0, s, P, limit, opt_tol, feas_tol, block_k);
    hipLaunchKernelGGL(k_ns_outputs, dim3(static_cast<unsigned>(((E > V ? E : V) + 255) / 256)), dim3(256), 0, s, V, E,
                       state, sh.root, vbasis_out, cbasis_out);
    SX_HIP(hipGetLastError());
    SX_HIP(hipMemcpyAsync(&sh, P.sh, sizeof(sh), hipMemcpyDeviceToHost, s));
    SX_HIP(hipStreamSynchronize(s));
    result->status = sh.status;
    result->iters = sh.iters;
    result->phase1_iters = 0;
    result->warm_start_used = 1;
    result->obj = sh.obj;
    result->max_violation = sh.max_violation;
    return SX_OK;
}
