/*
 * sxhip.h -- C ABI of libsxhip.so, the MI355X (gfx950) implementation of the
 * smart-crossover hot path: primal-dual indicator scoring of LP columns/rows,
 * cost perturbation, the projector CG that scales it, sub-problem compaction,
 * network flow indicators, ranking and reduced-cost pricing.
 *
 * The reference (wcwj0147/smart-crossover) is pure Python and has no FFI; the
 * entry points below are what a binding for its hot-path functions would call.
 * Each one names the reference lines it replaces (paths relative to
 * src/smart_crossover/).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every function returns SX_OK (0) or a negative SX_ERR_* code; the text
 *     of the last failure on the calling thread is sx_last_error().
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP
 *     device memory on the context's device (from sx_malloc, hipMalloc or
 *     torch.Tensor.data_ptr()).  Functions ending in _dev take device
 *     pointers for every array argument and only enqueue work on the
 *     context's stream (no host synchronisation, hipGraph-capturable unless
 *     stated); functions without the suffix take host pointers, copy in and
 *     out and return when the outputs are valid.
 *   - all floating point data is IEEE binary64.  Index vectors are int64,
 *     sparse inner indices int32, sparse pointers int64, flags uint8.
 *   - the caller owns every array it passes; the library owns handles.
 *   - a context must not be used from two threads at the same time.
 */
#ifndef SXHIP_H
#define SXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SX_ABI_VERSION 1

#define SX_OK 0
#define SX_ERR_INVALID (-1)     /* NULL / negative size / inconsistent shapes  -> ValueError  */
#define SX_ERR_HIP (-2)         /* HIP runtime failure                           -> RuntimeError */
#define SX_ERR_NOMEM (-3)       /* host or device allocation failed              -> MemoryError  */
#define SX_ERR_UNSUPPORTED (-4) /* valid request this build cannot serve         -> NotImplementedError */

/* bits of the per-column code written by sx_score_columns */
#define SX_CODE_LOW 1u /* x - l < gamma * s_d      : fix to lower bound */
#define SX_CODE_UP 2u  /* u - x < gamma * (-s_d)   : fix to upper bound */

typedef struct sx_ctx sx_ctx;       /* device + stream + workspace */
typedef struct sx_matrix sx_matrix; /* one sparse matrix resident in HBM, CSR and CSC */

/* ------------------------------------------------------------------ library */
int sx_abi_version(void);
const char *sx_last_error(void);
/* number of visible HIP devices (does not create a context) */
int sx_device_count(int *count);

/* ------------------------------------------------------------------ context */
/* stream: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream), or NULL
 * to let the context create its own non-blocking stream. */
int sx_ctx_create(int device, void *stream, sx_ctx **out);
int sx_ctx_destroy(sx_ctx *ctx);
int sx_ctx_sync(sx_ctx *ctx);
/* Tuning knobs (performance only; the kernels of the scoring / pricing / CG path return identical bits
 * under every setting, the simplex the same optimum up to rounding): "xcd_swizzle" 0/1 (default 1), "nt_stream"
 * (cache policy of the streamed entry loads: 0 plain [default], 1 non-temporal, 2/16/17/18 buffer loads
 * with nt / sc1 / sc0 sc1 / nt sc1), "chunk" 2048/4096 (default 4096), "window" (LDS operand window of the
 * column walk in K1/K10: -1 auto [default: decided per matrix from its index clustering on first use],
 * 0 off, 1/2/4/8 tiles per window load), "graph" 0/1 (default 1: hipGraph replay of the CG iteration batch),
 * "rb_stage_long" (0/1, default 0: in the column-blocked row layout the products of the long rows come from a
 * column-ordered pre-pass instead of one gathered 128-byte line per entry -- bit-identical, measured slower, kept
 * as an experiment; read when a layout is built),
 * "slabs" (operand slabs of a walk whose gathers have no locality, csrc/sx_slabs.h -- K1, K2, K10: -1 auto [default:
 * operand beyond 3.6 MB, one slab per 3.2 MB, at least 4M entries, carries no heavier than 1.25 x the entry stream; the
 * LDS window and the column-blocked rows are asked first], 0 never, 2..256 that many slabs; bit-identical either way --
 * a matrix with a segment whose indices descend somewhere keeps the plain walk),
 * "run_prefetch" (0/1, default 0: the windowed column walk with its loads one step ahead, csrc/sx_runwalk.h --
 * bit-identical, measured 4 % slower, kept as an experiment),
 * "spx_defer" (basis inverse of sx_simplex_solve*: 0 = rank-one update after every pivot, 1 = the updates
 * of a batch of 64 pivots are kept in product form and folded in as one rank-64 update by a hand-written kernel,
 * -1 auto [default]),
 * "spx_pricing" (entering variable of sx_simplex_solve*: 0 = Dantzig, largest reduced cost; 1 = Devex
 * reference weights [default], fewer pivots on general LPs, identical to Dantzig on network matrices),
 * "spx_check" (pivots between two checks of A x + s = b against the explicit inverse's drift, a multiple of 64,
 * default 2048; a failed check rebuilds the inverse from the basis columns) and "spx_force_reinvert" (0/1,
 * tests: rebuild at every check),
 * "rowblock" (column-blocked copy of the rows for the row walk of sx_score_rows and the projector CG:
 * -1 auto [default: built on first use for matrices of >= 4M entries when at least half of them fall into
 * column blocks dense enough for an LDS window], 0 off, 1 whenever the matrix admits it; results are
 * bit-identical either way -- a matrix with a row whose column indices descend somewhere keeps the plain walk),
 * "netsimplex" (sx_netsimplex_dev: -1 / 1 = run whenever problem and basis are in its domain [default],
 * 0 = always answer status 5, i.e. send network re-solves to the general simplex), "ns_lds" (0/1, default 1:
 * tree arrays and potentials of the network simplex in LDS when they fit, V <= 4480), "ns_block" (arcs priced
 * per lane and block by the network simplex: 0 = by size [default], 1..64), "netdual" (sx_netdual_dev: -1 / 1 =
 * run whenever problem and basis are in its domain [default], 0 = always answer status 5) and "nd_grid"
 * (workgroups of its cooperative grid: 0 = by size [default], 1..256).
 * Unknown keys return SX_ERR_INVALID. */
int sx_ctx_set_option(sx_ctx *ctx, const char *key, int64_t value);
/* name (e.g. "gfx950:sramecc+:xnack-"), CU count and total HBM bytes of the context's device */
int sx_ctx_device_info(sx_ctx *ctx, char *name, size_t name_len, int *cu_count, uint64_t *hbm_bytes);

/* plumbing for hosts that do not bring their own device allocator */
int sx_malloc(sx_ctx *ctx, size_t bytes, void **dev_out);
int sx_free(sx_ctx *ctx, void *dev);
int sx_upload(sx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int sx_download(sx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int sx_memset(sx_ctx *ctx, void *dst_dev, int byte, size_t bytes);

/* HIP-event stopwatch on the context's stream (nestable up to 8 deep) */
int sx_timer_start(sx_ctx *ctx);
int sx_timer_stop(sx_ctx *ctx, float *ms_out); /* synchronises on the stop event */
/* Non-blocking markers: record HIP event number `id` (0 <= id < 4096) on the stream; later ask for
 * the time between two recorded markers (synchronises on the later one).  Lets a benchmark time one
 * kernel inside a loop without putting a host sync in the loop. */
int sx_marker_record(sx_ctx *ctx, int id);
int sx_marker_elapsed(sx_ctx *ctx, int id_from, int id_to, float *ms_out);
/* hipDeviceSynchronize on the context's device (all streams) */
int sx_ctx_sync_device(sx_ctx *ctx);
/* One big device block per context, kept between calls and allocated AHEAD of its use on a helper thread: the sparse
 * crossover (sx_crossover_band_*_dev) carves its eta file and tableau out of it -- hipMalloc of the 28 GB it needs at 1e6
 * rows takes 0.5-1.4 s, which a caller hides behind the first-order stage by asking for the block first.  Returns at once;
 * a block of that size already held or on its way: nothing happens; bytes = 0 frees what the context holds; a request
 * beyond 60 % of the free memory is ignored (the call then allocates what it can by itself). */
int sx_ctx_prefetch_block(sx_ctx *ctx, size_t bytes);
/* Device memory pool of the library (process-wide, per device): the library's own allocations -- matrices, factors, work
 * blocks -- are kept when freed and handed to the next request of their size class, so that a pipeline of crossovers
 * (the reference calls run_perturb_algorithm once per LP, lp_methods/algorithms.py:42-74) neither pays hipMalloc again nor
 * runs into the driver's clean-up of what the previous call freed.  At most 16 GiB are kept per device (SX_POOL_MAX, bytes),
 * blocks above 8 GiB are not (SX_POOL_BLOCK_MAX); SX_POOL=0 turns the pool off.  sx_pool_trim hands everything kept back
 * to the driver; sx_pool_stats reports bytes kept / bytes in use and requests served by the pool / by the driver. */
int sx_pool_trim(void);
int sx_pool_stats(uint64_t *cached_bytes, uint64_t *live_bytes, uint64_t *hits, uint64_t *misses);

/* ------------------------------------------------------------------ matrix */
/* Upload an m x n matrix.  Host CSR arrays are required (the reference keeps A as scipy CSR,
 * formats.py:18); host CSC arrays are optional -- when NULL the library derives them by a stable
 * counting sort, which yields per-column entries in row-major walk order, the accumulation order of
 * scipy's A.transpose() @ y (formats.py:72).  When given they must be in that order.
 * Duplicates and unsorted inner indices are allowed (they are walked as stored). */
int sx_matrix_create(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const int64_t *csr_rowptr,
                     const int32_t *csr_col, const double *csr_val, const int64_t *csc_colptr,
                     const int32_t *csc_row, const double *csc_val, sx_matrix **out);
/* Single-layout matrix for shards: is_csc != 0 -> (ptr, idx, val) are colptr[n+1]/row/val and only
 * the column kernels (sx_score_columns, sx_price) accept it; is_csc == 0 -> rowptr[m+1]/col/val and
 * only the row kernels accept it.  A kernel that needs the missing layout returns SX_ERR_INVALID. */
int sx_matrix_create_single(sx_ctx *ctx, int64_t m, int64_t n, int64_t nnz, int is_csc,
                            const int64_t *ptr, const int32_t *idx, const double *val,
                            sx_matrix **out);
int sx_matrix_destroy(sx_matrix *A);
int sx_matrix_dims(const sx_matrix *A, int64_t *m, int64_t *n, int64_t *nnz);
/* device addresses of the six arrays (for tests and zero-copy consumers) */
int sx_matrix_arrays(const sx_matrix *A, const int64_t **csr_rowptr, const int32_t **csr_col,
                     const double **csr_val, const int64_t **csc_colptr, const int32_t **csc_row,
                     const double **csc_val);
/* download CSR arrays of a device matrix (rowptr[m+1], col[nnz], val[nnz]) */
int sx_matrix_download_csr(const sx_matrix *A, int64_t *rowptr, int32_t *col, double *val);
/* Column-blocked row layout of A under the context's "rowblock" option (built now if it is due):
 * info[0..5] = super-tiles, cells, chunks, entries incl. gaps, entries in windowed cells, uint16 slots per
 * cell of the row-start table; all zero when the matrix uses the plain walk.  The second call copies the
 * layout's arrays to the host (NULL = skip): st[info0] (24-byte records), chunks[info2] (32-byte records),
 * rowstart[info1 * info5], idx / val [info3 + 8] -- for tests and tools (csrc/sx_rowblock.h). */
int sx_matrix_rowblock_info(sx_ctx *ctx, const sx_matrix *A, int64_t *info);
int sx_matrix_rowblock_download(sx_ctx *ctx, const sx_matrix *A, void *st, void *chunks, uint16_t *rowstart,
                                int32_t *idx, double *val);
/* Operand slabs of A's row walk (which = 0) or column walk (which = 1) under the context's "slabs" option (built now
 * if they are due; csrc/sx_slabs.h): info[0..2] = slabs, operand indices per slab, segments; all zero when the walk
 * is the plain one.  For tests, tools and bench.py; the walks (sx_score_columns_dev / sx_score_rows_dev /
 * sx_price_dev, reference formats.py:70-76, net_manager.py:302-303) pick the slabs up by themselves. */
int sx_matrix_slabs_info(sx_ctx *ctx, const sx_matrix *A, int which, int64_t *info);

/* ------------------------------------------------------------------ K1: column scoring
 * replaces GeneralLP.get_dual_slack (formats.py:70-72) and the two np.where tests of
 * get_perturb_problem (lp_methods/algorithms.py:99,104-105):
 *     s_d[j]  = c[j] - sum_i a_ij * y[i]      (entries in stored CSC order, product and sum
 *                                               rounded separately, running sum from +0.0)
 *     code[j] = SX_CODE_LOW * (x[j]-l[j] < gamma*s_d[j])  |  SX_CODE_UP * (u[j]-x[j] < gamma*(-s_d[j]))
 * y has m entries; c, x, l, u, s_d, code have n.  s_d or code may be NULL (output skipped).
 * Unless the "window" option is 0, the first sx_score_columns_dev / sx_price_dev call on a matrix with
 * m >= 4096 builds its window table (one kernel, a device allocation and a stream synchronisation):
 * make that call outside hipGraph capture; later calls only enqueue. */
int sx_score_columns_dev(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                         const double *x, const double *l, const double *u, double gamma,
                         double *s_d, uint8_t *code);
int sx_score_columns(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                     const double *x, const double *l, const double *u, double gamma, double *s_d,
                     uint8_t *code);

/* ------------------------------------------------------------------ K2: row scoring
 * replaces GeneralLP.get_primal_slack (formats.py:74-76) and the row test
 * (lp_methods/algorithms.py:100,106):
 *     s_p[i]  = b[i] - sum_j a_ij * x[j]      (stored CSR order, same rounding rule)
 *     flag[i] = s_p[i] < gamma_dual * (-y[i])
 * x has n entries; b, y, s_p, flag have m. */
int sx_score_rows_dev(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *b,
                      const double *y, double gamma_dual, double *s_p, uint8_t *flag);
int sx_score_rows(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *b,
                  const double *y, double gamma_dual, double *s_p, uint8_t *flag);

/* ------------------------------------------------------------------ index sets
 * np.where(...)[0] of a flag vector (lp_methods/algorithms.py:104-106): ascending int64 positions
 * i with (flags[i] & mask) != 0.  idx_out needs room for n entries; *count_out (device int64 for
 * _dev, host int64 otherwise) receives how many were written. */
int sx_select_indices_dev(sx_ctx *ctx, int64_t n, const uint8_t *flags, uint8_t mask,
                          int64_t *idx_out, int64_t *count_out);
int sx_select_indices(sx_ctx *ctx, int64_t n, const uint8_t *flags, uint8_t mask, int64_t *idx_out,
                      int64_t *count_out);

/* ------------------------------------------------------------------ K3: perturbed cost
 * replaces perturb_c / get_x_perturb_val after the random direction and the scale factor are
 * known (lp_methods/algorithms.py:130-132,139-141,148-151,196-202).  xi is the normalised
 * direction (legacy MT19937, seed 42, U(0.9,1), divided by its 2-norm) supplied by the host.
 *   is_feas != 0 : c_pt = c + xi
 *   otherwise    : xr = min(x-l, u-x); free -> x; xr < 1e-6 -> 1e-6; free -> 1
 *                  p = min(xi / xr * scale_factor / 1e-2, 1e6); free -> 0; c_pt = c + p */
int sx_perturb_cost_dev(sx_ctx *ctx, int64_t n, const double *x, const double *l, const double *u,
                        const double *c, const double *xi, double scale_factor, int is_feas,
                        double *c_pt);
int sx_perturb_cost(sx_ctx *ctx, int64_t n, const double *x, const double *l, const double *u,
                    const double *c, const double *xi, double scale_factor, int is_feas,
                    double *c_pt);

/* ------------------------------------------------------------------ K10: pricing
 * replaces get_reduced_cost_for_original_mcf + the reduced-cost half of
 * check_optimality_condition (network_methods/net_manager.py:302-303,318 and :483,496):
 *     rc[j] = c[j] - sum_i a_ij*y[i];  rc[j] = -rc[j] where vbasis[j] == -2
 * vbasis (int8: 0 basic, -1 at lower, -2 at upper, -3 free) may be NULL (no flips); rc may be NULL.
 * result: most negative reduced cost, its smallest column index, and the number of columns that
 * fail rc >= -tol (NaN counts as failing). */
typedef struct sx_price_result {
    double min_rc;
    int64_t argmin;
    int64_t n_violating;
} sx_price_result;
int sx_price_dev(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
                 const int8_t *vbasis, double tol, double *rc, sx_price_result *result_dev);
int sx_price(sx_ctx *ctx, const sx_matrix *A, const double *y, const double *c,
             const int8_t *vbasis, double tol, double *rc, sx_price_result *result);

/* ------------------------------------------------------------------ K6 / K12: sub-problem
 * replaces LPManager.fix_variables + update_subproblem (lp_methods/lp_manager.py:40-66) and
 * MCFManagerStd.update_subproblem (network_methods/net_manager.py:202-209).
 * sx_compact_columns_dev: non_fix = ascending columns with code[j] == 0 (device int64, room for n;
 *   *n_sub receives the count on the host); *A_sub = A[:, non_fix] as a new resident matrix (CSR and
 *   CSC, entry order preserved, destroy with sx_matrix_destroy).  Blocking (sizes come back to the
 *   host to allocate the result).
 * sx_fixed_rhs_dev: b_sub = b - A[:, up] @ u[up] - A[:, low] @ l[low] with up / low = columns whose
 *   code has SX_CODE_UP / SX_CODE_LOW, bit-exact with the reference's two column-sliced products.
 * sx_gather_f64_dev: dst[k] = src[idx[k]]  (c, l, u of the sub-problem; get_subx). */
int sx_compact_columns_dev(sx_ctx *ctx, const sx_matrix *A, const uint8_t *code, sx_matrix **A_sub,
                           int64_t *non_fix, int64_t *n_sub);
/* A[:, idx] for an explicit list of nsub distinct columns in any order (device int64): column k of
 * the result is column idx[k] of A; inside every row the entries keep their stored order, as scipy's
 * CSR fancy column indexing does (the column-generation managers append released columns in queue
 * order, network_methods/net_manager.py:204-205,242).  Blocking. */
int sx_gather_columns_dev(sx_ctx *ctx, const sx_matrix *A, const int64_t *idx, int64_t nsub,
                          sx_matrix **A_sub);
int sx_fixed_rhs_dev(sx_ctx *ctx, const sx_matrix *A, const uint8_t *code, const double *u,
                     const double *l, const double *b, double *b_sub);
int sx_gather_f64_dev(sx_ctx *ctx, int64_t n, const int64_t *idx, const double *src, double *dst);

/* ------------------------------------------------------------------ K7: MCF flow indicators
 * replaces the arithmetic of MCFManagerStd.get_sorted_flows (network_methods/net_manager.py:165-182):
 *     mask = x > u/2;  x_hat = x*(~mask) + u*mask - x*mask, 0 where x < 0 or x > u
 *     a_bar = -a on masked arcs;  f_i = max(sum_{a_bar>0} a_bar*x_hat, sum_{a_bar<0} |a_bar|*x_hat)
 *     ind_j = max_i |(f_inv_i * x_hat_j) * a_bar_ij|,  f_inv = 1/f where f != 0 else 0
 * node sums run over the arcs of a node in stored CSR order (canonical = ascending arc), bit-exact
 * with the reference for matrices in canonical form.  A: V x E with both layouts; x, u, ind: E.
 * xhat_out (E) and f_out (V) are optional outputs of the intermediates (NULL to skip). */
int sx_flow_indicator_mcf_dev(sx_ctx *ctx, const sx_matrix *A, const double *x, const double *u,
                              double *ind, double *xhat_out, double *f_out);

/* ------------------------------------------------------------------ K8: OT flow indicators
 * replaces OTManager.get_sorted_flows (network_methods/net_manager.py:377-378):
 *     ind[i*D + j] = max(X[i*D+j] / s[i], X[i*D+j] / d[j])    (IEEE division, numpy.maximum) */
int sx_flow_indicator_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *X, const double *s,
                             const double *d, double *ind);

/* ------------------------------------------------------------------ K13: TNET spanning tree
 * replaces max_weight_spanning_tree (network_methods/tree_BI.py:32-59: scipy minimum_spanning_tree of
 * the negated weights on the bipartite supplier x demander graph).  w: S*D flow weights, row-major;
 * arcs with w <= 0 (or NaN) are not edges.  in_tree: S*D flags, 1 on the arcs of the maximum-weight
 * spanning forest (S + D - 1 arcs when the positive-weight graph is connected).  Equal weights are
 * ordered by ascending arc index (the reference's order among ties is unspecified). */
int sx_spanning_tree_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *w, uint8_t *in_tree);

/* ------------------------------------------------------------------ K9: ranking
 * replaces np.argsort(indicators)[::-1] (network_methods/net_manager.py:184,379).  The reference's
 * default sort is unstable, so ties are defined here: descending key; equal keys by descending
 * index (== np.argsort(key, kind="stable")[::-1]); NaN ranks first.  idx_out: n int64.  Blocking. */
int sx_argsort_desc_dev(sx_ctx *ctx, int64_t n, const double *key, int64_t *idx_out);

/* ------------------------------------------------------------------ K10 on the OT structure
 * replaces get_reduced_cost_for_original_OT + the reduced-cost test (network_methods/net_manager.py:
 * 483,496) without materialising the kron incidence matrix of formats.py:156-159:
 *     rc[i*D + j] = M[i*D + j] - ((0 + (-1)*y[i]) + (+1)*y[S + j])
 * y has S + D entries (more are ignored); rc may be NULL; result as in sx_price_dev. */
int sx_price_ot_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *M, const double *y, double tol,
                    double *rc, sx_price_result *result_dev);

/* ------------------------------------------------------------------ K4: projector norm
 * replaces get_projector_Xc / apply_projector (lp_methods/algorithms.py:162-172,183-187; the
 * no-free-variable branch) and feeds get_scale_factor (:190-193):
 *     Y = [A, I_<] diag(xx),  v = diag(xx) c_std,  proj = v - Y^T cg(Y Y^T, Y v, tol, maxiter)
 * xa = xx[:n]; xs = the slack part of xx scattered to a length-m vector (entry i = xx of the slack
 * of row i when row i is '<', 0 when it is '='); c = structural costs (slack costs are 0).
 * The device never forms Y Y^T: each CG iteration is one CSC and one CSR pass over A.  Stopping
 * rule of the reference's scipy: x0 = 0; return at once when ||Yv|| <= tol; stop when
 * ||r|| < tol*||Yv|| (tested before every iteration) or after maxiter iterations.
 * Blocking call (the host polls a device flag every 25 iterations); _dev takes device pointers.
 * Floating-point contract: proj_norm agrees with the reference to 1e-6 relative when CG converges
 * (the reference's own explicit-YY^T arithmetic differs from any matrix-free one at that level). */
typedef struct sx_cg_result {
    double proj_norm;    /* ||proj||_2 */
    double b_norm;       /* ||Y v||_2 */
    double rel_residual; /* ||r|| / ||Y v|| at exit */
    int64_t iters;       /* completed CG iterations */
    int64_t converged;   /* 1 when the stopping rule fired, 0 when maxiter was hit */
} sx_cg_result;
int sx_projector_norm_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                          const double *c, double tol, int maxiter, sx_cg_result *result);
/* same, also returning the projection itself: proj_cols[n] = xa .* (c - A^T z) and
 * proj_rows[m] = -xs .* z (device pointers, either may be NULL) -- what apply_projector returns
 * (lp_methods/algorithms.py:187), with the slack block scattered to row positions */
int sx_projector_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                     const double *c, double tol, int maxiter, double *proj_cols, double *proj_rows,
                     sx_cg_result *result);
/* same with a cost on the slack columns as well (cs[m], read on the rows where xs != 0; NULL = zero):
 * v = [xa .* c ; xs .* cs], proj_rows[m] = xs .* (cs - z).  Needed by the free-variable branch of
 * get_projector_Xc (lp_methods/algorithms.py:174-180), whose adjusted cost c - A_1^T A_2 t is non-zero on
 * slack columns; the free columns themselves enter with a large scale xa[j] and zero cost, which is the
 * penalty form of the reference's QP  min ||x - v||^2  s.t.  A_1 X_1 x + A_2 f = 0. */
int sx_projector_std_dev(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                         const double *c, const double *cs, double tol, int maxiter, double *proj_cols,
                         double *proj_rows, sx_cg_result *result);
/* get_x_perturb_val (lp_methods/algorithms.py:196-202): min(x-l, u-x), free columns -> x; with
 * apply_floor != 0 also the two overwrites of perturb_c (:131-132): values < 1e-6 -> 1e-6, free
 * columns -> 1 */
int sx_x_real_dev(sx_ctx *ctx, int64_t n, const double *x, const double *l, const double *u,
                  int apply_floor, double *x_real);
/* dst[i] = mask[i] ? src[i] : 0 */
int sx_mask_f64_dev(sx_ctx *ctx, int64_t n, const double *src, const uint8_t *mask, double *dst);
int sx_projector_norm(sx_ctx *ctx, const sx_matrix *A, const double *xa, const double *xs,
                      const double *c, double tol, int maxiter, sx_cg_result *result);

/* Free-variable branch of get_projector_Xc (lp_methods/algorithms.py:173-180: cg on the normal equations of the
 * free columns, adjusted cost, then a Gurobi QP) on the device: free_idx[nf] (device, ascending) names the free
 * columns, xa[n] / xs[m] the scales of the structural / slack columns, row_lt[m] the '<' rows.  The QP is solved
 * in penalty form by the same CG (free columns at a large scale, zero cost).  proj_cols[n] / proj_rows[m] receive
 * the projection on the QP's variables -- entries of free columns and of '=' rows are zero -- and
 * result->proj_norm its norm.  Parity with Gurobi is unpinned; tests compare with the exact QP minimiser.
 * Blocking. */
int sx_projector_free_dev(sx_ctx *ctx, const sx_matrix *A, int64_t nf, const int64_t *free_idx, const double *xa,
                          const double *xs, const double *c, const uint8_t *row_lt, double *proj_cols,
                          double *proj_rows, sx_cg_result *result);

/* ------------------------------------------------------------------ sharded variants (one process per GPU)
 * K4 with the columns of Y sharded over ranks: the rank holds a column block A_loc (both
 * layouts) with its slices xa_loc, c_loc; xs, cs and every m-vector are replicated.  Protocol, all calls
 * asynchronous on the context's stream unless stated:
 *   sx_cg_shard_open    this rank's share of b = Y v lands in *reduce_vec (device, m doubles)
 *   [host: all-reduce(SUM) of reduce_vec over the ranks, on the same stream]
 *   sx_cg_shard_start   takes the reduced b (blocking: returns ||b|| and whether the loop is trivial)
 *   per iteration k = 0, 1, ...:  sx_cg_shard_local  (w = xa^2 .* A_loc^T p, q = A_loc w -> reduce_vec)
 *                                 [host: all-reduce(SUM) of reduce_vec]
 *                                 sx_cg_shard_update(k)  (slack block, alpha, z, r, stop test, beta, p)
 *   sx_cg_shard_poll    (blocking) done flag and iteration count -- every few dozen iterations
 *   sx_cg_shard_finish  (blocking) squared norm of this rank's columns of the projection (sum them over the
 *                       ranks) and of the slack rows (identical on every rank: count once); optional vectors
 * Kernels become no-ops once the stop test fired, so iterations enqueued past the end are harmless. */
typedef struct sx_cg_shard sx_cg_shard;
int sx_cg_shard_open(sx_ctx *ctx, const sx_matrix *A_loc, const double *xa_loc, const double *xs,
                     const double *c_loc, const double *cs, double tol, sx_cg_shard **out, double **reduce_vec);
int sx_cg_shard_start(sx_cg_shard *h, double *bnorm_out, int *trivial_out);
int sx_cg_shard_local(sx_cg_shard *h);
int sx_cg_shard_update(sx_cg_shard *h, int parity);
int sx_cg_shard_poll(sx_cg_shard *h, int *done, int64_t *iters);
int sx_cg_shard_finish(sx_cg_shard *h, double *proj_cols_loc, double *proj_rows, double *cols_sumsq,
                       double *rows_sumsq, sx_cg_result *result);
int sx_cg_shard_close(sx_cg_shard *h);
/* K7 step by step for arcs sharded over ranks (network_methods/net_manager.py:165-182; BASELINE config 4): x_hat
 * and the big-flow mask of the own arcs; node throughputs f_inv (and f) of the own node block A_rows (its rows
 * over ALL arcs, row layout) from the all-gathered x_hat / mask; indicators of the own arc block A_cols (column
 * layout) from the all-gathered f_inv.  Every sum is taken inside one rank, in the single-process order. */
int sx_mcf_xhat_dev(sx_ctx *ctx, int64_t E, const double *x, const double *u, double *xhat, uint8_t *mask);
int sx_mcf_node_flows_dev(sx_ctx *ctx, const sx_matrix *A_rows, const double *xhat_all, const uint8_t *mask_all,
                          double *f_inv, double *f_out);
int sx_mcf_arc_indicator_dev(sx_ctx *ctx, const sx_matrix *A_cols, const double *xhat_loc, const uint8_t *mask_loc,
                             const double *f_inv_all, double *ind);

/* ------------------------------------------------------------------ K16: device simplex
 * replaces the third-party re-solves behind the SolverCaller seam for the simplex family
 * (solver_caller/solving.py:32-68; call sites lp_methods/algorithms.py:69-74 and
 * network_methods/net_manager.py:222,468):
 *     min c^T x   s.t.  (A x)_i = b_i  (row_is_lt[i] == 0)  or  <= b_i  (== 1),   l <= x <= u
 * Bounded revised primal simplex, two phases, explicit dense basis inverse in HBM (8 m^2 bytes must fit the
 * free HBM, else SX_ERR_UNSUPPORTED) updated once per batch of 64 pivots and rebuilt from the basis columns
 * whenever A x + s = b fails its periodic check, Devex pricing with Bland fallback (options "spx_defer",
 * "spx_pricing", "spx_check" above).  vbasis_in[n] / cbasis_in[m] (both or neither; Gurobi codes 0 basic,
 * -1 lower, -2 upper, -3 free or superbasic / 0 basic, -1 non-basic) give a warm start: its columns are
 * pivoted in where they find a row, and a basis that is not primal feasible goes through phase 1 (bound
 * violations of basic structurals and logicals alike are driven out) instead of being dropped.  status 0 is
 * only reported for a point that satisfies A x + s = b to feas_tol (relative to 1 + max|b|).  Outputs
 * (device, any may be NULL): x[n], y[m] with reduced cost = c - A^T y, vbasis[n], cbasis[m].  Blocking.  All
 * arrays device. */
typedef struct sx_simplex_result {
    int64_t status;          /* 0 optimal, 1 infeasible, 2 unbounded, 3 iteration limit, 4 numerical trouble,
                              * 5 (sx_netsimplex_dev only) problem or start basis outside the solver's domain */
    int64_t iters;           /* pivots and bound flips, both phases */
    int64_t phase1_iters;
    int64_t warm_start_used; /* 1 when the given basis was installed, 2 when a session's inverse was reused */
    double obj;              /* c^T x */
    double max_violation;    /* largest bound violation of a basic variable at exit */
} sx_simplex_result;
int sx_simplex_solve_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                         const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                         const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol,
                         double *x, double *y, int8_t *vbasis, int8_t *cbasis, sx_simplex_result *result);

/* Crossover proper (what Gurobi's barrier + crossover does for the re-solve of lp_methods/algorithms.py:50-54):
 * start from the interior point x_start[n] -- every column coded 0 or -3 in vbasis_in sits at its x_start
 * value, clipped to its bounds, as a superbasic variable; the slack of a '<' row is b - A x_start; logicals
 * coded 0 in cbasis_in keep their rows, the other rows go to the columns coded 0 that can take them -- so the
 * first basic solution is the point itself, and the primal simplex then pushes the superbasic variables to a
 * bound or into the basis one pivot at a time until it stands on an optimal vertex.  Same outputs as
 * sx_simplex_solve_dev; a variable that is still superbasic at exit is reported as -3. */
int sx_simplex_crossover_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                             const double *u, const uint8_t *row_is_lt, const int8_t *vbasis_in,
                             const int8_t *cbasis_in, const double *x_start, int64_t max_iter, double feas_tol,
                             double opt_tol, double *x, double *y, int8_t *vbasis, int8_t *cbasis,
                             sx_simplex_result *result);

/* First-order stage of the same re-solve (K16p, csrc/sx_pdlp.hip): what the reference's backends do with
 * "barrier" before their crossover (lp_methods/algorithms.py:50-54 -> solver_caller/gurobi.py:111-115) -- carry
 * the interior point of the original LP next to the optimum of the PERTURBED sub-problem, so that the crossover
 * (sx_simplex_crossover_dev) needs few pivots.  Restarted, diagonally preconditioned primal-dual hybrid gradient
 * (PDLP) on  min c^T x, A x (= | <=) b, l <= x <= u  from (x0, y0) (either may be NULL = 0): per iteration one walk
 * of the column layout and one of the row layout with the proximal steps fused in; KKT errors of iterate and
 * running average, restart decision and primal weight every 64 iterations on the device; stops when primal
 * residual <= tol (1 + ||b||), dual residual <= tol (1 + ||c||) and gap <= tol (1 + |p| + |d|), or after max_iter
 * iterations (<= 0: 20000; rounded up to a multiple of 64).  Outputs x[n] (inside its bounds; at a bound exactly
 * where the projection put it) and y[m] (reduced cost = c - A^T y, y <= 0 on '<' rows).  Blocking; arrays device.
 * Third-party counterpart absent (Gurobi's barrier): parity unpinned; oracle/pdlp.py states the iteration. */
typedef struct sx_pdlp_result {
    int64_t status;          /* 0 converged to tol, 3 iteration limit */
    int64_t iters;
    int64_t restarts;
    double primal_residual;  /* || violation of A x (=|<=) b ||_2 */
    double dual_residual;    /* || part of c - A^T y no bound can absorb ||_2 */
    double gap;              /* | c^T x - (b^T y + l^T rc+ + u^T rc-) | */
    double primal_obj, dual_obj;
    double b_norm, c_norm;
    double step, primal_weight;
} sx_pdlp_result;
int sx_pdlp_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l, const double *u,
                const uint8_t *row_is_lt, const double *x0, const double *y0, int64_t max_iter, double tol, double *x,
                double *y, sx_pdlp_result *result);

/* ------------------------------------------------------------------ K16f: band LU of a basis
 * The reference leaves basis factorisations to its solvers (solver_caller/gurobi.py:202-210); the sparse device
 * crossover (sx_crossover_band_dev) factors its starting basis -- a band matrix in the natural order of a staged
 * LP once the dense rows are set aside -- with this module (csrc/sx_bandlu.hip).  n x n matrix with kl sub- and ku
 * super-diagonals given as device triplets (row, column, value: int32, int32, double); LAPACK general-band storage
 * in HBM ((2 kl + ku + 1) n doubles).  factor: partial pivoting inside the band, panels of 32 columns; a column
 * whose largest candidate pivot is <= pivot_tol is REPLACED by the unit vector of the row on its diagonal and
 * reported (replaced_host[n], host, may be NULL; ipiv_host[n] the row swapped with j at step j).  solve: nrhs
 * right-hand sides, column major with leading dimension ldx, in place; trans 0: A x = b, 1: A^T x = b (A = the
 * matrix with its replaced columns).  solve_sparse: A x = b for right-hand sides with few entries each (the columns
 * of an LP): panels of 32 rows whose entries are all <= tiny in magnitude are skipped, in the forward sweep (before the
 * first entry of b, and once the fill past its last entry has decayed below tiny) and in the backward sweep; tiny = 0
 * skips exact zeros only and returns the result of the plain solve's sequential sweeps (solve takes partitioned sweeps --
 * blocks of panels side by side, rounding differently -- for up to 1,024 right-hand sides unless SX_BANDLU_SEQ is set).  Limits: kl + 32 <= 1536, kl + ku + 32 <= 1980.
 * Blocking (factor) / stream-ordered (solve); arrays device unless named host. */
typedef struct sx_bandlu sx_bandlu;
int sx_bandlu_create_dev(sx_ctx *ctx, int64_t n, int kl, int ku, int64_t nnz, const int32_t *row, const int32_t *col,
                         const double *val, sx_bandlu **out);
int sx_bandlu_factor_dev(sx_bandlu *h, double pivot_tol, int64_t *n_replaced_out, int32_t *replaced_host,
                         int32_t *ipiv_host);
/* factor_blocks: the same for a matrix that is block diagonal with identity padding between its blocks -- block b holds the
 * columns and rows [b stride, b stride + real_len) (real_len_last for the last of nblocks), identity up to the next
 * block, stride >= real_len + kl + ku + 32, no entry coupling two blocks: the blocks' panels are factored side by side
 * (the sparse crossover cuts its band at separators that go to the border; csrc/sx_crossover_band.hip). */
int sx_bandlu_factor_blocks_dev(sx_bandlu *h, double pivot_tol, int nblocks, int64_t stride, int64_t real_len,
                                int64_t real_len_last, int64_t *n_replaced_out, int32_t *replaced_host, int32_t *ipiv_host);
int sx_bandlu_solve_dev(sx_bandlu *h, int trans, int64_t nrhs, double *X, int64_t ldx);
int sx_bandlu_solve_sparse_dev(sx_bandlu *h, int64_t nrhs, double *X, int64_t ldx, double tiny);
int sx_bandlu_destroy(sx_bandlu *h);

/* ------------------------------------------------------------------ K16g: dense LU of a Schur complement
 * Companion of K16f for the BORDER of a bordered band basis (csrc/sx_border.h, sx_border.hip): the linking rows of a staged LP, the rows
 * the band matching leaves over and the separators between the band's blocks form a dense Schur complement of a few
 * thousand rows -- factored here (csrc/sx_denselu.hip) the way the reference's solvers factor whatever basis they meet
 * (solver_caller/gurobi.py:202-210 model.optimize()).  n x n, n <= 16384, column major on the device.  factor: partial
 * pivoting, right-looking, panels of 8 inside blocks of 64, trailing updates on the fp64 matrix cores; a column whose
 * largest candidate pivot is <= pivot_tol is REPLACED by the unit vector of the row on its diagonal and reported
 * (replaced_host[n]; rowperm_host[n]: the original row that ended at position i -- a replaced column j stands for the unit
 * vector of row rowperm_host[j]).  solve: nrhs right-hand sides, column major with leading dimension ldx, in place; trans 0:
 * A x = b, 1: A^T x = b (A = the matrix with its replaced columns).  Blocking (factor) / stream-ordered (set, solve). */
typedef struct sx_denselu sx_denselu;
int sx_denselu_create_dev(sx_ctx *ctx, int64_t n, sx_denselu **out);
int sx_denselu_set_dev(sx_denselu *h, const double *src, int64_t lds);
int sx_denselu_factor_dev(sx_denselu *h, double pivot_tol, int64_t *n_replaced_out, int32_t *replaced_host,
                          int32_t *rowperm_host);
int sx_denselu_solve_dev(sx_denselu *h, int trans, int64_t nrhs, double *X, int64_t ldx);
int sx_denselu_destroy(sx_denselu *h);

/* Sparse crossover (K16s, csrc/sx_crossover_band.hip + csrc/sx_border.hip): the same job as sx_simplex_crossover_dev --
 * from the point x_start[n] (what sx_pdlp_dev leaves: columns at a bound exactly where the projection put them) to an
 * optimal vertex and its basis -- without the dense m x m inverse.  The basis is factored in BORDERED form: rows in their
 * natural (or Cuthill-McKee) order, rows with more than max(24, 6 x average) entries ("dense", the linking rows) set aside;
 * the basic variables matched to the band rows by entry size form a band matrix (K16f), the others (linking activities,
 * logicals of dense rows, what the band LU set aside) the border, whose Schur complement goes to the dense LU (K16g).
 * Every basic variable is in the factors: a fresh factorisation (full eta file, numerical trouble, the final check of the
 * vertex against A) is a true refactorisation.  The simplex works on an explicit tableau of the columns that can still
 * move (superbasic ones + what pricing adds; grown on demand), every entering column kept as an eta vector for duals and
 * new columns.  Memory O(nnz + m (kl + ku) + border^2 + m |tracked|).  Replaces the crossover the reference's backends run
 * behind their barrier (lp_methods/algorithms.py:50-54 -> solver_caller/gurobi.py:111-115).
 *   sx_crossover_band_dev        the starting basis is guessed from x_start: the m variables with the largest margins;
 *   sx_crossover_band_basis_dev  the starting basis is GIVEN (vbasis_in[n], cbasis_in[m]: 0 basic, -1 / -2 non-basic at
 *                                its lower / upper bound, -3 superbasic at x_start; codes of smart_crossover/output.py) --
 *                                the reference's warm-started final solve (lp_methods/algorithms.py:69-74,
 *                                lp_manager.py:79-89); a basis with too few / too many members is completed by logicals /
 *                                thinned to superbasic columns; both NULL = sx_crossover_band_dev.
 * Return SX_ERR_UNSUPPORTED when the matched basis is no band matrix the band LU takes (kl + 32 <= 1536 and
 * kl + ku + 32 <= 1980) or its border passes 16,384 rows (the caller then takes sx_simplex_crossover_dev).  Outputs and
 * result as sx_simplex_crossover_dev; result->phase1_iters counts the columns pricing added to the tableau,
 * result->max_violation covers bound violations of the basic variables AND the relative row residual of x.  Status 0 is
 * only returned for a vertex whose row residuals and basic reduced costs passed the check against A.  Blocking; arrays device. */
int sx_crossover_band_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                          const double *u, const uint8_t *row_is_lt, const double *x_start, int64_t max_iter,
                          double feas_tol, double opt_tol, double *x, double *y, int8_t *vbasis, int8_t *cbasis,
                          sx_simplex_result *result);
/* sx_crossover_band_probe_dev: would the sparse crossover take this LP from this point?  Its set-up up to the band-width
 * check only (host copies, row order, guessed basic set, matching): SX_OK or SX_ERR_UNSUPPORTED, nothing is solved. */
int sx_crossover_band_probe_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                const double *u, const uint8_t *row_is_lt, const double *x_start);
int sx_crossover_band_basis_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                                const double *u, const uint8_t *row_is_lt, const double *x_start, const int8_t *vbasis_in,
                                const int8_t *cbasis_in, int64_t max_iter, double feas_tol, double opt_tol, double *x,
                                double *y, int8_t *vbasis, int8_t *cbasis, sx_simplex_result *result);

/* Network simplex (K16n) for the re-solves of the network crossover (network_methods/net_manager.py:211-222
 * solve_subproblem -> solve_mcf / solve_ot with warm_start_basis; the reference hands these to Gurobi's /
 * CPLEX's simplex).  A must be a node-arc incidence matrix -- every column exactly one +1 (tail row) and one
 * -1 (head row) -- with l = 0 <= x <= u (u may be +inf), all rows equalities, and (vbasis_in, cbasis_in) a
 * primal feasible spanning-tree basis: vbasis 0 on m - 1 arcs that connect all rows, -1 / -2 elsewhere, cbasis
 * 0 on exactly one row (the root, whose dual is 0).  Then: primal network simplex on the tree (preorder
 * arrays, block pricing, one persistent workgroup; csrc/sx_netsimplex.hip), outputs as sx_simplex_solve_dev.
 * result->status 5 = outside that domain (not a network, not a tree, tree flows out of bounds): nothing was
 * solved and the caller falls back on sx_simplex_solve_dev, which has a phase 1.  Blocking; arrays device. */
int sx_netsimplex_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                      const double *u, const int8_t *vbasis_in, const int8_t *cbasis_in, int64_t max_iter,
                      double feas_tol, double opt_tol, double *x, double *y, int8_t *vbasis, int8_t *cbasis,
                      sx_simplex_result *result);

/* Dual network simplex (K16d) for the same re-solves, the whole GPU on one pivot (csrc/sx_netdual.hip; algorithm
 * stated in oracle/net_simplex.py).  Domain: A a node-arc incidence matrix as above, l = 0, (vbasis_in, cbasis_in)
 * a spanning tree -- it need NOT be primal feasible -- and every non-tree arc whose reduced cost has the wrong
 * sign for its bound must have a finite capacity (it is moved to its other bound, which makes the tree dual
 * feasible).  That is the situation of every column-generation round of the network crossover: the previous
 * round's optimal tree plus new arcs at a bound (network_methods/algorithms.py:109-140).  Then: dual simplex on
 * the tree in preorder arrays, leaving arc by exact dual steepest edge (violation^2 / subtree size), bound-flipping
 * ratio test over the arcs of the cut, a cooperative grid of workgroups sharing every pass.  result->status:
 * 0 optimal, 1 primal infeasible, 3 iteration limit (max_iter, or 4 m + n / 2 + 100000 when max_iter <= 0), 5 outside
 * the domain (nothing solved: take
 * sx_netsimplex_dev / sx_simplex_solve_dev); result->phase1_iters counts the arcs moved bound to bound;
 * result->warm_start_used = 2 when the tree arrays the context kept from its previous solve describe this basis
 * (a column generation's next round: same nodes, same arc numbers, more arcs) and the tree set-up was skipped.
 * Outputs as sx_simplex_solve_dev.  Blocking; arrays device. */
int sx_netdual_dev(sx_ctx *ctx, const sx_matrix *A, const double *b, const double *c, const double *l,
                   const double *u, const int8_t *vbasis_in, const int8_t *cbasis_in, int64_t max_iter,
                   double feas_tol, double *x, double *y, int8_t *vbasis, int8_t *cbasis, sx_simplex_result *result);

/* ------------------------------------------------------------------ entropic OT warm start
 * The step before the OT crossover in the reference's driver (scripts/run_network_crossover.py:95-97:
 * sinkhorn(ot.s, ot.d, ot.M, reg=10, numItermax=1000) from the third-party package POT, absent and
 * unpinned -> parity unpinned; published algorithm: Sinkhorn-Knopp scaling, oracle/sinkhorn.py).
 * a[S], b[D] marginals, M[S*D] cost (row-major), all device.  Outputs (device, any may be NULL):
 * plan[S*D] = (u .* exp(-M/reg)) .* v, u_out[S], v_out[D].  Blocking. */
typedef struct sx_sinkhorn_result {
    int64_t iters;  /* completed iterations */
    int64_t status; /* 0 iteration limit, 1 marginal violation < stop_thr, 2 numerical breakdown (previous pair kept) */
    double err;     /* last tested || plan^T 1 - b ||_2 (tested at iterations 0, 10, 20, ...) */
} sx_sinkhorn_result;
int sx_sinkhorn_dev(sx_ctx *ctx, int64_t S, int64_t D, const double *a, const double *b, const double *M,
                    double reg, int64_t max_iter, double stop_thr, double *plan, double *u_out, double *v_out,
                    sx_sinkhorn_result *result);

/* The same for B <= 16 instance pairs that share the cost matrix M (the reference's driver runs ten MNIST
 * image pairs over one 28 x 28 grid one after the other, scripts/run_network_crossover.py:95-101): a[B*S],
 * b[B*D] instance-major, a grid point an instance does not use carries mass 0 (IEEE arithmetic takes it out:
 * the instance sees exactly the problem on its own support).  With the scaling vectors side by side K^T U and
 * K V are dense (D x S)(S x B) / (S x D)(D x B) products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64);
 * instances stop independently.  Outputs (device, any may be NULL): plans[B*S*D], u_out[B*S], v_out[B*D];
 * results[B] is a HOST array.  Blocking. */
int sx_sinkhorn_batch_dev(sx_ctx *ctx, int64_t S, int64_t D, int64_t B, const double *a, const double *b,
                          const double *M, double reg, int64_t max_iter, double stop_thr, double *plans,
                          double *u_out, double *v_out, sx_sinkhorn_result *results);

/* Session: keeps the basis inverse of the last solve on the device so that the next solve of a
 * column-generation sequence (network_methods/algorithms.py:105-139: same rows, more columns, warm basis
 * = previous optimal basis) starts from it instead of re-installing the basis pivot by pivot.
 * col_ids (HOST pointer, n entries) names the structural columns by identifiers that are stable from one
 * solve to the next (e.g. the arc index in the full problem); the inverse is reused when the warm
 * basis consists of exactly the variables the session's inverse belongs to, otherwise the call
 * behaves like sx_simplex_solve_dev.  result->warm_start_used is 2 when the inverse was reused.
 * One session serves one sequence; it must not be shared between contexts or threads. */
typedef struct sx_simplex_session sx_simplex_session;
int sx_simplex_session_create(sx_ctx *ctx, sx_simplex_session **out);
int sx_simplex_session_destroy(sx_simplex_session *session);
int sx_simplex_solve_session_dev(sx_ctx *ctx, sx_simplex_session *session, const sx_matrix *A, const double *b,
                                 const double *c, const double *l, const double *u, const uint8_t *row_is_lt,
                                 const int8_t *vbasis_in, const int8_t *cbasis_in, const int64_t *col_ids,
                                 int64_t max_iter, double feas_tol, double opt_tol, double *x, double *y,
                                 int8_t *vbasis, int8_t *cbasis, sx_simplex_result *result);

#ifdef __cplusplus
}
#endif
#endif /* SXHIP_H */
