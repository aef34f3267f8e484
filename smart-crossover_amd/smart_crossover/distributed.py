"""Multi-GPU layout of the scoring path: one process per GPU (``torch.distributed``; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" for CPU rehearsals).

The path shards by *columns* (SURVEY.md section 8e): rank r owns a contiguous column block of A in
CSC (K1 scoring, K10 pricing need only the replicated m-vector y) and a contiguous row block in CSR
(K2 needs the replicated input x).  No matrix data ever moves between GPUs.  The only exchanges are

  * pricing   : one 24-byte record (min reduced cost, its global column, violation count) per rank,
                all-gathered; every rank reduces the W records identically (lexicographic min on
                (value, column), sum of counts) -- the "all-reduce MIN for the global pricing minimum"
                of BASELINE.json done on a (value, index) pair, which RCCL has no MIN operator for;
  * set sizes : an all-reduce(SUM) of three int64 (|fix_low|, |fix_up|, |fixed_rows|).

Index sets stay sharded (global index = local index + block offset).  This module holds the
partitioning arithmetic and the two collectives; kernels are launched by the caller (bench.py,
ShardedLP users) through ``smart_crossover.hip``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

PRICE_RECORD_BYTES = 24       # sizeof(sx_price_result): double min_rc, int64 argmin, int64 n_violating


@dataclass
class Block:
    """Half-open range of columns (or rows) owned by one rank."""
    start: int
    stop: int

    @property
    def size(self) -> int:
        return self.stop - self.start


def split_even(n: int, world: int) -> List[Block]:
    """Contiguous blocks whose sizes differ by at most one (first n % world blocks are longer)."""
    if world < 1 or n < 0:
        raise ValueError("world must be >= 1 and n >= 0")
    base, extra = divmod(n, world)
    out, pos = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append(Block(pos, pos + size))
        pos += size
    return out


def split_by_nnz(ptr: np.ndarray, world: int) -> List[Block]:
    """Contiguous blocks of segments (columns of a CSC / rows of a CSR pointer array) holding about
    nnz / world entries each: block r ends at the first segment boundary at or after r+1 shares."""
    ptr = np.asarray(ptr, dtype=np.int64)
    nseg = ptr.size - 1
    if world < 1 or nseg < 0:
        raise ValueError("bad arguments")
    total = int(ptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        k = int(np.searchsorted(ptr, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), nseg))
    cuts.append(nseg)
    return [Block(cuts[r], cuts[r + 1]) for r in range(world)]


def pack_price(min_rc: float, argmin: int, n_violating: int) -> bytes:
    return struct.pack("<dqq", float(min_rc), int(argmin), int(n_violating))


def unpack_price(raw: bytes) -> Tuple[float, int, int]:
    return struct.unpack("<dqq", raw)


def reduce_price_records(records: Sequence[Tuple[float, int, int]], offsets: Sequence[int]) -> Tuple[float, int, int]:
    """Global pricing result from per-rank records: smallest reduced cost, ties to the smallest global
    column; ranks without a candidate (argmin < 0: empty block or all NaN) are skipped; counts add."""
    best_v, best_j, bad = float("nan"), -1, 0
    for (v, j, nb), off in zip(records, offsets):
        bad += int(nb)
        if j < 0:
            continue
        gj = int(j) + int(off)
        if best_j < 0 or v < best_v or (v == best_v and gj < best_j):
            best_v, best_j = float(v), gj
    return best_v, best_j, bad


class Exchange:
    """The two collectives of the sharded path on top of ``torch.distributed`` (any backend).
    With ``dist=None`` (single process) they degenerate to local operations."""

    def __init__(self, dist=None, device: Optional[str] = None):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.device = device

    def gather_price(self, local_record, col_offsets: Sequence[int]) -> Tuple[float, int, int]:
        """``local_record``: a uint8 tensor of 24 bytes (device or CPU) holding sx_price_result, or a
        (min_rc, argmin, n_violating) tuple."""
        import torch
        if isinstance(local_record, tuple):
            local_record = torch.frombuffer(bytearray(pack_price(*local_record)), dtype=torch.uint8)
            if self.device:
                local_record = local_record.to(self.device)
        if self.dist is None:
            return reduce_price_records([unpack_price(bytes(local_record.cpu().numpy().tobytes()))], col_offsets[:1])
        out = torch.empty(PRICE_RECORD_BYTES * self.world, dtype=torch.uint8, device=local_record.device)
        self.dist.all_gather_into_tensor(out, local_record.contiguous())
        raw = out.cpu().numpy().tobytes()
        recs = [unpack_price(raw[k * PRICE_RECORD_BYTES:(k + 1) * PRICE_RECORD_BYTES]) for k in range(self.world)]
        return reduce_price_records(recs, col_offsets)

    def sum_counts(self, counts):
        """All-reduce(SUM) of an int64 tensor (in place); returns it."""
        if self.dist is not None:
            self.dist.all_reduce(counts)
        return counts

    def max_scalar(self, value: float) -> float:
        """Max over ranks of a host scalar (bench timing)."""
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


# ======================================================================================================
# Sharded problems: the crossover host path with an LP / MCF spread over the ranks of one node
# ======================================================================================================
# What a rank computes locally is behind a small "ops" object -- ``HipOps`` below drives libsxhip.so on the
# rank's GPU with torch tensors as device vectors (so that RCCL can operate on them); the CPU rehearsal of the
# test-suite plugs in an object with the same methods -- and everything that crosses ranks is a
# ``torch.distributed`` collective issued here.  No floating-point value is ever summed across ranks except
# the one m-vector per CG iteration: scores, index sets, right-hand sides and flow indicators
# of the sharded path are bit-identical to the single-GPU path.


class HipOps:
    """Rank-local kernels: libsxhip.so on this rank's GPU.  Vectors are torch CUDA tensors; a kernel sees them
    through ``Context.wrap`` (no copies).  ``stream`` = the torch stream the context was created on, so kernels
    and collectives are ordered without host synchronisation."""

    def __init__(self, ctx, torch_module):
        self.ctx, self.torch = ctx, torch_module
        self.device = f"cuda:{ctx.device}"

    # ---- vectors
    def vec(self, host: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(host)).to(self.device)

    def empty(self, n: int, dtype):
        return self.torch.empty(int(n), dtype={np.float64: self.torch.float64, np.uint8: self.torch.uint8,
                                               np.int64: self.torch.int64}[dtype], device=self.device)

    def host(self, t) -> np.ndarray:
        return t.cpu().numpy()

    def _w(self, t):
        if t is None:
            return None
        dt = {self.torch.float64: np.float64, self.torch.uint8: np.uint8, self.torch.int64: np.int64,
              self.torch.int8: np.int8}[t.dtype]
        return self.ctx.wrap(t.data_ptr(), t.numel(), dt, owner=t)

    # ---- matrices
    def matrix(self, csr):
        return self.ctx.matrix(csr)

    def row_matrix(self, csr):
        return self.ctx.row_shard(csr)

    # ---- LP kernels
    def score_columns(self, A, y, c, x, l, u, gamma, code):
        self.ctx.score_columns(A, self._w(y), self._w(c), self._w(x), self._w(l), self._w(u), gamma, None, self._w(code))

    def score_rows(self, A_rows, x, b, y, gamma_dual, flag):
        self.ctx.score_rows(A_rows, self._w(x), self._w(b), self._w(y), gamma_dual, None, self._w(flag))

    def count(self, flags, mask: int) -> int:
        return int(self.ctx.where(self._w(flags), mask).size)

    def price(self, A, y, c, vbasis, tol) -> Tuple[float, int, int]:
        return self.ctx.read_price(self.ctx.price(A, self._w(y), self._w(c), self._w(vbasis), tol, None))

    def dual_slack(self, A, y, c):
        """c - A^T y of the own column block (the K1 walk, codes off)."""
        s_d = self.empty(A.shape[1], np.float64)
        self.ctx.score_columns(A, self._w(y), self._w(c), None, None, None, 0.0, self._w(s_d), None)
        return s_d

    def fixed_rhs(self, A_rows, code_all, u_all, l_all, b_loc, out):
        self.ctx.fixed_rhs(A_rows, self._w(code_all), self._w(u_all), self._w(l_all), self._w(b_loc), self._w(out))

    # ---- sharded CG (sx_cg_shard_*)
    def cg_open(self, A, xa, xs, c, tol):
        import ctypes as C
        h, vec = C.c_void_p(), C.c_void_p()
        keep = (self._w(xa), self._w(xs), self._w(c))
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_cg_shard_open(self.ctx.handle, A.handle, keep[0].ptr, keep[1].ptr, keep[2].ptr, None,
                                                float(tol), C.byref(h), C.byref(vec)))
        m = A.shape[0]
        # the reduce vector as a torch tensor over the library's buffer: RCCL reduces it in place
        q = self._as_tensor(vec.value, m)
        return {"h": h, "q": q, "keep": keep}

    def _as_tensor(self, ptr: int, n: int):
        class _Cai:            # __cuda_array_interface__ view of foreign device memory
            def __init__(self, p, k):
                self.__cuda_array_interface__ = {"shape": (k,), "typestr": "<f8", "data": (p, False), "version": 2}
        return self.torch.as_tensor(_Cai(ptr, n), device=self.device)

    def cg_start(self, s) -> Tuple[float, bool]:
        import ctypes as C
        from smart_crossover.hip import lib as _l
        bn, tr = C.c_double(0), C.c_int(0)
        _l.check(self.ctx._lib.sx_cg_shard_start(s["h"], C.byref(bn), C.byref(tr)))
        return float(bn.value), bool(tr.value)

    def cg_local(self, s):
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_cg_shard_local(s["h"]))

    def cg_update(self, s, k: int):
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_cg_shard_update(s["h"], int(k) & 1))

    def cg_poll(self, s) -> Tuple[bool, int]:
        import ctypes as C
        from smart_crossover.hip import lib as _l
        d, it = C.c_int(0), C.c_int64(0)
        _l.check(self.ctx._lib.sx_cg_shard_poll(s["h"], C.byref(d), C.byref(it)))
        return bool(d.value), int(it.value)

    def cg_finish(self, s) -> Tuple[float, float, int, bool]:
        import ctypes as C
        from smart_crossover.hip import lib as _l
        cols, rows, res = C.c_double(0), C.c_double(0), _l.CgResult()
        _l.check(self.ctx._lib.sx_cg_shard_finish(s["h"], None, None, C.byref(cols), C.byref(rows), C.byref(res)))
        return float(cols.value), float(rows.value), int(res.iters), bool(res.converged)

    def cg_close(self, s) -> None:
        """Releases the handle and its device block (idempotent; the caller's ``finally``)."""
        h = s.pop("h", None)
        if h is not None:
            self.ctx._lib.sx_cg_shard_close(h)

    # ---- MCF kernels
    def mcf_xhat(self, x, u, xhat, mask):
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_mcf_xhat_dev(self.ctx.handle, x.numel(), x.data_ptr(), u.data_ptr(), xhat.data_ptr(),
                                               mask.data_ptr()))

    def mcf_node_flows(self, A_rows, xhat_all, mask_all, f_inv):
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_mcf_node_flows_dev(self.ctx.handle, A_rows.handle, xhat_all.data_ptr(), mask_all.data_ptr(),
                                                     f_inv.data_ptr(), None))

    def mcf_arc_indicator(self, A_cols, xhat, mask, f_inv_all, ind):
        from smart_crossover.hip import lib as _l
        _l.check(self.ctx._lib.sx_mcf_arc_indicator_dev(self.ctx.handle, A_cols.handle, xhat.data_ptr(), mask.data_ptr(),
                                                        f_inv_all.data_ptr(), ind.data_ptr()))

    def top_k(self, key, k: int):
        """(keys, local indices) of the k largest keys, descending key, ties by descending index (the library's rule)."""
        order = self.ctx.argsort_desc(self._w(key))
        order.size, order.nbytes = int(k), int(k) * 8                      # the first k ranks only
        keys = self.ctx.gather(order, self._w(key)).download()
        return keys, order.download()


def _ops_default(dist):
    """HipOps on this rank's GPU (LOCAL_RANK), with the library context created ON torch's current stream so that
    kernels, tensor copies and RCCL collectives are ordered without host synchronisation.  torch's default stream
    has handle 0, which the library reads as "create your own": an explicit stream is made current first.  torch
    must have initialised its HIP runtime before the first library context of the process exists (the other order
    leaves torch without a device): create the ShardedLP / ShardedMCF before any other smart_crossover device call,
    or pass ``ops=HipOps(ctx, torch)`` with a context made on a torch stream."""
    import os
    import torch
    from smart_crossover.hip import Context
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    return HipOps(Context(dev, stream.cuda_stream), torch)


class ShardedLP:
    """One rank's share of a GeneralLP ``min c^T x, A x (=|<) b, l <= x <= u``: a column block of A (both layouts;
    K1 scoring, K10 pricing, the CG's two products, compaction) and a row block (row layout; K2 scoring, the
    exact right-hand side of the sub-problem).  ``x`` and ``y`` -- the inputs of the crossover -- are replicated.

    Reference arithmetic: lp_methods/algorithms.py:79-111 (get_perturb_problem), :162-193 (projector),
    lp_methods/lp_manager.py:57-64 (right-hand side)."""

    def __init__(self, lp, dist=None, ops=None, blocks: Optional[Tuple[List[Block], List[Block]]] = None):
        import scipy.sparse as sp
        self.ex = Exchange(dist, None)
        self.dist = dist
        self.ops = ops if ops is not None else _ops_default(dist)
        if hasattr(self.ops, "device"):
            self.ex.device = self.ops.device
        A = sp.csr_matrix(lp.A)
        self.m, self.n = A.shape
        world, rank = self.ex.world, self.ex.rank
        if blocks is None:
            blocks = (split_by_nnz(sp.csc_matrix(A).indptr, world), split_by_nnz(A.indptr, world))
        self.col_blocks, self.row_blocks = blocks
        self.cols, self.rows = self.col_blocks[rank], self.row_blocks[rank]
        cs, ce, rs, re = self.cols.start, self.cols.stop, self.rows.start, self.rows.stop
        self.A_cols = self.ops.matrix(A[:, cs:ce].tocsr())            # m x n_loc, both layouts
        self.A_rows = self.ops.row_matrix(A[rs:re, :].tocsr())         # m_loc x n, row layout
        v = self.ops.vec
        self.c_loc, self.l_loc, self.u_loc = v(lp.c[cs:ce]), v(lp.l[cs:ce]), v(lp.u[cs:ce])
        self.b_loc = v(lp.b[rs:re])
        self.l_all, self.u_all = v(lp.l), v(lp.u)
        self.lt = np.asarray(lp.sense) == "<"
        self.n_lt = int(np.count_nonzero(self.lt))
        # host copies the sharded simplex driver needs (entering-column fetch, replicated right-hand side and costs)
        self._A_cols_host = A[:, cs:ce].tocsc()
        self._b_host = np.asarray(lp.b, dtype=np.float64).copy()
        self._c_host = np.asarray(lp.c, dtype=np.float64).copy()

    # ---- K1 + K2 + set sizes + K10 ------------------------------------------------------------------
    def scoring_pass(self, x: np.ndarray, y: np.ndarray, gamma: float, gamma_dual: float):
        """Column codes of the own columns, row flags of the own rows, global sizes of the three index sets."""
        o = self.ops
        cs, ce, rs, re = self.cols.start, self.cols.stop, self.rows.start, self.rows.stop
        x_all, y_all = o.vec(x), o.vec(y)
        self.code_loc = o.empty(self.cols.size, np.uint8)
        self.flag_loc = o.empty(self.rows.size, np.uint8)
        o.score_columns(self.A_cols, y_all, self.c_loc, x_all[cs:ce], self.l_loc, self.u_loc, gamma, self.code_loc)
        o.score_rows(self.A_rows, x_all, self.b_loc, y_all[rs:re], gamma_dual, self.flag_loc)
        import torch
        counts = torch.tensor([o.count(self.code_loc, 1), o.count(self.code_loc, 2), o.count(self.flag_loc, 0xFF)],
                              dtype=torch.int64, device=getattr(o, "device", "cpu"))
        counts = self.ex.sum_counts(counts)
        return self.code_loc, self.flag_loc, [int(t) for t in counts.cpu()]

    def price(self, y: np.ndarray, vbasis_loc: Optional[np.ndarray] = None, tol: float = 1e-6):
        """Global pricing result (min reduced cost, its global column, violations): one 24-byte all-gather."""
        o = self.ops
        vb = o.vec(vbasis_loc.astype(np.int8)) if vbasis_loc is not None else None
        rec = o.price(self.A_cols, o.vec(y), self.c_loc, vb, tol)
        return self.ex.gather_price(rec, [b.start for b in self.col_blocks])

    # ---- K4: projector norm, columns of Y sharded -----------------------------------------------------
    def projector_norm(self, xa: np.ndarray, xs: np.ndarray, tol: float = 1e-8, maxiter: int = 1000, poll: int = 25):
        """|| (I - Y^T (Y Y^T)^+ Y) [xa .* c ; 0] || with Y = [A diag(xa), diag(xs)]: one all-reduce of an m-vector
        per CG iteration.  ``xa`` (n) and ``xs`` (m) are replicated host vectors."""
        o = self.ops
        cs, ce = self.cols.start, self.cols.stop
        import torch
        s = o.cg_open(self.A_cols, o.vec(xa[cs:ce]), o.vec(xs), self.c_loc, tol)
        try:
            self._allreduce(s["q"])
            bnorm, trivial = o.cg_start(s)
            it = 0
            done = self._agreed(trivial)
            while not done and it < maxiter:
                upto = min(it + poll, maxiter)
                for k in range(it, upto):
                    o.cg_local(s)
                    self._allreduce(s["q"])
                    o.cg_update(s, k)
                it = upto
                # every all_reduce above is a collective: the ranks must leave the loop together, so the flag
                # each rank reads from its own device is itself reduced (MAX: stop as soon as any rank stops)
                done = self._agreed(o.cg_poll(s)[0])
            cols, rows, iters, converged = o.cg_finish(s)
        finally:
            close = getattr(o, "cg_close", None)
            if close is not None:
                close(s)
        t = torch.tensor([cols], dtype=torch.float64, device=getattr(o, "device", "cpu"))
        self._allreduce(t)
        return float(np.sqrt(float(t.item()) + rows)), iters, converged

    def _allreduce(self, t):
        if self.dist is not None:
            self.dist.all_reduce(t)

    def _agreed(self, flag) -> bool:
        """The same yes/no on every rank: MAX over the ranks of a one-element int tensor."""
        if self.dist is None:
            return bool(flag)
        import torch
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=getattr(self.ops, "device", "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def _allgather(self, t, sizes: Sequence[int]):
        """Concatenation of every rank's 1-D tensor (blocks may differ in length)."""
        import torch
        if self.dist is None:
            return t
        # collectives want equal shapes: pad the local block to the longest one, gather, cut the padding out
        longest = int(max(sizes))
        mine = torch.zeros(longest, dtype=t.dtype, device=t.device)
        mine[:t.numel()] = t
        out = torch.empty(longest * len(sizes), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, mine)
        return torch.cat([out[r * longest:r * longest + int(k)] for r, k in enumerate(sizes)])

    # ---- K6: right-hand side of the sub-problem, exact --------------------------------------------------
    # ---- K16 over column shards: the pricing minimum is global, everything else replicated ------------------
    def simplex_price(self, y: np.ndarray, vbasis_loc: np.ndarray, tol: float = 1e-6):
        """One pricing round of a column-sharded simplex pivot (reference: the pricing the re-solves hide inside the
        solver, globalised as net_manager.py:293-319 globalises its optimality test): the rank-local K10 walk over the
        own column block -- reduced cost c_j - a_j^T y, sign flipped at an upper bound, basic columns price at 0 --
        leaves one 24-byte record (value, column, violations); ONE all-gather later every rank holds the same global
        entering column (most negative value, ties to the smallest global column)."""
        return self.price(y, vbasis_loc, tol)

    def primal_simplex(self, max_iter: int = 100000, opt_tol: float = 1e-9, trace: Optional[list] = None):
        """Bounded primal simplex (Dantzig pricing, textbook ratio test, ties to the smallest index) on
        ``min c^T x, A x <= b, l <= x <= u`` from the slack basis, which must be feasible (b - A l >= 0, every row
        '<', l finite).  What is SHARDED is what scales with the columns: pricing (``simplex_price``: K10 on the
        rank's block + the 24-byte all-gather) and the fetch of the entering column (its owner broadcasts it: 8 m
        bytes); FTRAN, ratio test and the basis inverse (dense, on the host: this driver is for sub-problems of a
        few thousand rows) are replicated and run identically on every rank, so every rank makes the same pivots.
        The single-GPU solver with the basis on the device is sx_simplex_solve_dev.  Returns x (this rank's block),
        y, the global pivot sequence [(entering variable, leaving position)] and the status."""
        import torch
        o = self.ops
        m, cs, ce = self.m, self.cols.start, self.cols.stop
        n_loc = ce - cs
        if not bool(np.all(self.lt)):
            raise ValueError("primal_simplex starts from the slack basis: every row must be '<'")
        l_loc, u_loc, c_loc = (np.asarray(o.host(t), dtype=np.float64) for t in (self.l_loc, self.u_loc, self.c_loc))
        l_all, u_all = np.asarray(o.host(self.l_all)), np.asarray(o.host(self.u_all))
        if not np.all(np.isfinite(l_all)):
            raise ValueError("primal_simplex needs finite lower bounds")
        A_loc = self._host_cols()                              # this rank's columns on the host (entering-column fetch)
        # b - A l: every rank's share of A l summed over the ranks (one m-vector all-reduce, once)
        t = torch.from_numpy(np.ascontiguousarray(A_loc @ l_loc))
        t = t.to(getattr(o, "device", "cpu"))
        self._allreduce(t)
        b_all = self._b_all()
        xB = b_all - t.cpu().numpy()                           # slacks basic
        if xB.min() < -1e-12:
            raise ValueError("the slack basis is not feasible")
        n = self.n
        head = np.arange(n, n + m)                             # variable at every basis position (logical of row i: n + i)
        Binv = np.eye(m)
        vb_loc = np.full(n_loc, -1, dtype=np.int8)             # -1 at lower, -2 at upper, 0 basic
        x_loc = l_loc.copy()
        cB = np.zeros(m)
        lB, uB = np.zeros(m), np.full(m, np.inf)
        logical_nb = np.zeros(m, dtype=bool)                   # logicals that left the basis (non-basic at 0)
        pivots, status = [], "ITERATION_LIMIT"
        owners = [b.start for b in self.col_blocks]
        for it in range(int(max_iter)):
            y = Binv.T @ cB
            rc_min, gcol, _ = self.simplex_price(y, vb_loc, opt_tol)          # <- the exchange of this pivot
            # non-basic logicals (replicated): reduced cost -y_i at 0
            li = -1
            if logical_nb.any():
                d = np.where(logical_nb, -y, np.inf)
                li = int(np.argmin(d))
                if not (d[li] < -opt_tol):
                    li = -1
            take_logical = li >= 0 and (gcol < 0 or not (rc_min < -opt_tol) or -y[li] < rc_min)
            if not take_logical and (gcol < 0 or not (rc_min < -opt_tol)):
                status = "OPTIMAL"
                break
            if take_logical:
                q, a_q, direction, own_range = n + li, np.zeros(m), 1.0, np.inf
                a_q[li] = 1.0
            else:
                q = int(gcol)
                owner = int(np.searchsorted(owners, q, side="right") - 1)
                col = torch.zeros(m + 2, dtype=torch.float64)
                if owner == self.ex.rank:
                    j = q - cs
                    col[:m] = torch.from_numpy(np.asarray(A_loc[:, j].todense()).ravel())
                    col[m] = -1.0 if vb_loc[j] == -2 else 1.0
                    col[m + 1] = u_loc[j] - l_loc[j]
                col = col.to(getattr(o, "device", "cpu"))
                if self.dist is not None:
                    self.dist.broadcast(col, src=owner)
                colh = col.cpu().numpy()
                a_q, direction, own_range = colh[:m], float(colh[m]), float(colh[m + 1])
            alpha = Binv @ a_q
            rate = -direction * alpha                           # change of x_B per unit step
            with np.errstate(divide="ignore", invalid="ignore"):
                t_lo = np.where(rate < -1e-11, (xB - lB) / -rate, np.inf)
                t_up = np.where((rate > 1e-11) & np.isfinite(uB), (uB - xB) / rate, np.inf)
            tt = np.minimum(t_lo, t_up)
            r = int(np.argmin(tt))
            theta = max(float(tt[r]), 0.0)
            if own_range <= theta:                              # the entering column reaches its other bound first
                if not np.isfinite(own_range):
                    status = "UNBOUNDED"
                    break
                xB = xB + rate * own_range
                if not take_logical and owner == self.ex.rank:
                    j = q - cs
                    vb_loc[j] = -1 if vb_loc[j] == -2 else -2
                    x_loc[j] = u_loc[j] if vb_loc[j] == -2 else l_loc[j]
                pivots.append((q, -1))
                continue
            if not np.isfinite(theta):
                status = "UNBOUNDED"
                break
            hit_upper = t_up[r] <= t_lo[r]
            xB = xB + rate * theta
            vout = int(head[r])
            # leaving variable -> non-basic at the bound it hit
            if vout >= n:
                logical_nb[vout - n] = True
            elif cs <= vout < ce:
                vb_loc[vout - cs] = -2 if hit_upper else -1
                x_loc[vout - cs] = u_loc[vout - cs] if hit_upper else l_loc[vout - cs]
            # entering variable -> position r
            if take_logical:
                logical_nb[li] = False
                x_new, lo_q, up_q, c_q = theta, 0.0, np.inf, 0.0
            else:
                lo_q, up_q = float(l_all[q]), float(u_all[q])
                x_new = (up_q - theta) if direction < 0 else (lo_q + theta)
                c_q = self._cost_of(q)
                if owner == self.ex.rank:
                    vb_loc[q - cs] = 0
            head[r], xB[r], lB[r], uB[r], cB[r] = q, x_new, lo_q, up_q, c_q
            piv = alpha[r]
            row = Binv[r] / piv
            Binv = Binv - np.outer(alpha, row)
            Binv[r] = row
            pivots.append((q, r))
        if trace is not None:
            trace.extend(pivots)
        for p in range(m):                                      # basic structurals of this block: their values
            if head[p] < n and cs <= head[p] < ce:
                x_loc[head[p] - cs] = xB[p]
        return x_loc, Binv.T @ cB, pivots, status

    # ---- the re-solve over column shards: restricted LP replicated, pricing of everything else rank-local ------------
    def restricted_resolve(self, start_cols, solver: str = "HIP", x_start: Optional[np.ndarray] = None,
                           y_start: Optional[np.ndarray] = None, first_method: str = "default", batch: int = 2048,
                           opt_tol: float = 1e-6, max_rounds: int = 200, trace: Optional[list] = None, settings=None,
                           max_seconds: Optional[float] = None):
        """The LP re-solve with the COLUMNS sharded (north_star: "columns shard naturally ... all-reduce for the global pricing
        minimum"; the reference's last step, lp_methods/algorithms.py:69-74, prices all columns inside its solver).  What
        a vertex needs is m columns out of n: the restricted LP over a column set R (``start_cols`` at first: the columns
        the crossover left free) is REPLICATED and solved by ``solver`` on every rank from the same start -- the device
        solvers are deterministic, so every rank holds the same vertex and basis --; what scales with n, the pricing of
        the columns outside R with the duals of that vertex, is rank-local (K1 walk over the rank's column block, the
        ``batch`` largest violations kept); ONE all-gather of (|reduced cost|, column) records later every rank knows the
        same ``batch`` entering columns, their owners hand over the entries (one all-gather of the columns themselves),
        R grows and the restricted LP is solved again FROM THE BASIS IT HAD (``sx_crossover_band_basis_dev`` factors that
        very basis; new columns non-basic at a bound).  Ends when no column outside R prices out: the vertex is optimal
        for the whole LP.  Returns (x over R, y, R, basis over R, status, rounds); ``trace`` collects the columns each
        round added.  Factorisation and tableau are replicated ("replicas only" for those).  ``max_seconds``:
        stop before the next re-solve once ANY rank has spent that long (the flags travel with the record counts, so all
        ranks leave in the same round): status "TIME_LIMIT", the last vertex returned."""
        import time
        import scipy.sparse as sp
        import torch
        from smart_crossover.formats import GeneralLP
        from smart_crossover.output import Basis
        from smart_crossover.solver_caller.caller import SolverSettings
        from smart_crossover.solver_caller.solving import solve_lp
        o = self.ops
        m, cs, ce = self.m, self.cols.start, self.cols.stop
        world = self.ex.world
        dev = getattr(o, "device", "cpu")
        settings = settings if settings is not None else SolverSettings(presolve="on", log_console=0, optimalityTol=opt_tol)
        A_loc = self._host_cols()
        c_all = self._c_host
        l_all, u_all = np.asarray(o.host(self.l_all), dtype=np.float64), np.asarray(o.host(self.u_all), dtype=np.float64)
        l_loc, u_loc = l_all[cs:ce], u_all[cs:ce]
        sense = np.where(self.lt, "<", "=")

        def fetch(ids: np.ndarray):
            """m x len(ids) block of the columns ``ids`` (ascending, the same on every rank), each from its owner."""
            mine = ids[(ids >= cs) & (ids < ce)]
            piece = sp.csc_matrix(A_loc[:, mine - cs])
            payload = (mine.size, piece.indptr.astype(np.int64), piece.indices.astype(np.int32), piece.data.astype(np.float64))
            parts = [payload]
            if self.dist is not None:
                parts = [None] * world
                self.dist.all_gather_object(parts, payload)
            blocks = [sp.csc_matrix((d, i, p), shape=(m, k)) for (k, p, i, d) in parts if k > 0]     # rank order = column order
            return sp.hstack(blocks, format="csc") if blocks else sp.csc_matrix((m, 0))

        in_R = np.zeros(self.n, dtype=bool)
        in_R[np.asarray(start_cols, dtype=np.int64)] = True
        R = np.flatnonzero(in_R)
        A_R = fetch(R)
        basis, x_R, y, status, rounds = None, None, None, "UNKNOWN", 0
        t_begin = time.perf_counter()
        for rounds in range(1, max_rounds + 1):
            lp_R = GeneralLP(sp.csr_matrix(A_R), self._b_host, c_all[R], l_all[R], u_all[R], sense)
            # a failure on ONE rank (device memory, a refused basis ...) must not leave the others waiting in the exchange
            # below: it is caught, travels with the record counts, and every rank leaves in this round
            err = None
            try:
                if basis is None:
                    ws = (np.clip(np.asarray(x_start)[R], l_all[R], u_all[R]), y_start) if x_start is not None and y_start is not None else None
                    out = solve_lp(lp_R, solver, first_method, settings, warm_start_solution=ws)
                else:
                    out = solve_lp(lp_R, solver, "primal_simplex", settings, warm_start_basis=basis, warm_start_solution=(x_R, y))
                status = out.status
            except Exception as exc:        # noqa: BLE001 -- re-raised below, after the exchange
                err, status = exc, "FAILED"
            ids_loc, score_loc = np.zeros(0, dtype=np.int64), np.zeros(0)
            if status == "OPTIMAL":
                x_R, y, basis = np.asarray(out.x, dtype=np.float64), np.asarray(out.y, dtype=np.float64), out.basis
                # ---- rank-local pricing of the own columns outside R: non-basic at the lower bound (at the upper one when
                #      there is no lower), reduced cost of the wrong sign = may enter
                rc = np.asarray(o.host(o.dual_slack(self.A_cols, o.vec(y), self.c_loc)), dtype=np.float64)
                at_up = ~np.isfinite(l_loc) & np.isfinite(u_loc)
                free = ~np.isfinite(l_loc) & ~np.isfinite(u_loc)
                wrong = np.where(free, np.abs(rc) > opt_tol, np.where(at_up, rc > opt_tol, rc < -opt_tol))
                cand = np.flatnonzero(wrong & ~in_R[cs:ce] & (l_loc != u_loc))
                order = np.lexsort((cand, -np.abs(rc[cand])))[:batch]        # largest violation first, ties to the smaller column
                ids_loc, score_loc = (cand[order] + cs).astype(np.int64), np.abs(rc[cand[order]])
            # ---- ONE exchange of the records: every rank ends with the same global list
            late = int(max_seconds is not None and time.perf_counter() - t_begin > max_seconds)
            flags = torch.tensor([ids_loc.size, late, int(status != "OPTIMAL")], dtype=torch.int64, device=dev)
            cnt = self._allgather(flags, [3] * world).cpu().numpy().reshape(world, 3)
            if cnt[:, 2].any():              # not optimal somewhere (the same everywhere when the replicas agree): all leave
                if err is not None:
                    raise err
                if status == "OPTIMAL":
                    raise RuntimeError(f"restricted_resolve: rank(s) {np.flatnonzero(cnt[:, 2]).tolist()} left round {rounds} without an optimal vertex")
                break
            sizes = [int(v) for v in cnt[:, 0]]
            ids_all = self._allgather(torch.from_numpy(ids_loc).to(dev), sizes).cpu().numpy()
            score_all = self._allgather(torch.from_numpy(np.ascontiguousarray(score_loc)).to(dev), sizes).cpu().numpy()
            if ids_all.size == 0:
                break                                                        # optimal for the whole LP
            if cnt[:, 1].any():
                status = "TIME_LIMIT"
                break
            pick = np.lexsort((ids_all, -score_all))[:batch]
            add = np.sort(ids_all[pick])
            if trace is not None:
                trace.append([int(j) for j in add])
            # ---- the entering columns from their owners; R stays in ascending order
            A_add = fetch(add)
            in_R[add] = True
            R_new = np.flatnonzero(in_R)
            where_old, where_add = np.searchsorted(R_new, R), np.searchsorted(R_new, add)
            merged = sp.hstack([A_R, A_add], format="csc")
            perm = np.empty(R_new.size, dtype=np.int64)
            perm[where_old] = np.arange(R.size)
            perm[where_add] = R.size + np.arange(add.size)
            A_R = merged[:, perm]
            vb = np.empty(R_new.size, dtype=np.int64)
            vb[where_old] = basis.vbasis
            up_new = ~np.isfinite(l_all[add]) & np.isfinite(u_all[add])
            vb[where_add] = np.where(up_new, -2, -1)
            x_new = np.empty(R_new.size)
            x_new[where_old] = x_R
            bound = np.where(up_new, u_all[add], l_all[add])
            x_new[where_add] = np.where(np.isfinite(bound), bound, 0.0)
            basis, x_R, R = Basis(vb, basis.cbasis), x_new, R_new
        return x_R, y, R, basis, status, rounds

    def _host_cols(self):
        import scipy.sparse as sp
        if getattr(self, "_A_cols_host", None) is None:
            raise ValueError("ShardedLP was built without keep_host=True")
        return sp.csc_matrix(self._A_cols_host)

    def _b_all(self) -> np.ndarray:
        return self._b_host

    def _cost_of(self, q: int) -> float:
        return float(self._c_host[q])

    def sub_problem_rhs(self):
        """b - A[:, fix_up] u - A[:, fix_low] l for the own rows after ``scoring_pass`` -- every row is summed by
        the rank that owns it, over all columns, so the result equals the single-GPU one bit for bit; the codes
        of the other ranks' columns arrive by one all-gather of a byte per column."""
        o = self.ops
        code_all = self._allgather(self.code_loc, [b.size for b in self.col_blocks])
        out = o.empty(self.rows.size, np.float64)
        o.fixed_rhs(self.A_rows, code_all, self.u_all, self.l_all, self.b_loc, out)
        return out


class ShardedMCF:
    """Arcs of a min-cost-flow problem sharded over ranks (BASELINE config 4): flow indicators and their ranking
    (network_methods/net_manager.py:156-184).  A rank owns a block of arcs (their x, u and incidence columns) and a
    block of nodes (their rows over all arcs)."""

    def __init__(self, mcf, dist=None, ops=None):
        import scipy.sparse as sp
        self.ex = Exchange(dist, None)
        self.dist = dist
        self.ops = ops if ops is not None else _ops_default(dist)
        if hasattr(self.ops, "device"):
            self.ex.device = self.ops.device
        A = sp.csr_matrix(mcf.A)
        self.V, self.E = A.shape
        world, rank = self.ex.world, self.ex.rank
        self.arc_blocks, self.node_blocks = split_even(self.E, world), split_by_nnz(A.indptr, world)
        self.arcs, self.nodes = self.arc_blocks[rank], self.node_blocks[rank]
        self.A_cols = self.ops.matrix(A[:, self.arcs.start:self.arcs.stop].tocsr())
        self.A_rows = self.ops.row_matrix(A[self.nodes.start:self.nodes.stop, :].tocsr())
        self.u_loc = self.ops.vec(np.asarray(mcf.u, dtype=np.float64)[self.arcs.start:self.arcs.stop])

    _allgather = ShardedLP._allgather

    def flow_indicators(self, x_loc: np.ndarray):
        """Indicators of the own arcs from the own slice of the inexact flow."""
        o = self.ops
        x = o.vec(np.asarray(x_loc, dtype=np.float64))
        xhat, mask = o.empty(self.arcs.size, np.float64), o.empty(self.arcs.size, np.uint8)
        o.mcf_xhat(x, self.u_loc, xhat, mask)
        sizes = [b.size for b in self.arc_blocks]
        xhat_all, mask_all = self._allgather(xhat, sizes), self._allgather(mask, sizes)
        f_inv = o.empty(self.nodes.size, np.float64)
        o.mcf_node_flows(self.A_rows, xhat_all, mask_all, f_inv)
        f_inv_all = self._allgather(f_inv, [b.size for b in self.node_blocks])
        self.ind_loc = o.empty(self.arcs.size, np.float64)
        o.mcf_arc_indicator(self.A_cols, xhat, mask, f_inv_all, self.ind_loc)
        return self.ind_loc

    def top_arcs(self, k: int) -> np.ndarray:
        """Global indices of the k arcs with the largest indicators, ranked as the single-GPU queue ranks them
        (descending indicator, ties by descending arc index): each rank ranks its own arcs, the W candidate lists
        of length <= k are all-gathered and merged -- only queue *prefixes* are ever consumed
        (network_methods/algorithms.py:114-115), so no global sort is needed."""
        import torch
        o = self.ops
        kk = min(int(k), self.arcs.size)
        keys, idx = o.top_k(self.ind_loc, kk)
        rec = np.zeros((int(k), 2), dtype=np.float64)
        rec[:, 0] = -np.inf
        rec[:kk, 0] = keys
        rec[:kk, 1] = (idx + self.arcs.start).astype(np.float64)       # arc indices are exact in a double (< 2^53)
        t = torch.from_numpy(rec.reshape(-1))
        if self.dist is not None:
            t = t.to(getattr(o, "device", "cpu"))
            parts = [torch.empty_like(t) for _ in range(self.ex.world)]
            self.dist.all_gather(parts, t)
            allrec = np.concatenate([p.cpu().numpy().reshape(-1, 2) for p in parts])
        else:
            allrec = rec
        allrec = allrec[allrec[:, 0] > -np.inf]
        order = np.lexsort((-allrec[:, 1], -allrec[:, 0]))               # key descending, then index descending
        return allrec[order[:int(k)], 1].astype(np.int64)
