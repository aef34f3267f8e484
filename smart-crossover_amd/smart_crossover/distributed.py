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
