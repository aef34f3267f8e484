"""Worker of tests/test_dist_gloo.py: one rank of the column/row-sharded scoring pass over gloo.

The local kernels are played by the CPU oracle here (test infrastructure -- on a GPU box bench.py
runs the same exchange with the HIP kernels); what is under test is the product's partitioning and
its two collectives (smart_crossover.distributed)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

import workloads                                # noqa: E402
from oracle import lp_path as L                 # noqa: E402
from oracle import net_path as N                # noqa: E402
from smart_crossover import distributed as D    # noqa: E402


def local_pass_device(sh):
    """The same rank-local pass through the HIP kernels (SX_DIST_DEVICE=1; every rank on device 0)."""
    from smart_crossover.hip import Context
    ctx = Context(0)
    n_loc, off = sh.n_block, sh.rank * sh.n_block
    m_loc = sh.row_block.shape[0]
    r0 = sh.rank * m_loc
    dC, dR = ctx.column_shard(sh.col_block), ctx.row_shard(sh.row_block)
    d_y, d_x = ctx.to_device(sh.y), ctx.to_device(sh.x)
    s_d, code = ctx.empty(n_loc, np.float64), ctx.empty(n_loc, np.uint8)
    s_p, flag = ctx.empty(m_loc, np.float64), ctx.empty(m_loc, np.uint8)
    ctx.score_columns(dC, d_y, ctx.to_device(sh.c), ctx.to_device(sh.x[off:off + n_loc]), ctx.to_device(sh.l),
                      ctx.to_device(sh.u), 1e-3, s_d, code)
    ctx.score_rows(dR, d_x, ctx.to_device(sh.b), ctx.to_device(sh.y[r0:r0 + m_loc]), 1e-3, s_p, flag)
    res = ctx.price(dC, d_y, ctx.to_device(sh.c), ctx.to_device(np.full(n_loc, -1, dtype=np.int8)), N.RC_TOL, None)
    mn, am, bad = ctx.read_price(res)
    code_h, flag_h = code.download(), flag.download()
    counts = torch.tensor([int(ctx.where(code, 1).size), int(ctx.where(code, 2).size), int(ctx.where(flag, 0xFF).size)],
                          dtype=torch.int64)
    out = (code_h, flag_h, (float(mn), int(am), int(bad)), counts, s_d.download())
    ctx.close()
    return out


def local_pass(sh):
    """What one rank computes from its column block and its row block."""
    if os.environ.get("SX_DIST_DEVICE") == "1":
        return local_pass_device(sh)
    n_loc, off = sh.n_block, sh.rank * sh.n_block
    x_loc = sh.x[off:off + n_loc]
    s_d = sh.c - sh.col_block.T @ sh.y
    code = L.column_codes(x_loc, sh.l, sh.u, s_d, 1e-3)
    m_loc = sh.row_block.shape[0]
    r0 = sh.rank * m_loc
    s_p = sh.b - sh.row_block @ sh.x
    flag = L.row_flags(s_p, sh.y[r0:r0 + m_loc], 1e-3)
    rc = s_d.copy()                                             # vbasis == -1 everywhere: no sign flips
    j = int(np.argmin(rc))
    record = (float(rc[j]), j, int(np.count_nonzero(~(rc >= -N.RC_TOL))))
    counts = torch.tensor([int(np.count_nonzero(code & 1)), int(np.count_nonzero(code & 2)), int(flag.sum())],
                          dtype=torch.int64)
    return code, flag, record, counts, s_d


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m, n_block = 8000, 20000
    structure = os.environ.get("SX_STRUCTURE", "staircase")
    extra = {}
    if structure == "staircase-weak":        # bench.py's weak-scaling layout: 8 row regions and m rows per rank
        structure, m = "staircase", m * world
        extra = {"regions": workloads.STAIR_REGIONS * world}
    sh = workloads.lp_shard(rank, world, m=m, n_block=n_block, structure=structure, **extra)
    code, flag, record, counts, s_d = local_pass(sh)

    ex = D.Exchange(dist)
    col_blocks = D.split_even(world * n_block, world)
    assert col_blocks[rank].start == rank * n_block and col_blocks[rank].size == n_block
    price = ex.gather_price(record, [b.start for b in col_blocks])
    counts = ex.sum_counts(counts)
    slowest = ex.max_scalar(float(rank + 1))

    gathered_code = [torch.zeros(n_block, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(code.copy()), gathered_code, dst=0)
    gathered_flag = [torch.zeros(flag.size, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.from_numpy(flag.copy()), gathered_flag, dst=0)

    if rank == 0:
        # single-process statement of the same global problem
        import scipy.sparse as sp
        shards = [workloads.lp_shard(r, world, m=m, n_block=n_block, structure=structure, **extra) for r in range(world)]
        A = sp.vstack([s.row_block for s in shards]).tocsr()
        assert (sp.hstack([s.col_block for s in shards]).tocsr() != A).nnz == 0
        full = L.scoring_pass(A, np.concatenate([s.b for s in shards]), np.concatenate([s.c for s in shards]),
                              np.concatenate([s.l for s in shards]), np.concatenate([s.u for s in shards]), sh.x, sh.y)
        rc = full["s_d"]
        result = {
            "world": world,
            "codes_equal": bool(np.array_equal(np.concatenate([t.numpy() for t in gathered_code]), full["code"])),
            "flags_equal": bool(np.array_equal(np.concatenate([t.numpy() for t in gathered_flag]), full["rowflag"])),
            "counts": [int(v) for v in counts],
            "counts_want": [int(full["fix_low"].size), int(full["fix_up"].size), int(full["fixed_rows"].size)],
            "price": [price[0], price[1], price[2]],
            "price_want": [float(rc.min()), int(np.argmin(rc)), int(np.count_nonzero(~(rc >= -N.RC_TOL)))],
            "slowest": slowest,
        }
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
