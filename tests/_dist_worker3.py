"""Worker of the gloo rehearsal of the column-sharded simplex (TEST INFRASTRUCTURE): ShardedLP.primal_simplex over
`world` ranks, rank-local pricing played by the CPU oracle; rank 0 writes the pivot sequence and the optimum."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "smart-crossover_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from smart_crossover import distributed as D   # noqa: E402
from smart_crossover.formats import GeneralLP  # noqa: E402
from _dist_ops import OracleOps                # noqa: E402


def problem(m=60, n=400, seed=4):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=0.08, random_state=np.random.RandomState(seed), data_rvs=lambda k: rng.uniform(0.1, 1.0, k)).tocsr()
    b = rng.uniform(5.0, 10.0, m)
    c = -rng.uniform(0.1, 1.0, n)                      # every column wants to grow: plenty of pivots
    l = np.zeros(n)
    u = rng.uniform(0.5, 3.0, n)
    return GeneralLP(A, b, c, l, u, np.array(["<"] * m))


def main():
    out_path = sys.argv[1]
    single = os.environ.get("WORLD_SIZE", "1") == "1"
    if not single:
        dist.init_process_group("gloo")
    lp = problem()
    sh = D.ShardedLP(lp, None if single else dist, OracleOps())
    x_loc, y, pivots, status = sh.primal_simplex(max_iter=20000)
    obj_loc = float(lp.c[sh.cols.start:sh.cols.stop] @ x_loc)
    if single:
        res = {"world": 1, "status": status, "pivots": pivots, "obj": obj_loc}
    else:
        import torch
        t = torch.tensor([obj_loc], dtype=torch.float64)
        dist.all_reduce(t)
        res = {"world": dist.get_world_size(), "status": status, "pivots": pivots, "obj": float(t.item())}
    if single or dist.get_rank() == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    if not single:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
