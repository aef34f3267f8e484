"""Worker of tests/test_dist_gloo.py::test_sharded_*: one rank of the product's ShardedLP / ShardedMCF
(smart_crossover/distributed.py) over gloo.  SX_DIST_OPS=oracle (default): rank-local kernels played by the CPU
oracle (tests/_dist_ops.py); SX_DIST_OPS=hip: the real kernels, every rank on GPU 0 (tests/test_gpu_dist.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "smart-crossover_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch.distributed as dist                # noqa: E402

import workloads                                # noqa: E402
from oracle import lp_path as L                 # noqa: E402
from oracle import net_path as N                # noqa: E402
from smart_crossover import distributed as D    # noqa: E402
from smart_crossover.formats import GeneralLP, MinCostFlow  # noqa: E402


def make_ops():
    if os.environ.get("SX_DIST_OPS", "oracle") == "hip":
        import torch
        from smart_crossover.hip import Context
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        return D.HipOps(Context(0, stream.cuda_stream), torch)
    from _dist_ops import OracleOps
    return OracleOps()


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ops = make_ops()
    res = {"world": world}

    # ------------------------------------------------------------------ LP: scoring, pricing, CG, exact rhs
    inst = workloads.sparse_lp(900, 4000, 6, seed=17, stratified=False, frac_upper=0.3)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    slp = D.ShardedLP(lp, dist, ops)
    code_loc, flag_loc, counts = slp.scoring_pass(inst.x, inst.y, 1e-3, 1e-3)
    price = slp.price(inst.y, None, N.RC_TOL)
    # scales of a well-conditioned projector (the interior point's own scales span six decades and leave the
    # reference's CG unconverged at its 1000-iteration cap, where results depend on rounding order)
    rng = np.random.default_rng(23)
    xr = rng.uniform(0.1, 1.0, inst.x.size)
    xs = np.where(inst.sense == "<", rng.uniform(0.1, 1.0, inst.b.size), 0.0)
    norm, iters, conv = slp.projector_norm(xr, xs, 1e-8, 1000, poll=10)
    rhs_loc = ops.host(slp.sub_problem_rhs())
    gathered = [None] * world
    dist.all_gather_object(gathered, (slp.cols.start, ops.host(code_loc).copy(), slp.rows.start, ops.host(flag_loc).copy(),
                                     rhs_loc.copy()))
    if rank == 0:
        full = L.scoring_pass(inst.A, inst.b, inst.c, inst.l, inst.u, inst.x, inst.y)
        code = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])])
        flag = np.concatenate([g[3] for g in sorted(gathered, key=lambda g: g[2])])
        rhs = np.concatenate([g[4] for g in sorted(gathered, key=lambda g: g[2])])
        rc = full["s_d"]
        sub = L.sub_problem(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, full["fix_low"], full["fix_up"], full["fixed_rows"])
        c_std = np.concatenate([inst.c, np.zeros(int(np.count_nonzero(inst.sense == "<")))])
        xx = np.concatenate([xr, xs[inst.sense == "<"]])
        proj, it_ref = L.projector_matrix_free(inst.A, inst.sense, xx, c_std, 1e-8, 1000)
        res["lp"] = {
            "codes_equal": bool(np.array_equal(code, full["code"])), "flags_equal": bool(np.array_equal(flag, full["rowflag"])),
            "counts": counts, "counts_want": [int(full["fix_low"].size), int(full["fix_up"].size), int(full["fixed_rows"].size)],
            "price": [price[0], price[1], price[2]],
            "price_want": [float(rc.min()), int(np.argmin(rc)), int(np.count_nonzero(~(rc >= -N.RC_TOL)))],
            "rhs_bits_equal": bool(np.array_equal(rhs.view(np.uint64), np.asarray(sub["b"]).view(np.uint64))),
            "proj_norm": norm, "proj_norm_want": float(np.linalg.norm(proj)), "cg_iters": iters, "cg_iters_want": int(it_ref),
            "cg_converged": bool(conv), "blocks": [[b.start, b.stop] for b in slp.col_blocks]}

    # ------------------------------------------------------------------ MCF: arcs over ranks
    mi = workloads.mcf(600, 5000, seed=29)
    mcf = MinCostFlow(A=mi.A, b=mi.b, c=mi.c, u=mi.u)
    smcf = D.ShardedMCF(mcf, dist, ops)
    ind_loc = ops.host(smcf.flow_indicators(mi.x[smcf.arcs.start:smcf.arcs.stop]))
    top = smcf.top_arcs(700)
    gathered = [None] * world
    dist.all_gather_object(gathered, (smcf.arcs.start, ind_loc.copy()))
    if rank == 0:
        ind = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])])
        want, _ = N.mcf_flow_indicators(mi.A, mi.x, mi.u)
        queue = N.rank_desc(want)
        res["mcf"] = {"ind_bits_equal": bool(np.array_equal(ind.view(np.uint64), want.view(np.uint64))),
                      "top_equal": bool(np.array_equal(top, queue[:700])), "top_len": int(top.size)}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
