"""GPU: the N > 1 path with the real kernels -- two ranks (both on device 0, exchange over gloo on
127.0.0.1) run their column / row blocks through libsxhip.so; the merged result must equal the
single-process oracle on the assembled global LP.  The RCCL transport itself is exercised by
``bench.py --gpus N`` on a multi-GPU node; here it is the sharded kernels plus the product's
partitioning and reduction logic."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("structure", ["staircase-weak", "uniform"])
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, structure):
    out = tmp_path / "result.json"
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SX_STRUCTURE=structure, SX_DIST_DEVICE="1", OMP_NUM_THREADS="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(out)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            text, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(text.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = json.loads(out.read_text())
    assert res["world"] == 2
    assert res["codes_equal"] and res["flags_equal"]
    assert res["counts"] == res["counts_want"] and sum(res["counts"]) > 0
    assert res["price"] == res["price_want"]


def test_sharded_lp_and_mcf_with_the_real_kernels(tmp_path):
    """The product's ShardedLP / ShardedMCF (smart_crossover/distributed.py) with HipOps: two ranks, both on device
    0, collectives over gloo -- K1/K2/K10, the sharded projector CG (sx_cg_shard_*: one m-vector all-reduce per
    iteration), the exact sharded right-hand side, arcs-over-ranks flow indicators (sx_mcf_*_dev) and the merged
    top-k -- against the single-process oracle."""
    sys.path.insert(0, HERE)
    from test_dist_gloo import check_sharded_results, run_workers
    res = run_workers("_dist_worker2.py", tmp_path / "res.json", 2,
                      {"SX_DIST_OPS": "hip", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "LOCAL_RANK": "0"})
    assert res["world"] == 2
    check_sharded_results(res)


def test_bench_n_gt_1_code_path_over_rccl_with_one_rank():
    """``bench.py``'s N > 1 path over RCCL itself, as far as a box with one GPU allows: one rank under
    torch.distributed.run with SX_BENCH_REHEARSAL=rccl1 -- process group "nccl" on the device, kernels and
    collectives on one stream, the all-gather of the 48-byte records, barrier, MAX all-reduce of the time and the
    sharded CG's m-vector all-reduce.  The step's result must be the single-process one."""
    root = os.path.dirname(HERE)
    common = ["--gpus", "1", "--workload", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-crossover",
              "--no-uniform"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, env=env, capture_output=True,
                           timeout=600, cwd=root)
    assert plain.returncode == 0, plain.stderr.decode(errors="replace")[-2000:]
    env["SX_BENCH_REHEARSAL"] = "rccl1"
    dist = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py")]
                          + common + ["--cg-iters", "5", "--resolve-rows", "4000"], env=env, capture_output=True, timeout=600, cwd=root)
    assert dist.returncode == 0, dist.stderr.decode(errors="replace")[-2000:]
    a = json.loads(plain.stdout.decode().strip().splitlines()[-1])
    b = json.loads(dist.stdout.decode().strip().splitlines()[-1])
    assert a["result"] == b["result"] and b["result"]["fix_low"] > 0
    assert b["n_gpus"] == 1 and b["sharded_cg"]["iterations"] == 5
    # the column-sharded re-solve leg went over RCCL too (record counts, records, entering columns) and reached the optimum
    rs = b["sharded_resolve"]
    assert "failed" not in rs, rs
    assert rs["status"] == "OPTIMAL" and rs["rounds"] >= 2 and sum(rs["columns_added"]) > 0


def test_column_sharded_re_solve_on_the_device(tmp_path):
    """ShardedLP.restricted_resolve with the real kernels and the device solvers: two ranks on GPU 0, the restricted LP
    (2,400 rows: the sparse crossover's size) replicated and re-solved from its own basis each round
    (sx_crossover_band_basis_dev), pricing of the other columns by the K1 walk over each rank's block, records and entering
    columns exchanged over gloo.  Same rounds, same columns, same basis as one process; HiGHS' optimum of the whole LP."""
    import importlib.util
    import numpy as np
    from scipy.optimize import linprog
    sys.path.insert(0, HERE)
    from test_dist_gloo import run_workers
    env = {"SX_DIST_OPS": "hip", "SX_TEST_SOLVER": "HIP", "SX_TEST_SIZE": "2400,24000", "SX_TEST_BATCH": "128",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0", "LOCAL_RANK": "0"}
    single = run_workers("_dist_worker4.py", tmp_path / "w1.json", 1, env)
    assert single["status"] == "OPTIMAL" and single["rounds"] >= 2
    res = run_workers("_dist_worker4.py", tmp_path / "w2.json", 2, env)
    assert res["world"] == 2 and res["status"] == "OPTIMAL"
    assert res["trace"] == single["trace"] and res["R"] == single["R"] and res["basic"] == single["basic"]
    spec = importlib.util.spec_from_file_location("w4", os.path.join(HERE, "_dist_worker4.py"))
    w4 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w4)
    lp, start = w4.problem(2400, 24000)
    lt = np.asarray(lp.sense) == "<"
    ref = linprog(lp.c, A_ub=lp.A[lt], b_ub=lp.b[lt], A_eq=lp.A[~lt], b_eq=lp.b[~lt], bounds=np.c_[lp.l, lp.u], method="highs")
    assert ref.status == 0 and res["obj"] == pytest.approx(ref.fun, rel=1e-7, abs=1e-7)
