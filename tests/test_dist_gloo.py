"""CPU: the N > 1 path (column / row blocks, pricing all-gather, set-size all-reduce) rehearsed with
two gloo ranks on 127.0.0.1 and compared with the single-process result."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("structure", ["staircase", "staircase-weak", "uniform"])
def test_two_rank_scoring_matches_single_process(tmp_path, structure):
    out = tmp_path / "result.json"
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SX_STRUCTURE=structure, OMP_NUM_THREADS="2")
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
    assert res["slowest"] == 2.0


def run_workers(script, out, world=2, extra_env=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, script), str(out)], env=env,
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
    return json.loads(out.read_text())


def check_sharded_results(res):
    lp, mcf = res["lp"], res["mcf"]
    # scoring, pricing, set sizes: identical to the single-process pass
    assert lp["codes_equal"] and lp["flags_equal"]
    assert lp["counts"] == lp["counts_want"] and sum(lp["counts"]) > 0
    assert lp["price"] == lp["price_want"]
    # right-hand side of the sub-problem: every row summed by its owner -> bit-identical
    assert lp["rhs_bits_equal"]
    # projector CG: the one quantity summed across ranks (an m-vector per iteration) -> equal to the stated
    # tolerance, same iteration count up to rounding of the stop test
    assert lp["cg_converged"]
    assert lp["proj_norm"] == pytest.approx(lp["proj_norm_want"], rel=1e-9)
    assert abs(lp["cg_iters"] - lp["cg_iters_want"]) <= 2
    # MCF: indicators bit-identical, ranking identical under the library's tie rule
    assert mcf["ind_bits_equal"] and mcf["top_equal"] and mcf["top_len"] == 700


def test_sharded_lp_and_mcf_match_the_single_process_oracle(tmp_path):
    """The product's ShardedLP / ShardedMCF over two gloo ranks (rank-local kernels played by the CPU oracle):
    K1/K2/K10 + set sizes, the sharded projector CG (one m-vector all-reduce per iteration), the exact sharded
    right-hand side, arcs-over-ranks flow indicators and the merged top-k ranking."""
    res = run_workers("_dist_worker2.py", tmp_path / "res.json", 2)
    assert res["world"] == 2 and len(res["lp"]["blocks"]) == 2
    check_sharded_results(res)


def test_sharded_path_with_three_ranks(tmp_path):
    res = run_workers("_dist_worker2.py", tmp_path / "res.json", 3)
    assert res["world"] == 3
    check_sharded_results(res)


def test_column_sharded_simplex_makes_the_single_process_pivots(tmp_path):
    """ShardedLP.primal_simplex: one pricing all-gather (24 bytes per rank) and one broadcast of the entering column
    per pivot, everything else replicated.  Two and three gloo ranks make exactly the pivots of the single process
    (>= 100 of them) and reach HiGHS' optimum."""
    import importlib.util
    import numpy as np
    from scipy.optimize import linprog
    single = run_workers("_dist_worker3.py", tmp_path / "w1.json", 1)
    assert single["status"] == "OPTIMAL" and len(single["pivots"]) >= 100
    spec = importlib.util.spec_from_file_location("w3", os.path.join(HERE, "_dist_worker3.py"))
    w3 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w3)
    lp = w3.problem()
    ref = linprog(lp.c, A_ub=lp.A, b_ub=lp.b, bounds=np.c_[lp.l, lp.u], method="highs")
    assert ref.status == 0 and single["obj"] == pytest.approx(ref.fun, rel=1e-9)
    for world in (2, 3):
        res = run_workers("_dist_worker3.py", tmp_path / f"w{world}.json", world)
        assert res["world"] == world and res["status"] == "OPTIMAL"
        assert res["pivots"] == single["pivots"]
        assert res["obj"] == pytest.approx(single["obj"], rel=1e-12)


def test_column_sharded_re_solve_adds_the_single_process_columns(tmp_path):
    """ShardedLP.restricted_resolve: the restricted LP replicated, the pricing of the columns outside it rank-local, one
    all-gather of (|reduced cost|, column) records and one of the entering columns per round.  Two and three gloo ranks
    add exactly the columns of the single process in the same rounds (several rounds of at most 64 columns), end on
    its basis, and the optimum is HiGHS' optimum of the whole LP."""
    import importlib.util
    import numpy as np
    from scipy.optimize import linprog
    env = {"SX_TEST_BATCH": "64"}
    single = run_workers("_dist_worker4.py", tmp_path / "w1.json", 1, env)
    assert single["status"] == "OPTIMAL" and single["rounds"] >= 3 and all(0 < len(t) <= 64 for t in single["trace"])
    spec = importlib.util.spec_from_file_location("w4", os.path.join(HERE, "_dist_worker4.py"))
    w4 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w4)
    lp, start = w4.problem()
    lt = np.asarray(lp.sense) == "<"
    ref = linprog(lp.c, A_ub=lp.A[lt], b_ub=lp.b[lt], A_eq=lp.A[~lt], b_eq=lp.b[~lt], bounds=np.c_[lp.l, lp.u], method="highs")
    assert ref.status == 0 and single["obj"] == pytest.approx(ref.fun, rel=1e-8, abs=1e-8)
    for world in (2, 3):
        res = run_workers("_dist_worker4.py", tmp_path / f"w{world}.json", world, env)
        assert res["world"] == world and res["status"] == "OPTIMAL"
        assert res["trace"] == single["trace"] and res["R"] == single["R"]
        assert res["basic"] == single["basic"]
        assert res["obj"] == pytest.approx(single["obj"], rel=1e-12)


def test_a_failing_rank_takes_every_rank_out_of_the_sharded_resolve(tmp_path):
    """One rank's replicated solve raises in the second round: the failure travels with the record counts, the failing rank
    re-raises its own exception, the others a RuntimeError naming it -- and nobody waits in an all-gather (the bench's
    `sharded_resolve` leg hung exactly so in a rehearsal)."""
    out = tmp_path / "fail.json"
    run_workers("_dist_worker4.py", out, 3, {"SX_TEST_BATCH": "64", "SX_TEST_FAIL_RANK": "1"})
    msgs = [json.loads((tmp_path / f"fail.json.rank{r}").read_text()) for r in range(3)]
    assert msgs[1]["raised"].startswith("MemoryError: injected failure")
    for r in (0, 2):
        assert msgs[r]["raised"].startswith("RuntimeError: restricted_resolve: rank(s) [1] left round 2")
    assert all(m["rounds_added"] == 1 for m in msgs)
