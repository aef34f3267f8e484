"""GPU: the library's device memory pool (csrc/sx_pool.hip, `sx_pool_trim` / `sx_pool_stats`): a freed block serves the
next request of its size class, blocks come back to the driver on `pool_trim`, and a crossover called twice allocates
nothing from the driver the second time.  The reference has no counterpart (its solver owns its memory,
solver_caller/gurobi.py:111-115); what is checked is the bookkeeping and that results do not depend on recycled memory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def test_freed_block_serves_the_next_request(ctx):
    from smart_crossover.hip.device import DeviceArray, pool_stats, pool_trim
    pool_trim()
    n = 3_000_001
    a = DeviceArray(ctx, n, np.dtype(np.float64))
    a.upload(np.full(n, 7.0))
    p0 = a.ptr
    s0 = pool_stats()
    a.free()
    s1 = pool_stats()
    assert s1["cached_bytes"] >= s0["cached_bytes"] + 8 * n
    assert s1["live_bytes"] <= s0["live_bytes"] - 8 * n
    b = DeviceArray(ctx, n - 1000, np.dtype(np.float64))      # (same size class)
    s2 = pool_stats()
    assert b.ptr == p0 and s2["hits"] == s1["hits"] + 1 and s2["misses"] == s1["misses"]
    host = np.arange(n - 1000, dtype=np.float64)
    assert np.array_equal(b.upload(host).download(), host)
    b.free()
    pool_trim()
    assert pool_stats()["cached_bytes"] == 0
    c = DeviceArray(ctx, n, np.dtype(np.float64))
    assert pool_stats()["misses"] == s2["misses"] + 1
    c.free()


def test_second_crossover_allocates_nothing_from_the_driver(ctx, monkeypatch):
    """Two calls of the sparse crossover on the same LP: same vertex bit for bit (recycled memory does not leak into the
    results), and the second call is served by the pool alone."""
    import io
    from contextlib import redirect_stdout

    import workloads
    from smart_crossover.formats import GeneralLP
    from smart_crossover.hip.device import pool_stats
    from smart_crossover.lp_methods import algorithms as alg
    from smart_crossover.solver_caller import solving
    from smart_crossover.solver_caller.caller import SolverSettings
    inst = workloads.netlib_lp(4000, 40000, seed=5)
    lp = GeneralLP(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense)
    monkeypatch.setenv("SX_LP_CROSSOVER", "band")

    def run():
        with redirect_stdout(io.StringIO()):
            mgr = alg.get_perturb_problem(lp, inst.x, inst.y, 1e-3, 1e-3, False)
            caller = solving.generate_solver_caller("HIP", SolverSettings(presolve="on", log_console=0))
            caller.read_genlp(mgr.lp_sub)
            caller.add_warm_start_solution((mgr.get_subx(inst.x), inst.y))
            caller.run_barrier()
            out = caller.return_output()
        assert caller.solved_by == "crossover_band" and out.status == "OPTIMAL"
        return out

    o1 = run()
    s1 = pool_stats()
    o2 = run()
    s2 = pool_stats()
    assert np.array_equal(o1.x, o2.x) and np.array_equal(o1.y, o2.y)
    assert np.array_equal(o1.basis.vbasis, o2.basis.vbasis) and np.array_equal(o1.basis.cbasis, o2.basis.cbasis)
    assert s2["hits"] > s1["hits"]
    assert s2["misses"] == s1["misses"], (s1, s2)
