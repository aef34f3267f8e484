"""GPU: entropic OT warm start (sx_sinkhorn_dev) against the restatement of POT's sinkhorn_knopp
(oracle/sinkhorn.py; POT itself is absent -> parity unpinned, see the oracle's header).  Floating point:
the two matrix-vector products sum in a different order than numpy's BLAS, so plans agree to a relative
1e-9, not bit for bit; the tolerance is stated where it is used."""
import numpy as np
import pytest

from oracle import sinkhorn as OS
import workloads

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def marginals(S, D, seed):
    rng = np.random.default_rng(seed)
    a = rng.random(S) + 0.05
    b = rng.random(D) + 0.05
    a /= a.sum()
    b /= b.sum()
    return a, b


@pytest.mark.parametrize("S,D,seed,reg,iters", [(12, 15, 0, 2.0, 1000), (1, 7, 1, 1.0, 50), (9, 1, 2, 1.0, 50),
                                                (100, 130, 3, 5.0, 1000), (257, 64, 4, 3.0, 37)])
def test_plan_matches_the_restated_algorithm(S, D, seed, reg, iters):
    from smart_crossover.sinkhorn import sinkhorn
    a, b = marginals(S, D, seed)
    M = np.random.default_rng(seed + 10).integers(0, 20, size=(S, D)).astype(float)
    want, log = OS.sinkhorn_knopp(a, b, M, reg, numItermax=iters)
    got, glog = sinkhorn(a, b, M, reg, numItermax=iters, log=True)
    assert got.shape == (S, D)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(glog["u"], log["u"], rtol=RTOL)
    np.testing.assert_allclose(glog["v"], log["v"], rtol=RTOL)
    # the stopping test fires at iterations 0, 10, 20, ...: the counts may differ by one test interval at most
    assert abs(glog["niter"] - log["iters"]) <= 10
    if glog["niter"] == log["iters"]:
        assert glog["err"] == pytest.approx(log["err"], rel=1e-6, abs=1e-15)


def test_config3_warm_start_feeds_the_crossover():
    """Grid cost of config 3 with the reference's parameters (reg = 10, 1000 iterations): the plan has
    the right marginals, matches the restatement, and TNET started from it reaches the optimal cost."""
    import io
    from contextlib import redirect_stdout
    from scipy.optimize import linprog
    from smart_crossover.formats import OptTransport
    from smart_crossover.network_methods.algorithms import network_crossover
    from smart_crossover.sinkhorn import sinkhorn
    from smart_crossover.solver_caller.caller import SolverSettings
    M = workloads.grid_cost(12).astype(float)               # 144 x 144 keeps the reference LP solve small
    S = D = M.shape[0]
    a, b = marginals(S, D, 9)
    want, log = OS.sinkhorn_knopp(a, b, M, 10.0, numItermax=1000)
    got, glog = sinkhorn(a, b, M, reg=10, numItermax=1000, log=True)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-300)
    assert np.abs(got.sum(axis=0) - b).max() < 1e-8 and glog["niter"] <= 1000
    ot = OptTransport(a.copy(), b.copy(), M.copy())
    mcf = ot.to_MCF()
    ref = linprog(mcf.c, A_eq=mcf.A, b_eq=mcf.b, bounds=(0, None), method="highs")
    with redirect_stdout(io.StringIO()):
        out = network_crossover(got.flatten(), ot=ot, method="tnet", solver="HIP", solver_settings=SolverSettings(log_console=0))
    X = out.x.reshape(S, D)
    assert float((X * M).sum()) == pytest.approx(ref.fun, rel=1e-8)
    assert np.allclose(X.sum(axis=1), a, atol=1e-8) and np.allclose(X.sum(axis=0), b, atol=1e-8)


def test_breakdown_keeps_the_previous_pair():
    """Costs so large that exp(-M/reg) underflows to 0 make K^T u vanish: POT restores the previous
    scaling pair and stops with a warning; so does the device."""
    from smart_crossover.sinkhorn import sinkhorn
    a = np.array([0.5, 0.2, 0.3])
    b = np.array([0.25, 0.75])
    M = np.full((3, 2), 1e6)
    want, log = OS.sinkhorn_knopp(a, b, M, 1.0, numItermax=100)
    assert log["breakdown"] and log["iters"] == 0
    with pytest.warns(UserWarning):
        got, glog = sinkhorn(a, b, M, 1.0, numItermax=100, log=True)
    assert glog["niter"] == 0
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)


def test_bad_shapes_raise():
    from smart_crossover.sinkhorn import sinkhorn
    with pytest.raises(ValueError):
        sinkhorn(np.ones(3) / 3, np.ones(4) / 4, np.zeros((4, 3)), 1.0)


def grid_instances(B, side, seed, sparsity=0.6):
    """B "image pairs" on a side x side grid: random positive masses with a share of empty pixels, as MNIST images
    have (scripts/mnist2ot.py keeps only the non-zero pixels of each image)."""
    rng = np.random.default_rng(seed)
    n = side * side
    a = rng.random((B, n)) * (rng.random((B, n)) > sparsity)
    b = rng.random((B, n)) * (rng.random((B, n)) > sparsity)
    a[:, 0] += 0.1
    b[:, -1] += 0.1
    a /= a.sum(axis=1, keepdims=True)
    b /= b.sum(axis=1, keepdims=True)
    return a, b, workloads.grid_cost(side)


@pytest.mark.parametrize("B,side,reg,iters", [(10, 28, 10.0, 1000), (3, 9, 2.0, 300), (16, 12, 4.0, 45), (1, 7, 1.5, 200)])
def test_batched_warm_starts_match_the_algorithm_on_each_instances_own_support(B, side, reg, iters):
    """Every instance of a batch must get what POT's algorithm gives on ITS support (the non-zero pixels of its
    two images, which is what the reference feeds it): plans equal to 1e-9 relative on the support, exactly zero
    off it, same iteration count up to one test interval."""
    from smart_crossover.sinkhorn import sinkhorn_batch
    a, b, M = grid_instances(B, side, seed=B + side)
    plans, logs = sinkhorn_batch(a, b, M, reg, numItermax=iters, log=True)
    assert plans.shape == (B, side * side, side * side)
    for k in range(B):
        sa, sb = np.flatnonzero(a[k]), np.flatnonzero(b[k])
        want, wlog = OS.sinkhorn_knopp(a[k][sa], b[k][sb], M[np.ix_(sa, sb)], reg, numItermax=iters)
        got = plans[k]
        np.testing.assert_allclose(got[np.ix_(sa, sb)], want, rtol=RTOL, atol=1e-300)
        off = got.copy()
        off[np.ix_(sa, sb)] = 0.0
        assert not off.any()                                   # no mass outside the instance's support
        assert abs(logs[k]["niter"] - wlog["iters"]) <= 10
        np.testing.assert_allclose(logs[k]["u"][sa], wlog["u"], rtol=RTOL)     # same start, same iterates
        np.testing.assert_allclose(logs[k]["v"][sb], wlog["v"], rtol=RTOL)
        assert not logs[k]["u"][np.setdiff1d(np.arange(a.shape[1]), sa)].any()      # unused pixels: scaling 0


def test_batch_equals_the_single_instance_entry_point():
    from smart_crossover.sinkhorn import sinkhorn, sinkhorn_batch
    a, b, M = grid_instances(4, 10, seed=77, sparsity=0.0)       # full support: both entry points apply
    plans = sinkhorn_batch(a, b, M, 3.0, numItermax=200)
    for k in range(4):
        np.testing.assert_allclose(plans[k], sinkhorn(a[k], b[k], M, 3.0, numItermax=200), rtol=RTOL, atol=1e-300)


def test_more_than_sixteen_instances_run_in_several_batches():
    from smart_crossover.sinkhorn import sinkhorn_batch
    a, b, M = grid_instances(19, 6, seed=5, sparsity=0.3)
    plans = sinkhorn_batch(a, b, M, 2.0, numItermax=100)
    for k in (0, 15, 16, 18):
        sa, sb = np.flatnonzero(a[k]), np.flatnonzero(b[k])
        want, _ = OS.sinkhorn_knopp(a[k][sa], b[k][sb], M[np.ix_(sa, sb)], 2.0, numItermax=100)
        np.testing.assert_allclose(plans[k][np.ix_(sa, sb)], want, rtol=RTOL, atol=1e-300)
