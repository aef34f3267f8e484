"""GPU: the first-order stage of the LP re-solve (K16p, csrc/sx_pdlp.hip) against its CPU statement
oracle/pdlp.py -- the same iteration, compared after a fixed number of steps (1e-9 relative: the two sides sum
in different orders), its decisions (restarts, primal weight), and the optimum against HiGHS.  The reference
runs Gurobi's barrier here (lp_methods/algorithms.py:50-54): parity unpinned."""
import numpy as np
import pytest
from scipy.optimize import linprog

from oracle import pdlp as P
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import default_context
    return default_context()


def device(ctx, inst, x0, y0, max_iter, tol):
    m, n = inst.A.shape
    dA = ctx.matrix(inst.A)
    put = lambda v, t=np.float64: ctx.to_device(np.ascontiguousarray(v, dtype=t))   # noqa: E731
    d_x, d_y = ctx.empty(n, np.float64), ctx.empty(m, np.float64)
    res = ctx.pdlp(dA, put(inst.b), put(inst.c), put(inst.l), put(inst.u), put(inst.sense == "<", np.uint8),
                   None if x0 is None else put(x0), None if y0 is None else put(y0), max_iter, tol, d_x, d_y)
    out = res, d_x.download(), d_y.download()
    dA.free()
    return out


@pytest.mark.parametrize("m,n,k,seed,warm", [(27, 51, 2, 2024, True), (150, 400, 5, 2, True), (150, 400, 5, 2, False),
                                             (2000, 6000, 6, 3, True)])
def test_same_iteration_as_the_oracle(ctx, m, n, k, seed, warm):
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=False)
    inst.c = inst.c + 1e-2 * np.random.default_rng(seed).uniform(0.9, 1.0, n) / np.maximum(inst.x, 1e-2)   # perturbed cost
    x0, y0 = (inst.x, inst.y) if warm else (None, None)
    for iters in (64, 256):
        res, x, y = device(ctx, inst, x0, y0, iters, 1e-14)
        want = P.pdlp(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense == "<", x0, y0, max_iter=iters, tol=1e-14)
        assert int(res.status) == want["status"] == 3 and int(res.iters) == want["iters"] == iters
        assert int(res.restarts) == want["restarts"]
        assert float(res.step) == pytest.approx(want["step"], rel=1e-10)
        assert float(res.primal_weight) == pytest.approx(want["primal_weight"], rel=1e-8)
        np.testing.assert_allclose(x, want["x"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(y, want["y"], rtol=1e-9, atol=1e-11)
        assert float(res.primal_obj) == pytest.approx(want["primal_obj"], rel=1e-9, abs=1e-11)
        assert float(res.primal_residual) == pytest.approx(want["primal_residual"], rel=1e-6, abs=1e-12)


def test_converges_to_the_optimum_and_is_reproducible(ctx):
    inst = workloads.sparse_lp(300, 900, 5, seed=7, stratified=False)
    inst.c = inst.c + 1e-2 * np.random.default_rng(7).uniform(0.9, 1.0, 900) / np.maximum(inst.x, 1e-2)
    lt = inst.sense == "<"
    ref = linprog(inst.c, A_ub=inst.A[lt], b_ub=inst.b[lt], A_eq=inst.A[~lt], b_eq=inst.b[~lt],
                  bounds=list(zip(inst.l, [None if np.isinf(v) else v for v in inst.u])), method="highs")
    assert ref.status == 0
    res, x, y = device(ctx, inst, inst.x, inst.y, 400000, 1e-9)
    assert int(res.status) == 0
    assert float(res.primal_obj) == pytest.approx(ref.fun, rel=1e-6, abs=1e-6)
    r = inst.b - inst.A @ x
    assert np.abs(r[~lt]).max() < 1e-6 and r[lt].min() > -1e-6
    assert np.all(x >= inst.l) and np.all(x <= inst.u) and np.all(y[lt] <= 0)
    # no atomics, fixed reduction orders: the same run twice is the same bits
    res2, x2, y2 = device(ctx, inst, inst.x, inst.y, 400000, 1e-9)
    assert int(res2.iters) == int(res.iters) and x2.tobytes() == x.tobytes() and y2.tobytes() == y.tobytes()
    # graph replay is an execution detail
    ctx.set_option("graph", 0)
    try:
        res3, x3, y3 = device(ctx, inst, inst.x, inst.y, 400000, 1e-9)
    finally:
        ctx.set_option("graph", 1)
    assert int(res3.iters) == int(res.iters) and x3.tobytes() == x.tobytes()
