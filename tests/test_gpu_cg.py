"""GPU parity: K4 projector norm (matrix-free CG on the device) against the reference goldens and
the CPU statement of the same algorithm.  Floating point: tolerance stated per case."""
import time

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_from
from oracle import lp_path as L
import workloads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from smart_crossover.hip import Context
    c = Context(0)
    yield c
    c.close()


def device_projector(ctx, A, b, c, sense, x_real, tol=1e-8, maxiter=1000):
    m, n = A.shape
    xx = L.standard_x(A, b, sense, x_real)
    xs = np.zeros(m)
    xs[L.slack_rows(sense)] = xx[n:]
    dA = ctx.matrix(A)
    res = ctx.projector_norm(dA, ctx.to_device(xx[:n]), ctx.to_device(xs), ctx.to_device(c), tol, maxiter)
    dA.free()
    return res


@pytest.mark.parametrize("gname,rel", [("g1", 1e-9), ("g2", 1e-5)])
def test_projector_matches_reference_golden(ctx, gname, rel, request):
    g = request.getfixturevalue(gname)
    A = csr_from(g, "A")
    res = device_projector(ctx, A, g["b"], g["c"], g["sense"], g["x_real"])
    n_std = A.shape[1] + int(np.count_nonzero(g["sense"] == "<"))
    assert res.converged == 1
    assert res.proj_norm == pytest.approx(float(g["proj_norm"]), rel=rel)
    assert res.proj_norm / n_std == pytest.approx(float(g["sf"]), rel=rel)
    # iteration count is rounding sensitive; it must be in the same regime as the reference's
    assert abs(int(res.iters) - int(g["cg_iters"])) <= max(3, int(0.1 * int(g["cg_iters"])))


@pytest.mark.parametrize("m,n,k,seed,floor", [(200, 900, 5, 3, 1e-3), (1000, 5000, 8, 4, 0.1)])
def test_projector_matches_cpu_matrix_free(ctx, m, n, k, seed, floor):
    inst = workloads.sparse_lp(m, n, k, seed=seed, stratified=False)
    xr = L.x_perturb_val(inst.x, inst.l, inst.u)
    xr = np.maximum(xr, floor)           # keep CG in the converged regime (850 / 630 iterations on the CPU)
    proj, iters = L.projector_Xc(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, xr, explicit=False)
    res = device_projector(ctx, inst.A, inst.b, inst.c, inst.sense, xr)
    assert res.converged == 1 and iters < 1000
    assert res.proj_norm == pytest.approx(float(np.linalg.norm(proj)), rel=1e-7)
    assert abs(int(res.iters) - iters) <= max(3, int(0.1 * iters))
    # and the reference's own (explicit Y Y^T) arithmetic
    proj_ref, _ = L.projector_Xc(inst.A, inst.b, inst.c, inst.l, inst.u, inst.sense, xr, explicit=True)
    assert res.proj_norm == pytest.approx(float(np.linalg.norm(proj_ref)), rel=1e-6)


def test_projector_trivial_and_maxiter(ctx):
    inst = workloads.sparse_lp(100, 400, 4, seed=5, stratified=False)
    xr = np.full(400, 0.5)
    # c = 0 -> b = 0 -> immediate exit, projection of the zero vector
    res = device_projector(ctx, inst.A, inst.b, np.zeros(400), inst.sense, xr)
    assert res.iters == 0 and res.converged == 1 and res.proj_norm == 0.0
    # maxiter cut: not converged, finite result, iteration count honoured
    res = device_projector(ctx, inst.A, inst.b, inst.c, inst.sense, xr, tol=1e-14, maxiter=7)
    assert res.iters == 7 and res.converged == 0 and np.isfinite(res.proj_norm)
    # same cut on the CPU statement
    xx = L.standard_x(inst.A, inst.b, inst.sense, xr)
    proj, it = L.projector_matrix_free(inst.A, inst.sense, xx, L.standard_c(inst.c, inst.sense), tol=1e-14, maxiter=7)
    assert it == 7
    assert res.proj_norm == pytest.approx(float(np.linalg.norm(proj)), rel=1e-9)


def test_projector_graph_replay_equals_direct_launches(ctx):
    """The hipGraph replay of the iteration batch is an execution detail: same kernels, same order,
    hence bit-identical results and iteration counts."""
    inst = workloads.sparse_lp(300, 1500, 6, seed=12, stratified=False)
    xr = np.maximum(L.x_perturb_val(inst.x, inst.l, inst.u), 1e-2)
    out = {}
    for flag in (0, 1):
        ctx.set_option("graph", flag)
        for maxiter in (1000, 50, 7):          # 50: two graph batches + 2 direct; 7: direct only
            r = device_projector(ctx, inst.A, inst.b, inst.c, inst.sense, xr, maxiter=maxiter)
            out[(flag, maxiter)] = (r.proj_norm, int(r.iters), int(r.converged), r.rel_residual)
    ctx.set_option("graph", 1)
    for maxiter in (1000, 50, 7):
        assert out[(0, maxiter)] == out[(1, maxiter)], maxiter
    assert out[(1, 50)][1] == 50 and out[(1, 7)][1] == 7 and out[(1, 1000)][2] == 1


def test_projector_config2_size_hits_cap(ctx, capsys):
    """Config 2 (2e4 x 1e5): the reference's CG runs into maxiter=1000 here and its
    result then depends on rounding order; the device must finish, report 1000 iterations and land
    within a factor 2 of the CPU matrix-free statement.  Prints the wall time for DESIGN.md."""
    inst = workloads.config2()
    xr = L.x_perturb_val(inst.x, inst.l, inst.u)
    t0 = time.perf_counter()
    res = device_projector(ctx, inst.A, inst.b, inst.c, inst.sense, xr)
    gpu_s = time.perf_counter() - t0
    assert res.iters == 1000 and res.converged == 0 and np.isfinite(res.proj_norm)
    xx = L.standard_x(inst.A, inst.b, inst.sense, xr)
    t0 = time.perf_counter()
    proj, it = L.projector_matrix_free(inst.A, inst.sense, xx, L.standard_c(inst.c, inst.sense))
    cpu_s = time.perf_counter() - t0
    ratio = res.proj_norm / float(np.linalg.norm(proj))
    with capsys.disabled():
        print(f"\n[K4 @c2] device {gpu_s*1e3:.0f} ms (incl. upload) vs CPU matrix-free {cpu_s:.1f} s; "
              f"proj_norm {res.proj_norm:.6e} vs {np.linalg.norm(proj):.6e} (ratio {ratio:.4f}), rel_resid {res.rel_residual:.2e}")
    assert 0.5 < ratio < 2.0


def test_projector_of_a_row_less_lp_is_the_identity():
    """m = 0: Y is empty, the projector of the reference returns v unchanged (algorithms.py:183-187) -- norm ||xa .* c||,
    not 0 -- and a row-less LP is not mistaken for a feasibility problem."""
    import scipy.sparse as sp
    from smart_crossover.hip import Context
    ctx = Context(0)
    n = 37
    rng = np.random.default_rng(2)
    xa, c = rng.uniform(0.1, 1, n), rng.standard_normal(n)
    dA = ctx.matrix(sp.csr_matrix((0, n)))
    pc = ctx.empty(n, np.float64)
    res = ctx.projector_norm(dA, ctx.to_device(xa), ctx.to_device(np.zeros(0)), ctx.to_device(c), 1e-8, 100, pc, None)
    assert res.proj_norm == pytest.approx(float(np.linalg.norm(xa * c)), rel=1e-14)
    np.testing.assert_array_equal(pc.download(), xa * c)
    ctx.close()
