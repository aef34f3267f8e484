import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, "tests"); sys.path.insert(0, "smart-crossover_amd")
import importlib.util
spec = importlib.util.spec_from_file_location("tb", "tests/test_gpu_bandlu.py"); tb = importlib.util.module_from_spec(spec); spec.loader.exec_module(tb)
from smart_crossover.hip import default_context
ctx = default_context()
n, kl, ku = 100000, 117, 113
A = tb.dominant_band(n, kl, ku, 5, False)
t = time.perf_counter(); lu, rep, piv = tb.factor(ctx, A, kl, ku); ctx.sync(); print("factor", time.perf_counter() - t)
rng = np.random.default_rng(1)
for nrhs in (1, 8, 512):
    B = rng.standard_normal((n, nrhs))
    for trans in (False, True):
        X = ctx.to_device(np.asfortranarray(B).ravel(order="F")); ctx.sync()
        t = time.perf_counter(); lu.solve(X, nrhs, n, trans); ctx.sync(); dt = time.perf_counter() - t
        got = X.download().reshape((n, nrhs), order="F")
        M = sp.csc_matrix(A)
        r = (M.T if trans else M) @ got - B
        print(f"nrhs {nrhs} trans {trans}: {dt*1e3:.1f} ms, resid {np.abs(r).max():.2e}")
