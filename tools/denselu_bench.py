"""Timing of the dense LU (K16g, csrc/sx_denselu.hip) at the sizes of the sparse crossover's Schur complement.
usage: python tools/denselu_bench.py [n ...]"""
import sys, time, numpy as np
sys.path.insert(0, "smart-crossover_amd")
from smart_crossover.hip import default_context
from smart_crossover.hip.device import DenseLU
ctx = default_context()
for n in [int(a) for a in sys.argv[1:]] or [1000, 2000, 4000, 8000, 12000]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    A[np.arange(n), rng.permutation(n)] += 2.0
    dA = ctx.to_device(np.asfortranarray(A).ravel(order="F"))
    for rep in range(2):
        lu = DenseLU(ctx, n, dA); ctx.sync()
        t = time.perf_counter(); lu.factor(); ctx.sync(); dt = time.perf_counter() - t
        if rep == 0:
            lu.free()
    line = f"n {n}: factor {dt*1e3:.1f} ms = {2/3*n**3/dt/1e12:.2f} TFLOP/s;"
    for k in (1, 256):
        R = rng.standard_normal((n, k))
        for trans in (False, True):
            X = ctx.to_device(np.asfortranarray(R).ravel(order="F")); lu.solve(X, k, n, trans); ctx.sync()
            X = ctx.to_device(np.asfortranarray(R).ravel(order="F")); ctx.sync()
            t = time.perf_counter(); lu.solve(X, k, n, trans); ctx.sync(); ds = time.perf_counter() - t
            got = X.download().reshape((n, k), order="F")
            res = np.abs((A.T if trans else A) @ got - R).max()
            line += f" solve k={k}{'T' if trans else ''} {ds*1e3:.1f} ms (res {res:.0e});"
    print(line, flush=True)
    lu.free()
