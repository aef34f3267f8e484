import torch, time
m = 20000
C = torch.zeros(m, m, dtype=torch.float64, device="cuda")
for K in (32, 64, 128, 256):
    E = torch.randn(m, K, dtype=torch.float64, device="cuda"); R = torch.randn(m, K, dtype=torch.float64, device="cuda")
    for _ in range(2): C.addmm_(E, R.t(), alpha=-1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): C.addmm_(E, R.t(), alpha=-1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"K={K}: {dt*1e3:.3f} ms per rank-{K} update of a {m}^2 fp64 matrix = {dt/K*1e6:.1f} us per pivot; {2*m*m*K/dt/1e12:.1f} TFLOP/s, {16*m*m/dt/1e12:.2f} TB/s of C traffic")
