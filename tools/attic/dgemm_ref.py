"""Known-good fp64 GEMM reference on this box: rocBLAS through torch.mm (measurement only, never used by the product)."""
import torch, time
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, dtype=torch.float64, device="cuda"); b = torch.randn(n, n, dtype=torch.float64, device="cuda")
    c = a @ b; torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"rocBLAS dgemm n={n}: {ms:.2f} ms  {2*n**3/ms/1e9:.2f} TFLOP/s")
n = 16384
a = torch.randn(n, n, dtype=torch.float64, device="cuda"); a = a @ a.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t = time.time(); L = torch.linalg.cholesky(a); torch.cuda.synchronize(); dt = time.time() - t
print(f"torch.linalg.cholesky n={n}: {dt*1e3:.1f} ms  {n**3/3/dt/1e12:.2f} TFLOP/s")
