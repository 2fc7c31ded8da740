"""Which kernel does torch.mm (fp64) run on this box, and how fast?  (dev tool; run under rocprofv3 --kernel-trace --stats)"""
import time, torch
n = 16384
a = torch.randn(n, n, dtype=torch.float64, device="cuda"); b = torch.randn(n, n, dtype=torch.float64, device="cuda")
for _ in range(2): c = a @ b
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): c = a.t() @ b
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"torch.mm TN fp64 n={n}: {dt*1e3:.2f} ms = {2*n**3/dt/1e12:.1f} TFLOP/s")
t0 = time.perf_counter()
for _ in range(5): c = a @ b
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"torch.mm NN fp64 n={n}: {dt*1e3:.2f} ms = {2*n**3/dt/1e12:.1f} TFLOP/s")
