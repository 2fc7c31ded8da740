"""Prediction against a cached factorisation (models/gpregression.py:122-149 on the HIP back end: gpp_cross_kernel + V = K_*N L^-T on the
MFMA GEMM + gpp_predict): the first call of an eval() phase factors the training covariance, later calls reuse it.
usage: python tools/bench_predict.py [N] [M ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Ms = [int(a) for a in sys.argv[2:]] or [1000, 10000]
X, y, kw, theta = make_config("C2", N)
m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw)
apply_theta(m, theta)
g = torch.Generator().manual_seed(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.predict(X[:16].cuda(), return_std=True); torch.cuda.synchronize()
print(f"N={N}: first predict of the eval() phase (factorisation + inverse factor): {1e3 * (time.perf_counter() - t0):.1f} ms")
for M in Ms:
    Xt = (X[torch.randint(0, N, (M,), generator=g)] + 0.01 * torch.randn(M, X.shape[1], generator=g, dtype=X.dtype)).cuda()
    m.predict(Xt, return_std=True); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); mean, std = m.predict(Xt, return_std=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    m.predict(Xt, return_std=False); torch.cuda.synchronize(); t0 = time.perf_counter(); m.predict(Xt, return_std=False); torch.cuda.synchronize(); tm = time.perf_counter() - t0
    print(f"  M={M:6d}: mean + std {1e3 * t:8.2f} ms = {M / t:10.0f} points/s ({M * N * N / t / 1e12:.1f} TFLOP/s on M N^2); mean only {1e3 * tm:8.2f} ms")
