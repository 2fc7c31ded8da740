"""Multistart Adam fit at the reference's example sizes: the sequential driver (optim/mll_torch.py semantics) against the
batched one (all restarts in every launch).  usage: python tools/bench_restarts.py [N] [restarts] [iters]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import BatchedObjective, fit_model_torch, fit_model_torch_batched
from gpplus_amd.preprocessing import standard
from gpplus_amd.test_functions.analytical import borehole
from gpplus_amd.utils import set_seed

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 100
X, y = borehole(n=10000, random_state=12345)
X = torch.tensor(X[:N]); y = torch.tensor(y[:N]); X, _, _ = standard(X, {})

def model():
    set_seed(1)
    return GP_Plus(X, y, dtype=torch.float64, device="cuda")

m = model(); fit_model_torch_batched(m, num_restarts=1, num_iter=3)   # one-time costs: workspace, code objects
m = model(); fit_model_torch(m, num_restarts=0, num_iter=3, verbose=False)
torch.cuda.synchronize()
m = model(); set_seed(2); t0 = time.perf_counter(); fb, hb = fit_model_torch_batched(m, num_restarts=R, num_iter=IT); torch.cuda.synchronize(); tb = time.perf_counter() - t0
m = model(); set_seed(2); t0 = time.perf_counter(); fs, hs = fit_model_torch(m, num_restarts=R, num_iter=IT, verbose=False); torch.cuda.synchronize(); ts = time.perf_counter() - t0
ev = sum(len(h) for h in hs)
print("N=%d, %d restarts x %d Adam steps (%d evaluations): sequential %.2f s (%.2f ms/eval), batched %.2f s (%.3f ms/eval) -> %.1fx;  best loss %.5f vs %.5f"
      % (N, R + 1, IT, ev, ts, ts / ev * 1e3, tb, tb / ev * 1e3, ts / tb, fs, fb))
for B in (1, 8, 64, 256):
    obj = BatchedObjective(model(), B); obj.sample_restarts()
    def step():
        for p in obj.theta.values(): p.grad = None
        torch.nansum(obj.loss()).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("  batched evaluation, B=%3d: %.2f ms per launch set = %.3f ms per run = %.0f evals/s" % (B, dt * 1e3, dt * 1e3 / B, B / dt))
