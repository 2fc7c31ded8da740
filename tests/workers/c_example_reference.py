"""The single-GPU path on the inputs of examples/shard_eval_c/main.cpp (the same LCG design), for tests/test_gpu_00_sharded_lists.py:
prints one JSON line with the quantities the C program prints."""
import json, os, sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.linalg import KernelSpec, exact_mll

N = int(sys.argv[1]); D, S = 6, 2
M64 = (1 << 64) - 1
state = 12345
def lcg():
    global state
    state = (state * 6364136223846793005 + 1442695040888963407) & M64
    return (state >> 11) / 9007199254740992.0
U = np.array([lcg() for _ in range(N * D)]).reshape(N, D)
y = np.empty(N)
for i in range(N):
    y[i] = np.sin(3.0 * U[i, 0]) + U[i, 1] * U[i, 1] + 0.05 * (lcg() - 0.5)
dev = torch.device("cuda", 0)
Ud = torch.tensor(U, device=dev)
w = torch.full((D,), 2.5, dtype=torch.float64, device=dev).requires_grad_(True)
sf2 = torch.tensor(0.8, dtype=torch.float64, device=dev).requires_grad_(True)
tau = torch.tensor([2e-3, 4e-3], dtype=torch.float64, device=dev).requires_grad_(True)
mean = torch.full((N,), 0.1, dtype=torch.float64, device=dev).requires_grad_(True)
grp = (torch.arange(N) % S).to(torch.int32).to(dev)
mll = exact_mll(Ud, KernelSpec(w=w, sf2=sf2, kind=0, d_split=0), tau, mean, torch.tensor(y, device=dev), grp=grp, n_grad_dims=0)
mll.backward()
print("REFERENCE " + json.dumps({"mll": float(mll), "alpha_norm": float(mean.grad.norm()), "g_w0": float(w.grad[0]), "g_sf2": float(sf2.grad),
                                 "g_tau0": float(tau.grad[0]), "g_tau1": float(tau.grad[1])}))
