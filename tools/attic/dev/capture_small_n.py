"""Does the graphed scipy objective capture at small N?  usage: python tools/attic/dev/capture_small_n.py N [d]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import MLLObjective
n, d = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(21)
X = rng.standard_normal((n, d)); y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda"); m.train()
if len(sys.argv) > 3:
    m.likelihood.raw_noise.requires_grad_(False)
obj = MLLObjective(m, True, [0, 0])
x = obj.pack_parameters()
print("N", n, "d", d, "f", obj.fun(x)[0], "graphed", obj._graphed() is not None, flush=True)
