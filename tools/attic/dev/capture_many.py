"""Many graph captures in a row (one per noise level), optionally with a NaN level in between.  usage: capture_many.py [nan_at]"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import MLLObjective, fit_model_scipy
from gpplus_amd.errors import NanError, NotPSDError
nan_at = int(sys.argv[1]) if len(sys.argv) > 1 else -1
use_fit = len(sys.argv) > 2
rng = np.random.default_rng(21)
n = 96
X = rng.standard_normal((n, 3)); y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda"); m.train()
m.likelihood.raw_noise.requires_grad_(False)
torch.manual_seed(6)
for i in range(14):
    if i == nan_at:
        with torch.no_grad():
            m.likelihood.raw_noise.fill_(float("nan"))
    else:
        m.likelihood.initialize(noise=10.0 ** (-(i % 7)))
    if use_fit:
        res, nll = fit_model_scipy(m, True, num_restarts=0)
        print(i, "fit nll", nll, flush=True)
        continue
    obj = MLLObjective(m, True, [0, 0])
    try:
        print(i, obj.fun(obj.pack_parameters())[0], "graphed", getattr(obj, "_graph", None) is not None, flush=True)
    except (NanError, NotPSDError) as e:
        print(i, "raised", type(e).__name__, flush=True)
