"""usage: python tools/attic/dev/continuation_probe.py N num_restarts [maxiter]"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import fit_model_continuation, fit_model_scipy
n, nr = int(sys.argv[1]), int(sys.argv[2])
opts = {"maxiter": int(sys.argv[3])} if len(sys.argv) > 3 else {}
rng = np.random.default_rng(21)
X = rng.standard_normal((n, 3)); y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda"); m.train()
torch.manual_seed(6)
import gpplus_amd.optim.mll_noise_continuation as C
_orig = C.fit_model_scipy
def _wrapped(model, *a, **k):
    print("level: noise", float(model.likelihood.noise.reshape(-1)[0]), "raw", float(model.likelihood.raw_noise.reshape(-1)[0]), flush=True)
    out = _orig(model, *a, **k)
    print("   -> nll", out[1], flush=True)
    return out
C.fit_model_scipy = _wrapped
if os.environ.get("EAGER"):
    from gpplus_amd import settings
    settings.graphed_objective._v = False
if os.environ.get("ONLY_SCIPY"):
    m.likelihood.raw_noise.requires_grad_(False)
    m.likelihood.initialize(noise=1.0)
    print(fit_model_scipy(m, True, num_restarts=nr, options=opts)[1], flush=True)
else:
    nll, hist = fit_model_continuation(m, num_restarts=nr, initial_noise_var=1.0, verbose=False, options=opts)
    print("nll", nll, [float(v.reshape(-1)[0]) for v in hist["noise_history"]], hist["nll_history"], flush=True)
