"""A graph capture after an evaluation that failed with NanError: does it crash?  usage: python tools/attic/dev/capture_after_nan.py MODE
MODE: nan_graph (NaN point through the graphed objective's warm-up), nan_eager (NaN point with graphs off), none"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd import settings
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import MLLObjective
from gpplus_amd.errors import NanError, NotPSDError
mode = sys.argv[1]
rng = np.random.default_rng(21)
n = 96
X = rng.standard_normal((n, 3)); y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda"); m.train()
m.likelihood.raw_noise.requires_grad_(False)
def ev(tag):
    obj = MLLObjective(m, True, [0, 0])
    try:
        print(tag, obj.fun(obj.pack_parameters())[0], "graphed", getattr(obj, "_graph", None) is not None, flush=True)
    except (NanError, NotPSDError) as e:
        print(tag, "raised", type(e).__name__, flush=True)
m.likelihood.initialize(noise=1.0)
ev("first")
if mode != "none":
    with torch.no_grad():
        m.likelihood.raw_noise.fill_(float("nan"))
    if mode == "nan_eager":
        with settings.graphed_objective(False):
            ev("nan")
    else:
        ev("nan")
m.likelihood.initialize(noise=1e-3)
ev("after")
ev("again")
