"""Which part of fit_model_scipy at a NaN level makes the NEXT capture crash?  usage: capture_bisect.py VARIANT"""
import faulthandler, os, sys
faulthandler.enable()
from copy import deepcopy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import MLLObjective
from gpplus_amd.errors import NanError, NotPSDError
from scipy.optimize import minimize
var = sys.argv[1]
rng = np.random.default_rng(21)
n = 96
X = rng.standard_normal((n, 3)); y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda"); m.train()
m.likelihood.raw_noise.requires_grad_(False)
torch.manual_seed(6)
def good(tag):
    obj = MLLObjective(m, True, [0, 0])
    print(tag, obj.fun(obj.pack_parameters())[0], "graphed", getattr(obj, "_graph", None) is not None, flush=True)
m.likelihood.initialize(noise=1.0); good("first")
with torch.no_grad():
    m.likelihood.raw_noise.fill_(float("nan"))
obj = MLLObjective(m, True, [0, 0])
x0 = obj.pack_parameters()
keep = None
try:
    if var == "fun":
        obj.fun(x0)
    elif var == "fun_keep_exc":
        try:
            obj.fun(x0)
        except NanError as e:
            keep = e
    elif var == "minimize":
        minimize(fun=obj.fun, x0=x0, args=(True), method="L-BFGS-B", jac=True)
    elif var == "minimize_keep":
        try:
            minimize(fun=obj.fun, x0=x0, args=(True), method="L-BFGS-B", jac=True)
        except NanError as e:
            keep = e
    elif var == "keep_no_tb":
        try:
            obj.fun(x0)
        except NanError as e:
            keep = e.with_traceback(None)
    elif var == "keep_tb_clear_frames":
        import traceback
        try:
            obj.fun(x0)
        except NanError as e:
            keep = e
            traceback.clear_frames(e.__traceback__)
    elif var == "extra_allocs":
        try:
            obj.fun(x0)
        except NanError:
            pass
        keep = [torch.empty(sz, dtype=torch.float64, device="cuda") for sz in (3, 1, 1, 96, 96 * 3, 96, 3, 16, 96 * 96, 7)]
    elif var == "reset_fun":
        st = deepcopy(m.state_dict()); m.reset_parameters(); m.load_state_dict(st); obj.fun(x0)
except (NanError, NotPSDError) as e:
    print("nan level raised", type(e).__name__, flush=True)
print("kept", type(keep).__name__, flush=True)
m.likelihood.initialize(noise=1e-3); good("after")
good("again")
