"""Bisect the crash of the tail segment's capture with the real model.  usage: segment_model_probe.py VARIANT"""
import faulthandler, os, sys
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.graphed import GraphedSegment
from gpplus_amd.gpcore import ExactMarginalLogLikelihood
from gpplus_amd.gpcore.module import Module
from gpplus_amd.models import GP_Plus
from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config
var = sys.argv[1]
X, y, kw, theta = make_config("C2", 4096)
m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw); apply_theta(m, theta); m.train()
mll = ExactMarginalLogLikelihood(m.likelihood, m)
params = list(m.parameters())
x = m.train_inputs[0]
dev = x.device
def fwd_fn():
    out = Module.__call__(m, x)
    cov = out.lazy_covariance_matrix
    return (out.mean, cov.spec.w, cov.spec.sf2.reshape(1))
def tail_noise():
    return (m.likelihood.noise_covar.noise.reshape(-1),)
def tail_priors():
    return (mll._prior_sum(torch.float64).reshape(1),)
def tail_both():
    return (m.likelihood.noise_covar.noise.reshape(-1), mll._prior_sum(torch.float64).reshape(1))
tails = {"noise": tail_noise, "priors": tail_priors, "both": tail_both}
if os.environ.get("FWD") == "os":
    fwd_fn = lambda: (m.covar_module.outputscale.reshape(1),)
elif os.environ.get("FWD") == "mean":
    fwd_fn = lambda: (m.mean_module(x),)
elif os.environ.get("FWD") == "w":
    fwd_fn = lambda: (m.covar_module(x).spec.w,)
elif os.environ.get("FWD") == "rawexp":
    fwd_fn = lambda: (m.covar_module.raw_outputscale.exp().reshape(1),)
if os.environ.get("TAIL") == "rawexp":
    tails["noise"] = lambda: (m.likelihood.noise_covar.raw_noise.exp().reshape(-1),)
if os.environ.get("PARAMS") == "two":
    params = [m.covar_module.raw_outputscale, m.likelihood.noise_covar.raw_noise]
if var.startswith("tail_first_"):
    t = GraphedSegment(tails[var[len("tail_first_"):]], params, dev); print("tail built", flush=True)
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
elif var.startswith("fwd_then_"):
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
    t = GraphedSegment(tails[var[len("fwd_then_"):]], params, dev); print("tail built", flush=True)
elif var.startswith("fwdcall_then_"):
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
    o = f(); print("fwd replayed", flush=True)
    t = GraphedSegment(tails[var[len("fwdcall_then_"):]], params, dev); print("tail built", flush=True)
elif var.startswith("fwdcalldel_then_"):
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
    o = f(); print("fwd replayed", flush=True)
    del o
    t = GraphedSegment(tails[var[len("fwdcalldel_then_"):]], params, dev); print("tail built", flush=True)
elif var.startswith("fwdcallbwd_then_"):
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
    o = f(); sum(v.sum() for v in o).backward(); print("fwd replayed + backward", flush=True)
    t = GraphedSegment(tails[var[len("fwdcallbwd_then_"):]], params, dev); print("tail built", flush=True)
elif var.startswith("fwdcallnograd_then_"):
    f = GraphedSegment(fwd_fn, params, dev); print("fwd built", flush=True)
    with torch.no_grad():
        o = f()
    print("fwd replayed (no grad)", flush=True)
    t = GraphedSegment(tails[var[len("fwdcallnograd_then_"):]], params, dev); print("tail built", flush=True)
elif var == "two_tails":
    t1 = GraphedSegment(tail_noise, params, dev); print("t1 built", flush=True)
    t = GraphedSegment(tail_priors, params, dev); print("t2 built", flush=True)
    f = None
o2 = t()
print("ok", [float(v.sum()) for v in o2], flush=True)
