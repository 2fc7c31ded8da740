"""world_size-2 gloo test of the restart-parallel driver (the N>1 path of bench.py / GP_Plus fits).  The evaluation
engine is injected: here the CPU ORACLE plays the part of the HIP back end (tests may use the oracle as a stand-in
checker; the product's engine is fit_model_torch on the GPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleBackedModel(torch.nn.Module):
    """Quacks like a GPR for the driver: parameters, state_dict, reset_parameters."""

    def __init__(self, X, y):
        super().__init__()
        from oracle.gp_oracle import OracleGP

        self.o = OracleGP(X, y)
        self.theta = torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(v.clone()) for k, v in self.o.params.items()})

    def loss(self):
        p = {k.replace("__", "."): v for k, v in self.theta.items()}
        return self.o.loss(p)

    def reset_parameters(self):
        with torch.no_grad():
            for v in self.theta.values():
                v.copy_(torch.randn_like(v) * 0.5 - 1.0)


def _adam_fit(model, num_restarts=0, num_iter=15, lr=0.05):
    from copy import deepcopy

    best, best_state, hists = float("inf"), deepcopy(model.state_dict()), []
    for i in range(num_restarts + 1):
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        h = []
        for _ in range(num_iter):
            opt.zero_grad()
            loss = model.loss()
            loss.backward()
            opt.step()
            h.append(loss.item())
        hists.append(h)
        if h[-1] < best:
            best, best_state = h[-1], deepcopy(model.state_dict())
        if i < num_restarts:
            model.reset_parameters()
    model.load_state_dict(best_state)
    return best, hists


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpplus_amd.optim.mll_parallel import fit_restarts_parallel

    rng = np.random.default_rng(0)
    X = rng.standard_normal((60, 3))
    y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
    torch.set_num_threads(1)
    m = _OracleBackedModel(X, y)
    f, hist = fit_restarts_parallel(m, num_restarts=2, fit_fn=_adam_fit, seed=11)
    final = m.loss().item()
    flat = torch.cat([v.detach().reshape(-1) for v in m.state_dict().values()])
    torch.save({"f": f, "final": final, "flat": flat, "nhist": len(hist)}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_split_restarts():
    from gpplus_amd.optim.mll_parallel import split_restarts

    assert split_restarts(5, 2) == [3, 2]
    assert split_restarts(1, 4) == [1, 0, 0, 0]
    assert sum(split_restarts(65, 8)) == 65 and max(split_restarts(65, 8)) - min(split_restarts(65, 8)) <= 1


def test_restart_parallel_two_ranks(tmp_path):
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["nhist"] == 2 and r1["nhist"] == 1          # 3 starts split 2 + 1
    assert r0["f"] == r1["f"]                                # both agree on the incumbent
    assert torch.equal(r0["flat"], r1["flat"])              # ... and hold the same (best) state
    assert abs(r0["final"] - r0["f"]) < 5e-2                # the state reproduces (about) the reported loss


def _scipy_fit(model, theta0_list=None, num_restarts=0, maxiter=25):
    """Stand-in for fit_model_scipy with the same contract: L-BFGS from every start, the best optimum loaded into the model."""
    from scipy.optimize import minimize

    names = list(model.theta.keys())
    shapes = [model.theta[k].shape for k in names]

    def load(x):
        i = 0
        with torch.no_grad():
            for k, sh in zip(names, shapes):
                n = int(np.prod(sh)) if len(sh) else 1
                model.theta[k].copy_(torch.as_tensor(x[i:i + n]).reshape(sh))
                i += n

    def fun(x):
        load(x)
        for p in model.parameters():
            p.grad = None
        loss = model.loss()
        loss.backward()
        return loss.item(), np.concatenate([model.theta[k].grad.reshape(-1).numpy() for k in names])

    out = [minimize(fun, x0, jac=True, method="L-BFGS-B", options={"maxiter": maxiter}) for x0 in theta0_list]
    best = int(np.argmin([r.fun for r in out]))
    load(out[best].x)
    return out, float(out[best].fun)


def _pack(model):
    return np.concatenate([v.detach().reshape(-1).numpy() for v in model.theta.values()])


def _worker_scipy(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpplus_amd.optim.mll_parallel import fit_scipy_parallel

    rng = np.random.default_rng(0)
    X = rng.standard_normal((60, 3))
    y = np.sin(X[:, 0]) + 0.1 * X[:, 1]
    torch.set_num_threads(1)
    m = _OracleBackedModel(X, y)
    res, f = fit_scipy_parallel(m, num_restarts=4, seed=5, fit_fn=_scipy_fit, pack_fn=_pack)
    flat = torch.cat([v.detach().reshape(-1) for v in m.state_dict().values()])
    torch.save({"f": f, "final": m.loss().item(), "flat": flat, "nres": len(res), "funs": [float(r.fun) for r in res],
                "x0": [r.x.copy() for r in res]}, os.path.join(out, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_scipy_multistart_over_two_ranks(tmp_path):
    """fit_model_scipy's starts mapped over ranks (optim/mll_scipy.py:287-293 maps them over joblib workers): 5 starts split 3 + 2,
    both ranks end with the same best objective and the same state, which is the best of ALL starts."""
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_worker_scipy, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt", weights_only=False), torch.load(tmp_path / "s1.pt", weights_only=False)
    assert r0["nres"] == 3 and r1["nres"] == 2
    assert r0["f"] == r1["f"] == min(r0["funs"] + r1["funs"])
    assert torch.equal(r0["flat"], r1["flat"])
    assert abs(r0["final"] - r0["f"]) < 1e-9
