"""Soak of the replayed objectives (graphed.py, optim/mll_batched.py): many thousands of replays of the L-BFGS objective at several
sizes with eager evaluations, device synchronisations and allocations in between, each compared with the value the eager path gave
for the same point; then repeated batched fits that must reproduce the first one bit for bit.  Dev tool: python tools/attic/soak_replay.py [s]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd import settings
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import fit_model_torch_batched
from gpplus_amd.optim.mll_scipy import MLLObjective
from gpplus_amd.utils import set_seed

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(0)
objs = []
for n in (300, 500, 777, 1000, 1500):
    X = rng.uniform(size=(n, 6)); y = np.sin(X @ np.arange(1, 7) / 3.0) + 0.01 * rng.standard_normal(n)
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
    g = MLLObjective(m.train(), True, [0, 0])
    x0 = g.pack_parameters()
    pts = [x0 + 0.05 * rng.standard_normal(x0.shape) for _ in range(5)]
    with settings.graphed_objective(False):
        e = MLLObjective(m, True, [0, 0])
        ref = [e.fun(p) for p in pts]
    g.fun(x0)
    objs.append((n, g, e, pts, ref))
t_end, count = time.time() + budget, 0
while time.time() < t_end:
    n, g, e, pts, ref = objs[count % len(objs)]
    k = int(rng.integers(0, len(pts)))
    if count % 7 == 3:
        e.fun(pts[k]); torch.cuda.synchronize()
    if count % 11 == 5:
        junk = torch.empty(int(rng.integers(1, 50)) << 20, device="cuda"); del junk
    f, gr = g.fun(pts[k])
    assert f == ref[k][0] and np.array_equal(gr, ref[k][1]), (n, count)
    count += 1
print(f"{count} replays at N = 300..1500: all equal to the eager values; declined:", [o[1]._graph.declined for o in objs], flush=True)
set_seed(3)
X = rng.uniform(size=(400, 5)); y = np.cos(X.sum(1))
first = None
for rep in range(12):
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
    set_seed(7)
    f, h = fit_model_torch_batched(m, num_restarts=6, num_iter=60)
    lg = fit_model_torch_batched.last_graph
    assert lg is not None and lg.declined == 0
    if first is None: first = (f, h)
    assert (f, h) == first, rep
    torch.cuda.synchronize()
print("12 batched fits (7 runs x 60 steps, replayed): identical histories", flush=True)
