import os, sys, cProfile, pstats, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpplus_amd.gpcore import ExactMarginalLogLikelihood
from gpplus_amd.models import GP_Plus
from gpplus_amd.preprocessing import standard
from gpplus_amd.test_functions.analytical import borehole
X, y = borehole(n=10000, random_state=12345); X = torch.tensor(X[:500]); y = torch.tensor(y[:500]); X, _, _ = standard(X, {})
m = GP_Plus(X, y, dtype=torch.float64, device="cuda")
m.train(); mll = ExactMarginalLogLikelihood(m.likelihood, m)
params = [p for p in m.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=0.01)
def step():
    opt.zero_grad()
    loss = -mll(m(*m.train_inputs), m.train_targets); loss.backward(); opt.step(); return loss.item()
for _ in range(20): step()
torch.cuda.synchronize()
import time; t0=time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter()-t0)/200*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
