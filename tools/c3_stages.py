import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpplus_amd import linalg
from gpplus_amd.gpcore import ExactMarginalLogLikelihood
from gpplus_amd.models import GP_Plus
from gpplus_amd.preprocessing import standard
from gpplus_amd.test_functions.analytical import borehole_mixed_variables
torch.manual_seed(0); np.random.seed(4); qd = {0: 5, 5: 5}
U, y = borehole_mixed_variables(n=10000, qual_dict=qd, random_state=4, shuffle=False)
U, _, _ = standard(torch.as_tensor(U).double(), qd)
m = GP_Plus(U, torch.tensor(y), qual_dict=qd, dtype=torch.float64, device="cuda")
m.train(); mll = ExactMarginalLogLikelihood(m.likelihood, m)
params = [p for p in m.parameters() if p.requires_grad]
def step():
    for p in params: p.grad = None
    loss = -mll(m(*m.train_inputs), m.train_targets); loss.backward(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
linalg.STAGE_EVENTS = []
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
ev, linalg.STAGE_EVENTS = linalg.STAGE_EVENTS, None
st = {}
for n, a, b in ev: st.setdefault(n, []).append(a.elapsed_time(b))
print("wall %.2f ms/eval; gpu span %.2f" % (wall, e0.elapsed_time(e1) / 10))
tot = 0
for k, v in st.items():
    print("  %-12s %.3f" % (k, np.mean(v))); tot += np.mean(v)
print("  sum of stages %.2f" % tot)
# gaps: time between end of grad_reduce(k) and start of kernel_build(k+1)
kb = [a for n, a, b in ev if n == "kernel_build"]; gr = [b for n, a, b in ev if n == "grad_reduce"]
print("  gap grad_reduce -> next kernel_build: %.3f ms" % np.mean([gr[i].elapsed_time(kb[i + 1]) for i in range(len(kb) - 1)]))
