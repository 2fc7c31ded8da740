"""Per-stage times of one evaluation through the GP_Plus API for a BASELINE config (dev tool).  usage: config_stages.py C2|C3|C4 [reps]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd import linalg
from gpplus_amd.gpcore import ExactMarginalLogLikelihood
from gpplus_amd.models import GP_Plus
from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
X, y, kw, theta = make_config(cfg)
m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw); apply_theta(m, theta)
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
for _ in range(reps): step()
e1.record(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
ev, linalg.STAGE_EVENTS = linalg.STAGE_EVENTS, None
st = {}
for n, a, b in ev: st.setdefault(n, []).append(a.elapsed_time(b))
N = X.shape[0]
print("%s N=%d: wall %.2f ms/eval; gpu span %.2f" % (cfg, N, wall, e0.elapsed_time(e1) / reps))
tot = 0
for k, v in st.items():
    fl = N ** 3 / 3 if k in ("potrf", "trtri", "lauum") else 0
    print("  %-12s %.3f ms %s" % (k, np.mean(v), ("%.1f TFLOP/s" % (fl / np.mean(v) / 1e9)) if fl else "")); tot += np.mean(v)
print("  sum of stages %.2f" % tot)
kb = [a for n, a, b in ev if n == "kernel_build"]; gr = [b for n, a, b in ev if n == "grad_reduce"]
print("  gap grad_reduce -> next kernel_build: %.3f ms" % np.mean([gr[i].elapsed_time(kb[i + 1]) for i in range(len(kb) - 1)]))
