import os, sys, warnings, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpplus_amd.gpcore import ExactMarginalLogLikelihood
from gpplus_amd.models import GP_Plus
from gpplus_amd.preprocessing import standard
from gpplus_amd.test_functions.analytical import borehole_mixed_variables, borehole
from gpplus_amd.test_functions.multi_fidelity import multi_fidelity_wing
import traceback
def run(name, m):
    m.train(); mll = ExactMarginalLogLikelihood(m.likelihood, m)
    params = [p for p in m.parameters() if p.requires_grad]
    def step():
        for p in params: p.grad = None
        loss = -mll(m(*m.train_inputs), m.train_targets); loss.backward(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize()
    print("=====", name)
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step()
    torch.cuda.set_sync_debug_mode("default")
    for x in w:
        print(str(x.message)[:100], "@", x.filename.split("/")[-1], x.lineno)
torch.manual_seed(0); np.random.seed(4); qd = {0: 5, 5: 5}
U, y = borehole_mixed_variables(n=600, qual_dict=qd, random_state=4, shuffle=False)
U, _, _ = standard(torch.as_tensor(U).double(), qd)
run("C3-like", GP_Plus(U, torch.tensor(y), qual_dict=qd, dtype=torch.float64, device="cuda"))
X, y = multi_fidelity_wing(n={'0': 200, '1': 200, '2': 200}, noise_std={'0': 0.5, '1': 1.0, '2': 1.5}, random_state=4)
X, _, _ = standard(torch.tensor(X), {10: 3})
run("C4-like", GP_Plus(X, torch.tensor(y), qual_dict={10: 3}, multiple_noise=True, m_gp='multiple_constant', dtype=torch.float64, device="cuda"))
X, y = borehole(n=600, random_state=1); X, _, _ = standard(torch.tensor(X), {})
run("plain", GP_Plus(X, torch.tensor(y), dtype=torch.float64, device="cuda"))
