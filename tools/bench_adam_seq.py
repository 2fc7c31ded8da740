"""The sequential Adam driver (fit_model_torch, reference optim/mll_torch.py:104-137) at the examples' sizes, with the evaluation
replayed as a HIP graph and eagerly.  usage: python tools/bench_adam_seq.py [N ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd import settings
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim import fit_model_torch
from gpplus_amd.preprocessing import standard
from gpplus_amd.test_functions.analytical import borehole
from gpplus_amd.utils import set_seed

Xall, yall = borehole(n=10000, random_state=12345)
for N in [int(a) for a in sys.argv[1:]] or [100, 500, 1000, 2000, 3000]:
    X = torch.tensor(Xall[:N]); y = torch.tensor(yall[:N]); X, _, _ = standard(X, {})
    res = {}
    for on in (True, False):
        set_seed(1)
        m = GP_Plus(X, y, dtype=torch.float64, device="cuda")
        with settings.graphed_objective(on):
            fit_model_torch(m, num_restarts=0, num_iter=12, verbose=False)  # one-time costs
            torch.cuda.synchronize(); set_seed(2); t0 = time.perf_counter()
            f, h = fit_model_torch(m, num_restarts=4, num_iter=100, verbose=False)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[on] = (dt, sum(len(x) for x in h), f)
    (tg, eg, fg), (te, ee, fe) = res[True], res[False]
    print("N=%5d: 5 x 100 Adam steps: eager %.3f s = %.2f ms/eval, replayed %.3f s = %.2f ms/eval (incl. capture); best loss %.6f / %.6f"
          % (N, te, 1e3 * te / ee, tg, 1e3 * tg / eg, fe, fg), flush=True)
