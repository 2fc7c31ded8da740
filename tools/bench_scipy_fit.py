"""Time ``fit_model_scipy`` with the replayed objective on and off: evaluations per second of the L-BFGS loop at
the sizes of the reference's examples.  ``python tools/bench_scipy_fit.py [N ...]``"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gpplus_amd import settings  # noqa: E402
from gpplus_amd.models import GP_Plus  # noqa: E402
from gpplus_amd.optim.mll_scipy import MLLObjective, fit_model_scipy  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [100, 500, 1000, 2000, 3000]
dev = torch.device("cuda:0")
for n in sizes:
    rng = np.random.default_rng(n)
    X = rng.uniform(size=(n, 8))
    y = np.sin(X @ np.arange(1, 9) / 3.0) + 0.01 * rng.standard_normal(n)
    y = (y - y.mean()) / y.std()
    line = {"N": n}
    for on in (False, True):
        m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device=dev)
        with settings.graphed_objective(on):
            obj = MLLObjective(m.train(), True, [0, 0])
            x0 = obj.pack_parameters()
            obj.fun(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                obj.fun(x0 + 1e-3 * (i % 7))
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / 200
            torch.manual_seed(0)
            t0 = time.perf_counter()
            res, nll = fit_model_scipy(m, num_restarts=3)
            fit = time.perf_counter() - t0
        key = "replay" if on else "eager"
        line[key + "_ms_per_eval"] = round(per * 1e3, 3)
        line[key + "_fit_s"] = round(fit, 2)
        line[key + "_nfev"] = int(sum(r.nfev for r in res if not isinstance(r, Exception)))
        line[key + "_nll"] = float(nll)
    print(line, flush=True)
