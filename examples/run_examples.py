"""The reference's example workflows (Examples/01, 02, 03 of GP+) through this build's drop-in API on an MI355X: the same calls,
`gpplus` -> `gpplus_amd`, `device='cuda'` (the exact-GP path has no CPU branch here).  Prints the metrics of
`model.evaluation` and the wall time of `fit` (Adam, the reference's cuda branch: gp_plus.py:551-567) and of the evaluation.
usage: python examples/run_examples.py [01] [02] [03] [--scipy]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.models import GP_Plus  # noqa: E402
from gpplus_amd.preprocessing import train_test_split_normalizeX  # noqa: E402
from gpplus_amd.test_functions.analytical import borehole, borehole_mixed_variables  # noqa: E402
from gpplus_amd.test_functions.multi_fidelity import multi_fidelity_wing  # noqa: E402
from gpplus_amd.utils import set_seed  # noqa: E402


def run(name, build, scipy_fit=False):
    t0 = time.perf_counter()
    model, Xtest, ytest = build()
    t1 = time.perf_counter()
    if scipy_fit:
        # the reference's CPU default (gp_plus.py:569-570: L-BFGS-B through optim/mll_scipy.py) with every objective / gradient
        # evaluation on the GPU: what one runs for a converged fit (the cuda branch of fit() is 100 Adam steps per start)
        from gpplus_amd.optim import fit_model_scipy
        fit_model_scipy(model, num_restarts=4)
    else:
        model.fit(n_jobs=-1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    metrics = model.evaluation(Xtest, ytest)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n = model.train_inputs[0].shape[0]
    print(f"[{name}] N_train = {n}, N_test = {len(ytest)}: data + model {t1 - t0:.2f} s, fit {t2 - t1:.2f} s, evaluation {t3 - t2:.2f} s")
    print(f"[{name}] noise variance (original scale) = {(model.likelihood.noise_covar.noise.detach() * model.y_std ** 2).cpu().numpy()}, "
          f"outputscale = {float(model.covar_module.outputscale):.4f}")
    return metrics


def example_01():
    set_seed(1245)
    X, y = borehole(n=10000, random_state=12345)
    Xtrain, Xtest, ytrain, ytest = train_test_split_normalizeX(X, y, test_size=0.95)
    return GP_Plus(Xtrain, ytrain, device='cuda'), Xtest, ytest


def example_02():
    set_seed(4)
    qual_dict = {0: 5, 5: 5}
    U, y = borehole_mixed_variables(n=10000, qual_dict=qual_dict, random_state=4)
    Utrain, Utest, ytrain, ytest = train_test_split_normalizeX(U, y, test_size=0.99, qual_dict=qual_dict)
    return GP_Plus(Utrain, ytrain, qual_dict=qual_dict, device='cuda'), Utest, ytest


def example_03():
    set_seed(4)
    qual_dict = {10: 4}
    num = {'0': 5000, '1': 10000, '2': 10000, '3': 10000}
    noise_std = {'0': 0.5, '1': 1.0, '2': 1.5, '3': 2.0}
    X, y = multi_fidelity_wing(n=num, noise_std=noise_std, random_state=4)
    Xtrain, Xtest, ytrain, ytest = train_test_split_normalizeX(X, y, test_size=0.99, qual_dict=qual_dict,
                                                                stratify=X[..., list(qual_dict.keys())])
    return GP_Plus(Xtrain, ytrain, qual_dict=qual_dict, multiple_noise=True, m_gp='multiple_constant', device='cuda'), Xtest, ytest


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--scipy"]
    todo = args or ["01", "02", "03"]
    for key, fn in (("01", example_01), ("02", example_02), ("03", example_03)):
        if key in todo:
            run("Example " + key + (" (L-BFGS-B)" if "--scipy" in sys.argv else " (fit(): Adam, cuda branch)"), fn, "--scipy" in sys.argv)
