"""Oracle values for the BASELINE configs at FULL size (C2 N = 20 000, C3 N = 10 000, C4 N = 15 000): loss and every
parameter gradient of ``optim/mll_torch.py:114-117`` from ``oracle/gp_oracle.py`` (autograd through the dense fp64
Cholesky) and — round 4 — the predictive mean / std (with and without noise, models/gpregression.py:122-149) at 256
seeded test points (``baseline_configs.make_test_points``), written to ``tests/golden/fullsize_<cfg>.npz``.  Run HERE (the build container, CPU, minutes per config and
tens of GB of host memory at C2), never on the GPU box:

    python tests/golden/make_fullsize.py C3 C4 C2

The inputs are regenerated on the GPU box by the same ``gpplus_amd.test_functions.baseline_configs.make_config``; the
fixture keeps a checksum of them so that a drift of the generator is reported as such and not as a parity failure.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.gp_oracle import OracleGP  # noqa: E402
from gpplus_amd.test_functions.baseline_configs import make_config, make_test_points  # noqa: E402


def checksum(X, y):
    return np.array([float(X.double().sum()), float((X.double() ** 2).sum()), float(y.double().sum()),
                     float((y.double() ** 2).sum())])


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    for name in (sys.argv[1:] or ["C3", "C4", "C2"]):
        X, y, kw, theta = make_config(name)
        o = OracleGP(X.numpy(), y.numpy(), **kw)
        for k, v in theta.items():
            assert k in o.params, (k, list(o.params))
            o.params[k] = v.reshape(o.params[k].shape).clone()
        t0 = time.perf_counter()
        loss_t, grads = o.loss_and_grad()
        loss = loss_t.clone()
        dt = time.perf_counter() - t0
        out = {"loss": loss.numpy(), "checksum": checksum(X, y), "N": np.array(X.shape[0]), "oracle_seconds": np.array(dt),
               "oracle_threads": np.array(torch.get_num_threads())}
        for k, g in grads.items():
            out["grad::" + k] = g.numpy()
        for k, v in theta.items():
            out["theta::" + k] = v.numpy()
        del loss_t, grads
        t0 = time.perf_counter()
        Xt = make_test_points(name, X)
        mean, std, std0 = o.predict_all(Xt.numpy())
        out.update(pred_mean=mean.numpy(), pred_std=std.numpy(), pred_std_nonoise=std0.numpy(),
                   test_checksum=np.array([float(Xt.sum()), float((Xt ** 2).sum())]),
                   predict_seconds=np.array(time.perf_counter() - t0))
        np.savez_compressed(os.path.join(HERE, f"fullsize_{name.lower()}.npz"), **out)
        print(name, "N", X.shape, "loss", float(loss), f"{dt:.1f} s", "predict", f"{float(out['predict_seconds']):.1f} s",
              "std range", float(std.min()), float(std.max()), "nonoise", float(std0.min()), float(std0.max()), flush=True)


if __name__ == "__main__":
    main()
