"""Oracle-driven Adam trajectory for ``fit_model_torch`` (optim/mll_torch.py:99-137): 30 iterations of Adam(lr = 0.01)
on the C1 fixture (reference-pipeline inputs, tests/golden/c1_borehole_n500.npz) from theta1, and 12 iterations on the
mixed-input fixture from its theta1 (so the latent map's parameters move too).  Expected values come from
oracle/gp_oracle.py (``OracleGP.fit_adam``).  Run HERE (CPU):

    python tests/golden/make_trajectory.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.gp_oracle import OracleGP  # noqa: E402


def run(fixture, xkey, tag, iters, **kw):
    fx = dict(np.load(os.path.join(HERE, fixture)))
    o = OracleGP(fx[xkey], fx["ytrain"], **kw)
    for k in list(o.params):
        o.params[k] = torch.as_tensor(fx[f"{tag}::param::{k}"]).reshape(o.params[k].shape).clone()
    hist = o.fit_adam(num_iter=iters, lr=0.01, break_steps=50)
    out = {"loss_hist": np.asarray(hist), "iters": np.array(iters)}
    for k, v in o.params.items():
        out["final::" + k] = v.numpy()
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    for name, (fixture, xkey, iters, kw) in {
        "c1": ("c1_borehole_n500.npz", "Xtrain", 30, {}),
        "mixed": ("c3_borehole_mixed_n100.npz", "Utrain", 12, {"qual_dict": {0: 5, 5: 5}}),
    }.items():
        for k, v in run(fixture, xkey, "theta1", iters, **kw).items():
            out[f"{name}::{k}"] = v
        print(name, out[f"{name}::loss_hist"][[0, -1]])
    np.savez_compressed(os.path.join(HERE, "adam_trajectory.npz"), **out)


if __name__ == "__main__":
    main()
