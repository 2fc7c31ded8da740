"""Generates the golden fixtures in this directory.  Run HERE (the build container), never on the GPU box:

    python tests/golden/make_golden.py

INPUTS come from the reference's own, importable data pipeline (imported read-only from /root/reference through a
temporary ``gpplus`` alias; no reference source is copied):
    gpplus.test_functions.analytical.borehole / borehole_mixed_variables   (test_functions/analytical.py:57-165)
    gpplus.preprocessing.train_test_split_normalizeX                       (preprocessing/split.py:7-48)
    gpplus.utils.set_seed                                                  (utils/set_seed.py:6-15)
exactly as Examples/01 (cell 3) and Examples/02 (cell 3) call them.
EXPECTED OUTPUTS (loss, gradients, predictions) come from oracle/gp_oracle.py, because the reference's model code
cannot be imported here (gpytorch/botorch are absent) — see the oracle header ("parity unpinned").
Fixtures are plain .npz data: arrays of inputs and expected outputs only.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.gp_oracle import OracleGP  # noqa: E402


def _import_reference():
    tmp = tempfile.mkdtemp(prefix="refalias_")
    os.symlink("/root/reference", os.path.join(tmp, "gpplus"))
    sys.path.insert(0, tmp)
    from gpplus.preprocessing import train_test_split_normalizeX  # noqa
    from gpplus.test_functions.analytical import borehole, borehole_mixed_variables  # noqa
    from gpplus.utils import set_seed  # noqa

    return set_seed, borehole, borehole_mixed_variables, train_test_split_normalizeX


def f32(x):
    """fp32-representable fp64 values (reference parameters are fp32)."""
    return torch.tensor(np.float32(x), dtype=torch.float64)


def set_theta(o: OracleGP, omega, raw_os, raw_noise, consts):
    p = o.params
    if o.ls_key:
        p[o.ls_key] = torch.full_like(p[o.ls_key], float(np.float32(omega)))
    p["covar_module.raw_outputscale"] = f32(raw_os)
    p["likelihood.noise_covar.raw_noise"] = torch.as_tensor(np.float32(raw_noise), dtype=torch.float64).reshape(-1)
    for k, v in consts.items():
        p[k] = torch.as_tensor(np.float32([v]), dtype=torch.float64)


def outputs(o: OracleGP, xtest):
    loss, g = o.loss_and_grad()
    mean, std = o.predict(xtest, return_std=True, include_noise=True)
    mean2, std2 = o.predict(xtest, return_std=True, include_noise=False)
    out = {"loss": loss.numpy(), "mll": o.mll().detach().numpy(), "pred_mean": mean.numpy(), "pred_std": std.numpy(),
           "pred_std_nonoise": std2.numpy()}
    for k, v in g.items():
        out["grad::" + k] = v.numpy()
    for k, v in o.params.items():
        out["param::" + k] = v.numpy()
    return out


def main():
    set_seed, borehole, borehole_mixed, split = _import_reference()

    # ---- C1: Examples/01 cell 3 verbatim -------------------------------------------------------------
    set_seed(1245)
    X, y = borehole(n=10000, random_state=12345)
    Xtrain, Xtest, ytrain, ytest = split(X, y, test_size=0.95)
    Xtrain, Xtest = Xtrain.double().numpy(), Xtest.double().numpy()[:200]
    ytrain, ytest = ytrain.double().numpy(), ytest.double().numpy()[:200]
    fx = {"Xtrain": Xtrain, "ytrain": ytrain, "Xtest": Xtest, "ytest": ytest}
    for tag, (om, ros, rn, c) in {"theta0": (0.0, 0.0, [0.0], 0.0), "theta1": (-1.0, 0.3, [-6.0], 0.4)}.items():
        o = OracleGP(Xtrain, ytrain)
        set_theta(o, om, ros, rn, {"mean_module.constant": c})
        for k, v in outputs(o, Xtest).items():
            fx[f"{tag}::{k}"] = v
    np.savez_compressed(os.path.join(HERE, "c1_borehole_n500.npz"), **fx)
    print("c1: N", Xtrain.shape, "unique rows", len(np.unique(Xtrain, axis=0)), "loss0", fx["theta0::loss"], "loss1", fx["theta1::loss"])

    # ---- Example 02 (mixed inputs), cell 3 verbatim: N=100 train ------------------------------------------
    set_seed(4)
    qual_dict = {0: 5, 5: 5}
    U, y = borehole_mixed(n=10000, qual_dict=qual_dict, random_state=4)
    Utrain, Utest, ytrain, ytest = split(U, y, test_size=0.99, qual_dict=qual_dict)
    Utrain, Utest = Utrain.double().numpy(), Utest.double().numpy()[:200]
    ytrain, ytest = ytrain.double().numpy(), ytest.double().numpy()[:200]
    fx = {"Utrain": Utrain, "ytrain": ytrain, "Utest": Utest, "ytest": ytest}
    o = OracleGP(Utrain, ytrain, qual_dict=qual_dict, seed=0)
    o.params[o.latent_key] = torch.as_tensor(np.float32(o.params[o.latent_key].numpy()), dtype=torch.float64)
    set_theta(o, -1.0, 0.3, [-6.0], {"mean_module.constant": 0.4})
    for k, v in outputs(o, Utest).items():
        fx[f"theta1::{k}"] = v
    np.savez_compressed(os.path.join(HERE, "c3_borehole_mixed_n100.npz"), **fx)
    print("mixed: N", Utrain.shape, "loss", fx["theta1::loss"])

    # ---- C4-shaped multi-fidelity fixture (wing formulas restated in the product; the reference's module needs pyDOE).
    # Inputs are synthetic here: Sobol points, 3 sources, source column appended (test_functions/multi_fidelity.py:77-104).
    from scipy.stats.qmc import Sobol, scale

    l_bound = [150, 220, 6, -10, 16, 0.5, 0.08, 2.5, 1700, 0.025]
    u_bound = [200, 300, 10, 10, 45, 1, 0.18, 6, 2500, 0.08]
    rng = np.random.default_rng(4)
    Xs, ys = [], []
    for lvl, (n, sd) in enumerate([(120, 0.5), (120, 1.0), (120, 1.5)]):
        Xq = scale(Sobol(d=10, seed=4 + lvl).random(128)[:n], l_bounds=l_bound, u_bounds=u_bound)
        Sw, Wfw, A, Gam, q, lamb, tc, Nz, Wdg, Wp = [Xq[:, i] for i in range(10)]
        Gam = Gam * np.pi / 180.0
        e = [0.758, 0.758, 0.8][lvl]
        tail = [Sw * Wp, 1 * Wp, 1 * Wp][lvl]
        yy = 0.036 * Sw**e * Wfw**0.0035 * (A / np.cos(Gam) ** 2) ** 0.6 * q**0.006 * lamb**0.04 * \
            (100 * tc / np.cos(Gam)) ** (-0.3) * (Nz * Wdg) ** 0.49 + tail
        Xs.append(np.hstack([Xq, np.full((n, 1), float(lvl))]))
        ys.append(yy + rng.standard_normal(n) * sd)
    X, y = np.vstack(Xs), np.hstack(ys)
    Xq = X[:, :10]
    X[:, :10] = (Xq - Xq.mean(0)) / Xq.std(0)  # preprocessing/normalizeX.py:53-72 (population std)
    perm = rng.permutation(len(y))
    X, y = X[perm], y[perm]
    Xtr, ytr, Xte = X[:300], y[:300], X[300:]
    o = OracleGP(Xtr, ytr, qual_dict={10: 3}, multiple_noise=True, m_gp="multiple_constant", seed=1)
    o.params[o.latent_key] = torch.as_tensor(np.float32(o.params[o.latent_key].numpy()), dtype=torch.float64)
    set_theta(o, -1.0, 0.3, np.log([1e-4, 4e-4, 9e-4]), {"mean_module_1.constant": 0.1, "mean_module_2.constant": -0.2})
    fx = {"Xtrain": Xtr, "ytrain": ytr, "Xtest": Xte}
    for k, v in outputs(o, Xte).items():
        fx[f"theta1::{k}"] = v
    np.savez_compressed(os.path.join(HERE, "c4_wing_mf_n300.npz"), **fx)
    print("mf: N", Xtr.shape, "loss", fx["theta1::loss"])


if __name__ == "__main__":
    main()
