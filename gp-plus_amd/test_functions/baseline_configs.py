"""Synthetic inputs of the BASELINE.json configs C1..C5, shared by ``bench.py``, the full-size parity
tests and ``tests/golden/make_fullsize.py`` so that all three evaluate the SAME numbers.  Data come from this package's
restatements of the reference generators (test_functions/analytical.py:57-165, test_functions/multi_fidelity.py:8-104,
preprocessing/normalizeX.py:53-72); everything is plain numpy / CPU torch — no GPU, no oracle.

``make_config(name)`` returns ``(X, y, model_kwargs, theta)``: fp64 CPU tensors, the keyword arguments of ``GP_Plus`` and
the evaluation point as a dict ``state_dict key -> value`` (values are fp32-representable).  The manifold
matrix of the mixed-input configs is part of ``theta`` (drawn once from a seeded generator) because ``GP_Plus`` draws
its own from torch's global RNG.
"""
import numpy as np
import torch

from ..preprocessing import standard
from .analytical import borehole, borehole_mixed_variables
from .multi_fidelity import multi_fidelity_wing

SIZES = {"C1": 500, "C2": 20000, "C3": 10000, "C4": 15000, "C5": 60000}


def _f32(x):
    return torch.as_tensor(np.asarray(np.float32(x), dtype=np.float64))


def _latent(seed, dz, nlev):
    g = torch.Generator().manual_seed(seed)
    return _f32(torch.randn(dz, nlev, generator=g, dtype=torch.float64).numpy())


def make_config(name: str, n: int = None):
    n = SIZES[name] if n is None else int(n)
    if name in ("C1", "C2"):
        # C2: Sobol(d=8, seed 0) scaled to the Borehole bounds, no shuffle (unique rows), z-scored, y = Borehole.
        # C1 at this entry point is the same generator at N = 500 (the reference-pipeline C1 is the golden fixture).
        X, y = borehole(n=n, random_state=0, shuffle=False)
        X, _, _ = standard(torch.tensor(X), {})
        kw = {}
        theta = {"covar_module.base_kernel.raw_lengthscale": _f32(np.full((1, 8), -1.0)),
                 "covar_module.raw_outputscale": _f32(0.3), "likelihood.noise_covar.raw_noise": _f32([-6.0]),
                 "mean_module.constant": _f32([0.4])}
        return X.double(), torch.tensor(y).double(), kw, theta
    if name == "C3":
        np.random.seed(4)
        qd = {0: 5, 5: 5}
        U, y = borehole_mixed_variables(n=n, qual_dict=qd, random_state=4, shuffle=False)
        U, _, _ = standard(torch.as_tensor(U).double(), qd)
        theta = {"covar_module.base_kernel.kernels.1.raw_lengthscale": _f32(np.full((1, 6), -1.0)),
                 "covar_module.raw_outputscale": _f32(0.3), "likelihood.noise_covar.raw_noise": _f32([-6.0]),
                 "mean_module.constant": _f32([0.4]), "latent[0, 5]": _latent(0, 2, 10)}
        return U.double(), torch.tensor(y).double(), {"qual_dict": qd}, theta
    if name == "C4":
        per = n // 3
        X, y = multi_fidelity_wing(n={'0': per, '1': per, '2': n - 2 * per}, noise_std={'0': 0.5, '1': 1.0, '2': 1.5},
                                   random_state=4)
        X, _, _ = standard(torch.tensor(X), {10: 3})
        kw = {"qual_dict": {10: 3}, "multiple_noise": True, "m_gp": "multiple_constant"}
        theta = {"covar_module.base_kernel.kernels.1.raw_lengthscale": _f32(np.full((1, 10), -1.0)),
                 "covar_module.raw_outputscale": _f32(0.3),
                 "likelihood.noise_covar.raw_noise": _f32(np.log([1e-4, 4e-4, 9e-4])),
                 "mean_module_1.constant": _f32([0.1]), "mean_module_2.constant": _f32([-0.2]),
                 "latent[10]": _latent(1, 2, 3)}
        return X.double(), torch.tensor(y).double(), kw, theta
    if name == "C5":
        from scipy.stats.qmc import Sobol

        Xs = Sobol(d=16, seed=0).random(2 ** int(np.ceil(np.log2(n))))[:n]
        Xs = (Xs - Xs.mean(0)) / Xs.std(0)
        rng = np.random.default_rng(0)
        y = np.sin(Xs).sum(1) + 1e-2 * rng.standard_normal(n)
        theta = {"covar_module.base_kernel.raw_lengthscale": _f32(np.full((1, 16), -1.5)),
                 "covar_module.raw_outputscale": _f32(0.3),
                 "likelihood.noise_covar.raw_noise": _f32([np.log(1e-3)]), "mean_module.constant": _f32([0.0])}
        return torch.tensor(Xs).double(), torch.tensor(y).double(), {}, theta
    raise ValueError(name)


def make_test_points(name: str, X: torch.Tensor, m: int = 256) -> torch.Tensor:
    """``m`` seeded prediction points for config ``name``: training rows (seeded choice) whose quantitative columns are
    displaced by Gaussian steps of 0.02 ... 1.0 standard deviations (the step size cycles over five values), so the
    predictive variances run from noise-dominated to near-prior; categorical / source columns keep their levels.
    Shared by tests/golden/make_fullsize.py and the full-size parity test (models/gpregression.py:122-149)."""
    qual = {"C3": (0, 5), "C4": (10,)}.get(name, ())
    rng = np.random.default_rng(1234 + sum(map(ord, name)))
    rows = rng.choice(X.shape[0], size=m, replace=False)
    Xt = X[torch.as_tensor(rows)].clone().double()
    step = np.array([0.02, 0.1, 0.3, 0.6, 1.0])[np.arange(m) % 5][:, None] * rng.standard_normal((m, X.shape[1]))
    for c in qual:
        step[:, c] = 0.0
    return Xt + torch.as_tensor(step)


def apply_theta(model, theta) -> None:
    """Load the evaluation point into a ``GP_Plus`` (keys are the reference's state_dict names)."""
    sd = model.state_dict()
    for k, v in theta.items():
        if k not in sd:
            raise KeyError(f"model has no parameter {k}: {sorted(sd)}")
        sd[k] = v.reshape(sd[k].shape).to(sd[k])
    model.load_state_dict(sd)
