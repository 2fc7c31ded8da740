"""Analytic data generators feeding the hot path's configs (reference: test_functions/analytical.py:57-165).
Formulas, bounds and the with-replacement "shuffle" (hazard B-1) follow the reference so the same seeds give the same
data; ``shuffle=False`` gives the unique-row designs the benchmarks use."""
import numpy as np
from scipy.stats.qmc import Sobol, scale

from ..preprocessing import setlevels

BOREHOLE_L = [0.05, 100, 63070, 990, 63.1, 700, 1120, 9855]
BOREHOLE_U = [0.15, 50000, 115600, 1110, 116, 820, 1680, 12045]


def _borehole_formula(X):
    rw, r, Tu, Hu, Tl, Hl, L, Kw = [X[..., i] for i in range(8)]
    frac1 = 2 * np.pi * Tu * (Hu - Hl)
    frac2a = 2 * L * Tu / (np.log(r / rw) * rw ** 2 * Kw)
    frac2b = Tu / Tl
    frac2 = np.log(r / rw) * (1 + frac2a + frac2b)
    return frac1 / frac2


def _sobol_design(n, d, seed, l_bound, u_bound):
    sob = Sobol(d=d, seed=seed)
    X = sob.random(2 ** (np.log2(n) + 1).astype(int))[:n, :]
    return scale(X, l_bounds=l_bound, u_bounds=u_bound)


def borehole(n=100, X=None, noise_std=0.0, random_state=None, shuffle=True):
    if random_state is not None:
        np.random.seed(random_state)
    out_flag = 0
    if X is None:
        X = _sobol_design(n, 8, random_state, BOREHOLE_L, BOREHOLE_U)
        out_flag = 1
    X = np.asarray(X)
    y = _borehole_formula(X)
    if shuffle:
        index = np.random.randint(0, len(y), size=len(y))
        X, y = X[index, ...], y[index]
    if noise_std > 0.0:
        return (X, y + np.random.randn(*y.shape) * noise_std) if out_flag else y
    return (X, y) if out_flag else y


def borehole_mixed_variables(n=100, X=None, qual_dict={0: 5, 6: 3}, noise_std=0.0, random_state=None, shuffle=True):
    out_flag = 0
    if X is None:
        X = _sobol_design(n, 8, random_state, BOREHOLE_L, BOREHOLE_U)
        for key, value in qual_dict.items():
            levels = np.random.uniform(BOREHOLE_L[key], BOREHOLE_U[key], size=value)
            X[..., key] = np.random.choice(levels, size=len(X), replace=True)
        out_flag = 1
    X = np.asarray(X)
    y = _borehole_formula(X)
    if shuffle:
        index = np.random.randint(0, len(y), size=len(y))
        X, y = X[index, ...], y[index]
    X = setlevels(X, qual_index=list(qual_dict.keys()))
    if noise_std > 0.0:
        return (X, y + np.random.randn(*y.shape) * noise_std) if out_flag else y
    return (X, y) if out_flag else y
