"""Multi-fidelity wing generators (reference: test_functions/multi_fidelity.py:8-104)."""
import numpy as np
import torch
from scipy.stats.qmc import Sobol, scale

WING_L = [150, 220, 6, -10, 16, 0.5, 0.08, 2.5, 1700, 0.025]
WING_U = [200, 300, 10, 10, 45, 1, 0.18, 6, 2500, 0.08]


def wing(n=100, X=None, fidelity=0, noise_std=0.0, random_state=None, shuffle=True):
    if random_state is not None:
        np.random.seed(random_state)
    out_flag = 0
    if X is None:
        sob = Sobol(d=10, seed=random_state)
        X = sob.random(2 ** (np.log2(n) + 1).astype(int))[:n, :]
        X = scale(X, l_bounds=WING_L, u_bounds=WING_U)
        out_flag = 1
    X = np.asarray(X)
    Sw, Wfw, A = X[..., 0], X[..., 1], X[..., 2]
    Gama = X[..., 3] * (np.pi / 180.0)
    q, lamb, tc, Nz, Wdg, Wp = X[..., 4], X[..., 5], X[..., 6], X[..., 7], X[..., 8], X[..., 9]
    common = Wfw ** 0.0035 * (A / (np.cos(Gama)) ** 2) ** 0.6 * q ** 0.006 * lamb ** 0.04 * \
        ((100 * tc) / (np.cos(Gama))) ** (-0.3) * (Nz * Wdg) ** 0.49
    if fidelity == 0:
        y = 0.036 * Sw ** 0.758 * common + Sw * Wp
    elif fidelity == 1:
        y = 0.036 * Sw ** 0.758 * common + 1 * Wp
    elif fidelity == 2:
        y = 0.036 * Sw ** 0.8 * common + 1 * Wp
    elif fidelity == 3:
        y = 0.036 * Sw ** 0.9 * common + 0 * Wp
    else:
        raise ValueError('only 4 fidelities of 0,1,2,3 have been implemented ')
    # (the reference's shuffle branch is dead code: it tests ``X is None`` after X was assigned, multi_fidelity.py:52-56)
    if noise_std > 0.0:
        return (X, y + np.random.randn(*y.shape) * noise_std) if out_flag else y
    return (X, y) if out_flag else y


def multi_fidelity_wing(X=None, n={'0': 50, '1': 100, '2': 100, '3': 100},
                        noise_std={'0': 0.0, '1': 0.0, '2': 0.0, '3': 0.0}, random_state=None, shuffle=True):
    if X is None:
        X_list, y_list = [], []
        for level, num in n.items():
            if level in ['0', '1', '2', '3'] and num > 0:
                Xl, yl = wing(n=num, fidelity=int(level), noise_std=noise_std[level], random_state=random_state)
                X_list.append(np.hstack([Xl, np.ones(num).reshape(-1, 1) * float(level)]))
                y_list.append(yl)
            else:
                raise ValueError('Wrong label, should be h, l1, l2 or l3')
        return np.vstack(X_list), np.hstack(y_list)
    if isinstance(X, np.ndarray):
        X = torch.tensor(X)
    y_list = []
    for f in n.keys():
        index = [i[0] for i in torch.argwhere(X[..., -1] == int(f))]
        y_list.append(wing(X=X[index, 0:-1], fidelity=int(f), noise_std=noise_std[f]))
    return torch.tensor(np.hstack(y_list))
