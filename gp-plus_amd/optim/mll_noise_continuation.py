"""Noise-continuation fit and leave-one-out cross-validation error.

Mirrors ``optim/mll_noise_continuation.py`` of the reference: ``loocv_rrmse`` (:28-42) and
``fit_model_continuation`` (:45-244), a sequence of ``fit_model_scipy`` runs at fixed, decreasing noise variances.
Every likelihood evaluation inside runs on the MI355X back end (``linalg.ExactMLLFunction``); the LOOCV error uses
the same factorisation caches as prediction plus ``gpp_lauum`` for diag(Ky^-1).

The reference's control flow is reproduced as written, including its quirks (documented inline): the LOOCV history is
a list of the string 'NLL' (the call is commented out at :186), the ``red_factor`` re-initialisation at :209-216 is
overwritten by the next pass of the loop, and the refinement passes interpolate linearly between the neighbours of the
best noise level until the best level stops moving by more than ``accuracy``.
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Dict, Tuple

import numpy as np
import torch
from scipy.spatial import distance_matrix

from .mll_scipy import fit_model_scipy

__all__ = ["loocv_rrmse", "fit_model_continuation"]


def loocv_rrmse(model) -> float:
    """Root-mean-square leave-one-out error  sqrt(mean((Ky^-1 r)_i / (Ky^-1)_ii)^2)  (reference :28-42).

    The reference reads gpytorch's prediction caches (``mean_cache`` = Ky^-1 r, ``covar_cache`` = R with R R^T = Ky^-1);
    here the cached factorisation of ``ExactGP`` provides Ky^-1 r and L^-1, and ``gpp_lauum`` forms Ky^-1 = L^-T L^-1
    whose diagonal is needed."""
    from ..linalg import get_workspace
    model.eval()
    with torch.no_grad():
        cache = model._ensure_prediction_cache()
        N = cache.U.shape[0]
        ws = get_workspace(cache.gctx, N, slot=-1)  # the buffers the cache lives in: Ki is free scratch
        cache.gctx.lauum(cache.Linv, ws.Ki)
        kinv_diag = ws.Ki.diagonal().clone()
        loo_error = cache.alpha / kinv_diag
        return (loo_error ** 2).mean().sqrt().item()


def fit_model_continuation(model, add_prior: bool = True, num_restarts: int = 32, criterion: str = 'NLL',
                           initial_noise_var: float = 1, red_factor: float = math.sqrt(10), options: Dict = {},
                           n_jobs: int = -1, accuracy=1e-2, method='L-BFGS-B', constraint=False,
                           regularization_parameter=[0, 0], bounds=False, verbose: bool = True) -> Tuple[float, Dict]:
    """Optimise the (penalised) likelihood for a decreasing sequence of FIXED noise variances (reference :45-244).

    Returns ``(nll at the selected noise level, history)`` with ``history`` = {'noise_history', 'nll_history',
    'loocv_history', 'optimization_history'} of the LAST pass, and leaves the model at the selected state."""
    if criterion.upper() not in ['NLL', 'LOOCV']:
        raise AttributeError('criterion must be one of NLL or LOOCV')
    if red_factor < 2:
        raise RuntimeError('Reduction factor for noise variance needs to be greater then 2')
    if model.likelihood.raw_noise.requires_grad:
        model.likelihood.raw_noise.requires_grad_(False)

    t = 0
    theta0_list = None
    index = None
    history = None
    old_state_dict: Dict[int, dict] = {}
    while True:
        t += 1
        initial_noise_var_new = initial_noise_var
        if t == 1:
            noises = [initial_noise_var_new / (10 ** i) for i in range(int(10 / t))]
        else:
            n_hist = len(history['noise_history'])
            if (index >= 2 and index < n_hist - 2) or (index >= 1 and index < n_hist - 1):
                # refine between the neighbours of the best level (both reference branches :147-154 do the same thing)
                noises = np.linspace(float(history['noise_history'][index - 1]), float(history['noise_history'][index + 1]), 10)
                initial_noise_var = float(history['noise_history'][index - 1])
                model.load_state_dict(old_state_dict[index - 1])
            else:
                model.load_state_dict(old_state_dict[index])
                if verbose:
                    print(f"Negative log likelihood={history['nll_history'][index]}")
                return history['nll_history'][index], history

        noise_list, nll_list, loocv_list, reslist_list = [], [], [], []
        t += 1  # (the reference increments twice per pass, :140 and :167; only t == 1 is ever tested)
        old_state_dict = {}
        for i in range(len(noises)):
            model.train()
            model.likelihood.initialize(**{'noise': float(noises[i])})
            old_state_dict[i] = deepcopy(model.state_dict())
            reslist, nll = fit_model_scipy(model, add_prior, num_restarts=num_restarts, theta0_list=theta0_list, options=options,
                                           n_jobs=n_jobs, method=method, constraint=constraint,
                                           regularization_parameter=regularization_parameter, bounds=bounds)
            if all(isinstance(res, (RuntimeError, TypeError)) for res in reslist):
                break  # every start failed (singular covariance at this noise level)
            noise_list.append(model.likelihood.noise.data.clone())
            nll_list.append(nll)
            loocv_list.append('NLL')  # reference :186 — the loocv_rrmse(model) call is commented out
            reslist_list.append(reslist)
            # distinct optima of this level start the next one (:201-209)
            theta0_list = []
            for res in reslist:
                if isinstance(res, Exception):
                    continue
                if len(theta0_list) > 0:
                    dists = distance_matrix(res.x.reshape(1, -1), np.vstack(theta0_list)).ravel()
                    if np.any(dists < 1e-2 * res.x.shape[0]):
                        continue
                theta0_list.append(res.x)
            try:
                model.likelihood.initialize(**{'noise': float(noise_list[-1].reshape(-1)[0]) / red_factor})
            except Exception:
                try:
                    model.likelihood.initialize(**{'noise': float(noise_list[-1].reshape(-1)[0]) / red_factor + 1e-10})
                except Exception:
                    break
        if not nll_list:
            raise RuntimeError('fit_model_continuation: every start failed at the first noise level')
        history = {'noise_history': noise_list, 'nll_history': nll_list, 'loocv_history': loocv_list,
                   'optimization_history': reslist_list}
        index = int(np.argmin(history['nll_history']))
        if verbose:
            print('Finished for loop')
            print(history['nll_history'])
        if abs(initial_noise_var_new - float(history['noise_history'][index].reshape(-1)[0])) < accuracy:
            model.load_state_dict(old_state_dict[index])
            break
    return history['nll_history'][index], history
