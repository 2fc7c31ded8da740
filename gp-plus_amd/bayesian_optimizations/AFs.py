"""Acquisition functions of the reference's multi-fidelity Bayesian optimisation (``bayesian_optimizations/AFs.py:1-159``),
evaluated against this build's GP_Plus: every call is ONE prediction from the model's cached factorisation
(``gpp_cross_kernel`` + ``gpp_predict``; no (N+M)^2 rebuild).

All functions return the NEGATIVE cost-scaled utility for the point-wise versions (they are minimised by scipy,
BO_GP_plus.py:68) and the positive utility for the ``*_Engineering`` (pool-based) versions (maximised by argmax,
BO_GP_plus.py:196).  As in the reference the low-fidelity utility is the exploration part of expected improvement
(sigma * pdf), the high-fidelity utility its exploitation part (sigma * u), and AF_EI the full expected improvement.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.distributions import Normal

__all__ = ["AF_LF", "AF_HF", "AF_EI", "AF_LF_Engineering", "AF_HF_Engineering"]


def _point(samples, xmean, xstd):
    """Reference :4-5: standardise the quantitative coordinates, keep the trailing fidelity column."""
    samples = np.asarray(samples, dtype=np.float64)
    s = np.concatenate([((samples[0:-1] - xmean) / xstd).reshape(1, -1), samples[-1].reshape(-1, 1)], axis=-1)
    return torch.tensor(s.reshape(1, -1))


def _u_sigma_cost(mean, std, best_f, cost, maximize, si):
    mean = mean.reshape(-1, 1)
    view_shape = mean.shape[:-2] if mean.shape[-2] == 1 else mean.shape[:-1]
    mean = mean.view(view_shape)
    sigma = std.view(view_shape)
    u = (mean - best_f - np.sign(best_f) * si) / sigma
    cost = torch.ones(u.shape, dtype=u.dtype) if cost is None else cost.to(u).view(u.shape)
    if not maximize:
        u = -u
    return u, sigma, cost


def _predict_point(samples, model, xmean, xstd, cost_fun):
    x = _point(samples, xmean, xstd)
    with torch.no_grad():
        mean, std = model.predict(x, return_std=True, include_noise=True)
    cost = torch.tensor([float(cost_fun(v)) for v in x[:, -1].clone().detach()])
    return mean.detach().cpu().double(), std.detach().cpu().double(), cost


def AF_LF(samples, best_f, model, xmean, xstd, cost_fun, maximize=False, si=0.0):
    mean, std, cost = _predict_point(samples, model, xmean, xstd, cost_fun)
    u, sigma, cost = _u_sigma_cost(mean, std, best_f, cost, maximize, si)
    updf = torch.exp(Normal(torch.zeros_like(u), torch.ones_like(u)).log_prob(u))
    return float(-1 * (sigma * updf / cost))


def AF_HF(samples, best_f, model, xmean, xstd, cost_fun, maximize=False, si=0.0, data_gen_func=None):
    mean, std, cost = _predict_point(samples, model, xmean, xstd, cost_fun)
    u, sigma, cost = _u_sigma_cost(mean, std, best_f, cost, maximize, si)
    return float(-1 * (sigma * u / cost))


def AF_EI(samples, best_f, model, xmean, xstd, cost_fun, maximize=False, si=0.0):
    mean, std, cost = _predict_point(samples, model, xmean, xstd, cost_fun)
    u, sigma, cost = _u_sigma_cost(mean, std, best_f, cost, maximize, si)
    normal = Normal(torch.zeros_like(u), torch.ones_like(u))
    ei = sigma * (torch.exp(normal.log_prob(u)) + u * normal.cdf(u))
    return float(-1 * (ei / cost))


def AF_LF_Engineering(best_f, mean, std, x_val, cost_fun, maximize=True, si=0.0, cost=None):
    cost = torch.tensor([float(cost_fun(v)) for v in x_val[:, -1].clone().detach()])
    u, sigma, cost = _u_sigma_cost(mean.detach().cpu().double(), std.detach().cpu().double(), best_f, cost, maximize, si)
    updf = torch.exp(Normal(torch.zeros_like(u), torch.ones_like(u)).log_prob(u))
    return sigma * updf / cost


def AF_HF_Engineering(best_f, mean, std, x_val, cost_fun, maximize=True, si=0.0):
    cost = torch.tensor([float(cost_fun(v)) for v in x_val[:, -1].clone().detach()])
    u, sigma, cost = _u_sigma_cost(mean.detach().cpu().double(), std.detach().cpu().double(), best_f, cost, maximize, si)
    return sigma * u / cost
