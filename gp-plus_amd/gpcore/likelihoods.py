"""Gaussian likelihoods (gpytorch.likelihoods subset): ``likelihood(mvn)`` adds the diagonal noise lazily."""
from __future__ import annotations

from typing import Optional

import torch

from .distributions import DenseCovariance, MultivariateNormal
from .kernels import DiagNoise, LazyKernelMatrix
from .module import GreaterThan, Interval, Module


class HomoskedasticNoise(Module):
    """gpytorch.likelihoods.noise_models.HomoskedasticNoise: raw_noise (num_noises,), ``noise`` through the constraint."""

    def __init__(self, noise_prior=None, noise_constraint: Optional[Interval] = None, batch_shape=torch.Size(), num_tasks: int = 1):
        super().__init__()
        if noise_constraint is None:
            noise_constraint = GreaterThan(1e-4)
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(*batch_shape, num_tasks)))
        self.register_constraint("raw_noise", noise_constraint)
        if noise_prior is not None:
            self.register_prior("noise_prior", noise_prior, lambda m: m.noise, lambda m, v: m._set_noise(v))

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)

    @noise.setter
    def noise(self, value):
        self._set_noise(value)

    def _set_noise(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value)
        value = value.to(self.raw_noise)
        raw = self.raw_noise_constraint.inverse_transform(value)
        self.initialize(raw_noise=raw.expand_as(self.raw_noise) if raw.numel() == 1 else raw.reshape(self.raw_noise.shape))

    def forward(self, n: int, grp: Optional[torch.Tensor] = None) -> DiagNoise:
        return DiagNoise(self.noise, grp, n)


class _GaussianLikelihoodBase(Module):
    def __init__(self, noise_covar):
        super().__init__()
        self.noise_covar = noise_covar

    def _noise_for(self, mvn: MultivariateNormal) -> DiagNoise:
        return self.noise_covar(mvn.loc.shape[0], None)

    def marginal(self, function_dist: MultivariateNormal, *params, **kwargs) -> MultivariateNormal:
        covar = function_dist.lazy_covariance_matrix
        noise = self._noise_for(function_dist)
        if torch.is_tensor(covar):
            full = covar.clone()
            full.diagonal().add_(noise.diag().to(full))
        elif isinstance(covar, (LazyKernelMatrix, DenseCovariance)):
            full = covar + noise
        else:
            raise TypeError(type(covar))
        return function_dist.__class__(function_dist.mean, full)

    def forward(self, function_samples, *params, **kwargs):
        raise NotImplementedError("sampling likelihoods are outside the exact-GP hot path")

    def __call__(self, input, *params, **kwargs):
        if isinstance(input, MultivariateNormal):
            return self.marginal(input, *params, **kwargs)
        return self.forward(input, *params, **kwargs)


class GaussianLikelihood(_GaussianLikelihoodBase):
    """gpytorch GaussianLikelihood (models/gpregression.py:63): one homoskedastic noise."""

    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__(HomoskedasticNoise(noise_prior, noise_constraint, batch_shape, num_tasks=1))

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.initialize(noise=value)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    @raw_noise.setter
    def raw_noise(self, value):
        self.noise_covar.initialize(raw_noise=value)
