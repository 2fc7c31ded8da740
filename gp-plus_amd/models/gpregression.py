"""``GPR``: standard GP regression model of GP+ (reference: models/gpregression.py:38-221), on the own gpcore protocol
and the HIP back end.  Same constructor, attributes, buffers and state_dict keys; ``posterior``/``fantasize`` (botorch
glue for Bayesian optimisation) are outside the exact-GP hot path and raise.
"""
import math
from typing import List, Tuple, Union

import torch

from .. import kernels
from ..gpcore import (ExactGP, GaussianLikelihood, GreaterThan, Kernel, LogNormalPrior, MultivariateNormal, Positive,
                      settings as gptsettings)
from ..likelihoods_noise.multifidelity import Multifidelity_likelihood
from ..priors import LogHalfHorseshoePrior, MollifiedUniformPrior
from ..utils.transforms import inv_softplus, softplus


def _require(cond: bool, message: str) -> None:
    if not cond:
        raise RuntimeError(message)


def _named_correlation(name: str, n_features: int) -> Kernel:
    """A kernel of ``kernels`` chosen by name: ARD over every input column, ``l = exp(raw)``, mollified-uniform prior on
    ``raw`` over [log 0.1, log 10] (models/gpregression.py:89-102; any failure is reported as the reference reports it)."""
    try:
        k = getattr(kernels, name)(ard_num_dims=n_features,
                                   lengthscale_constraint=Positive(transform=torch.exp, inv_transform=torch.log))
        k.register_prior('lengthscale_prior', MollifiedUniformPrior(math.log(0.1), math.log(10)), 'raw_lengthscale')
    except Exception:
        raise RuntimeError("%s not an allowed kernel" % name)
    return k


class GPR(ExactGP):
    #: registered on the model in this order (state_dict keys of the reference, models/gpregression.py:73-76)
    _TARGET_BUFFERS = ('y_min', 'y_std', 'y_scaled')

    def __init__(self, train_x: torch.Tensor, train_y: torch.Tensor, correlation_kernel, noise_indices: List[int],
                 fix_noise: bool = False, fix_noise_val: float = 1e-5, lb_noise: float = 1e-12) -> None:
        # models/gpregression.py:50-56
        _require(torch.is_tensor(train_x), "'train_x' must be a tensor")
        _require(torch.is_tensor(train_y), "'train_y' must be a tensor")
        _require(train_x.shape[0] == train_y.shape[0], "Inputs and output have different number of observations")

        # tau = exp(raw) + lb_noise; one level, or one per data source named in the last input column (:59-66)
        noise_args = dict(noise_constraint=GreaterThan(lb_noise, transform=torch.exp, inv_transform=torch.log))
        if noise_indices:
            likelihood = Multifidelity_likelihood(noise_indices=noise_indices, fidel_indices=train_x[:, -1], **noise_args)
        else:
            likelihood = GaussianLikelihood(**noise_args)

        # targets scaled to [0, 1] by their range (:67-69); the three quantities travel with the state_dict
        lo = train_y.min()
        span = train_y.max() - lo
        scaled = (train_y - lo) / span
        ExactGP.__init__(self, train_x, scaled, likelihood)
        for key, value in zip(self._TARGET_BUFFERS, (lo, span, scaled)):
            self.register_buffer(key, value)
        self._num_outputs = 1

        # priors in the reference's registration order: noise, lengthscale (named kernels only), outputscale — this is the
        # order reset_parameters() consumes random numbers in (:84, :97-99, :113-115)
        self.likelihood.register_prior('noise_prior', LogHalfHorseshoePrior(0.01, lb_noise), 'raw_noise')
        if fix_noise:
            self.likelihood.raw_noise.requires_grad_(False)
            self.likelihood.noise_covar.noise = torch.tensor(fix_noise_val)

        if isinstance(correlation_kernel, str):
            correlation_kernel = _named_correlation(correlation_kernel, self.train_inputs[0].size(1))
        _require(isinstance(correlation_kernel, Kernel),
                 "specified correlation kernel is not a `gpytorch.kernels.Kernel` instance")
        self.covar_module = kernels.ScaleKernel(
            base_kernel=correlation_kernel, outputscale_constraint=Positive(transform=softplus, inv_transform=inv_softplus))
        self.covar_module.register_prior('outputscale_prior', LogNormalPrior(1e-6, 1.), 'outputscale')

    # a plain GPR has no mean module in the reference either (forward uses self.mean_module set by subclasses)
    def forward(self, x: torch.Tensor) -> MultivariateNormal:
        mean_x = self.mean_module(x)
        covar_x = self.covar_module(x)
        return MultivariateNormal(mean_x, covar_x)

    def predict(self, x: torch.Tensor, return_std: bool = False, include_noise: bool = False
                ) -> Union[torch.Tensor, Tuple[torch.Tensor]]:
        """models/gpregression.py:122-149."""
        self.eval()
        with gptsettings.fast_computations(log_prob=False):
            if self.train_targets.ndim != 1:
                raise NotImplementedError("batched GPs are outside the exact-GP hot path")
            output = self(x)
            self.fidel_indices = x[:, -1]
            if return_std and include_noise:
                self.likelihood.fidel_indices = x[:, -1]  # noise of the test points' own sources
                output = self.likelihood(output)
            out_mean = self.y_min + self.y_std * output.mean
            if return_std:
                out_std = output.variance.sqrt() * self.y_std
                return out_mean, out_std
            return out_mean

    def posterior(self, X, output_indices=None, observation_noise=True, posterior_transform=None, **kwargs):
        raise NotImplementedError("botorch posterior glue (models/gpregression.py:151-166) is out of scope of this build")

    def fantasize(self, X, sampler, observation_noise=True, **kwargs):
        raise NotImplementedError("botorch fantasize glue (models/gpregression.py:177-221) is out of scope of this build")

    def reset_parameters(self) -> None:
        """Reset parameters by sampling from their priors (models/gpregression.py:168-174)."""
        # The reference builds its priors from Python numbers (torch's default dtype, float32) and never casts them on its
        # CPU path, so every draw is made in that dtype and converted afterwards — whatever ``dtype`` the model was given.
        # Here a model is created in its dtype, buffers of the priors included; the expanded copy is put back first so
        # that one seed gives the start points the reference would draw (tests/test_restart_sampling.py).
        draw_dtype = torch.get_default_dtype()
        for _, module, prior, closure, setting_closure in self.named_priors():
            current = closure(module)
            if not current.requires_grad:
                continue  # (consumes no random numbers)
            sampler = prior.expand(current.shape).to(dtype=draw_dtype)
            setting_closure(module, sampler.sample().to(**self.tkwargs))
