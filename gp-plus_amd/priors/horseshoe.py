"""Log-half-horseshoe prior on the log noise variance (reference: priors/horseshoe.py:24-79)."""
from numbers import Number

import torch
from torch.distributions import HalfCauchy, HalfNormal
from torch.distributions.utils import broadcast_all

from ..gpcore.module import Prior


class LogHalfHorseshoePrior(Prior):
    """``scale``: horseshoe scale; ``lb``: lower bound of the noise variance on the original scale (default 1e-6).
    ``log_prob`` is the unnormalised spearmint approximation the reference uses (priors/horseshoe.py:60-66)."""

    def __init__(self, scale, lb=1e-6, validate_args=None):
        super().__init__()
        scale_t, lb_t = broadcast_all(scale, lb)
        self.register_buffer("scale", scale_t.to(torch.get_default_dtype()))
        self.register_buffer("lb", lb_t.to(torch.get_default_dtype()))
        self._batch_shape = torch.Size() if isinstance(scale, Number) else self.scale.size()

    def transform(self, x):
        return self.lb.to(x) + torch.exp(x)

    def log_prob(self, X):
        return torch.log(torch.log(1 + 3 * (self.scale.to(X) / self.transform(X)) ** 2)) + X

    def rsample(self, sample_shape=torch.Size([])):
        # priors/horseshoe.py:68-75
        local_shrinkage = HalfCauchy(1).rsample(self.scale.shape).to(self.lb)
        param_sample = HalfNormal(local_shrinkage * self.scale).rsample(sample_shape).to(self.lb)
        lb = self.lb.reshape(-1)[0] if self.lb.numel() > 1 else self.lb
        param_sample = torch.where(param_sample < lb, lb.expand_as(param_sample), param_sample)
        return param_sample.log()

    def expand(self, expand_shape, _instance=None):
        # priors/horseshoe.py:77-79: the reference drops ``lb`` here (restart samples use the 1e-6 default; SURVEY B-7)
        return LogHalfHorseshoePrior(self.scale.expand(torch.Size(expand_shape)))
