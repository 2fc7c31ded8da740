"""Uniform prior with Gaussian tails, differentiable everywhere (reference: priors/mollified_uniform.py:23-95)."""
import math
from numbers import Number

import torch
from torch.distributions import Normal, Uniform
from torch.distributions.utils import broadcast_all

from ..gpcore.module import Prior


class MollifiedUniformPrior(Prior):
    def __init__(self, a, b, tail_sigma=0.1):
        super().__init__()
        a_t, b_t, s_t = broadcast_all(a, b, tail_sigma)
        dt = torch.get_default_dtype()
        self.register_buffer("a", a_t.to(dt))
        self.register_buffer("b", b_t.to(dt))
        self.register_buffer("tail_sigma", s_t.to(dt))
        self._batch_shape = torch.Size() if (isinstance(a, Number) or isinstance(b, Number)) else self.a.size()

    @property
    def mean(self):
        return (self.a + self.b) / 2

    @property
    def _half_range(self):
        return (self.b - self.a) / 2

    @property
    def _log_normalization_constant(self):
        return -torch.log(1 + (self.b - self.a) / (math.sqrt(2 * math.pi) * self.tail_sigma))

    def log_prob(self, X):
        # priors/mollified_uniform.py:79-82
        tail_dist = ((X - self.mean.to(X)).abs() - self._half_range.to(X)).clamp(min=0)
        return Normal(loc=torch.zeros_like(self.a).to(X), scale=self.tail_sigma.to(X)).log_prob(tail_dist) + \
            self._log_normalization_constant.to(X)

    def rsample(self, sample_shape=torch.Size([])):
        return Uniform(self.a, self.b).rsample(sample_shape).to(self.a)

    def expand(self, expand_shape):
        s = torch.Size(expand_shape)
        return MollifiedUniformPrior(self.a.expand(s), self.b.expand(s), self.tail_sigma.expand(s))
