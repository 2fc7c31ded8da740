"""``ExactMarginalLogLikelihood`` (gpytorch.mlls subset; reference use: optim/mll_torch.py:17,96,116):
    (log p(y | X) + sum of prior log-densities) / N
"""
import torch

from .distributions import MultivariateNormal
from .module import Module


class ExactMarginalLogLikelihood(Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def forward(self, function_dist: MultivariateNormal, target: torch.Tensor, *params):
        if not isinstance(function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        output = self.likelihood(function_dist, *params)
        res = output.log_prob(target)
        for _, module, prior, closure, _ in self.named_priors():
            res = res + prior.log_prob(closure(module)).sum().to(res)
        num_data = function_dist.event_shape.numel()
        return res / num_data

    def named_priors(self, memo=None, prefix=""):
        # priors of the model (which includes the likelihood's) — the MLL module itself has none
        yield from self.model.named_priors(memo, prefix)
