"""gpytorch.means subset used by models/gp_plus.py:488-507."""
import torch

from .module import Module


class Mean(Module):
    pass


class ConstantMean(Mean):
    """Raw parameter ``constant`` of shape (1,) as in gpytorch <= 1.8 (state_dict key ``mean_module.constant``,
    models/gp_plus.py:971-972); optional prior registered as ``mean_prior`` on it."""

    def __init__(self, prior=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self.register_parameter("constant", torch.nn.Parameter(torch.zeros(*batch_shape, 1)))
        if prior is not None:
            self.register_prior("mean_prior", prior, "constant")

    def forward(self, input):
        if input.shape[:-2] == self.batch_shape:
            return self.constant.expand(input.shape[:-1])
        return self.constant.expand(*input.shape[:-1])


class ZeroMean(Mean):
    def forward(self, input):
        return torch.zeros(input.shape[:-1], dtype=input.dtype, device=input.device)
