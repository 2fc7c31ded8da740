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
        num_data = function_dist.event_shape.numel()
        tail = None if params else self._graphed_tail(function_dist)
        if tail is not None:
            tau, grp, prior_sum = tail
            output = MultivariateNormal(function_dist.mean, function_dist.lazy_covariance_matrix.add_diag(tau, grp))
            return (output.log_prob(target) + prior_sum.reshape(())) / num_data
        output = self.likelihood(function_dist, *params)
        res = output.log_prob(target)
        prior_sum = self._prior_sum(res.dtype)
        if prior_sum is not None:
            res = res + prior_sum
        return res / num_data

    def _prior_sum(self, dtype):
        """Sum of the prior log-densities ([3P] ExactMarginalLogLikelihood adds them to the likelihood term one by one; here they
        are summed first, in registration order, and added once — the association the graphed tail uses too, so both give the
        same bits).  None when the model has no priors."""
        total = None
        for _, module, prior, closure, _ in self.named_priors():
            term = prior.log_prob(closure(module)).sum().to(dtype)
            total = term if total is None else total + term
        return total

    def _graphed_tail(self, function_dist):
        """The likelihood's noise transform (models/gpregression.py:59; likelihoods_noise/multifidelity.py:78-136) and the priors
        (priors/*, gpregression.py:84-115) as a replayed pair of HIP graphs (gp-plus_amd/graphed.py::GraphedSegment): (tau, grp,
        sum of prior log-densities), or None where that does not apply."""
        from ..graphed import GraphedSegment, segment_key, segments_apply
        from .kernels import LazyKernelMatrix

        cov = function_dist.lazy_covariance_matrix
        if not isinstance(cov, LazyKernelMatrix) or not cov.is_square or cov.tau is not None:
            return None
        if not segments_apply(function_dist.mean.shape[0], function_dist.mean.device):
            return None
        dev = function_dist.mean.device
        model_params = [p for p in self.model.parameters()]
        if not any(p.requires_grad for p in model_params) or any(p.device != dev for p in model_params):
            return None
        fid = getattr(self.likelihood, "fidel_indices", None)
        key = segment_key(model_params, *([fid] if torch.is_tensor(fid) else [])) + (function_dist.mean.shape[0],)
        st = getattr(self, "_tail_segment", None)
        if st is None or st["key"] != key:
            meta = {}

            def fn():
                noisy = self.likelihood(function_dist).lazy_covariance_matrix
                meta["grp"] = noisy.grp
                total = self._prior_sum(torch.float64)
                if total is None:
                    total = torch.zeros((), dtype=torch.float64, device=dev)
                return noisy.tau.reshape(-1), total.reshape(1)

            st = {"key": key, "seg": None, "meta": meta}
            try:
                st["seg"] = GraphedSegment(fn, model_params, dev, module=self.model)
            except (TypeError, RuntimeError) as exc:
                import warnings

                warnings.warn(f"the likelihood / prior terms could not be captured as a HIP graph ({exc}); evaluating them op by op",
                              RuntimeWarning)
            self._tail_segment = st
        if st["seg"] is None:
            return None
        tau, prior_sum = st["seg"]()
        return tau, st["meta"]["grp"], prior_sum

    def named_priors(self, memo=None, prefix=""):
        # priors of the model (which includes the likelihood's) — the MLL module itself has none
        yield from self.model.named_priors(memo, prefix)
