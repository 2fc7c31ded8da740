"""``MultivariateNormal`` with a lazy covariance (gpytorch.distributions.MultivariateNormal subset).

``log_prob`` is the hot operator: for a :class:`LazyKernelMatrix` covariance it runs the fused HIP pipeline of
``linalg.exact_mll`` (reference: [3P] MultivariateNormal.log_prob -> inv_quad_logdet reached from
optim/mll_torch.py:116); dense covariances (predictive distributions) go through ``linalg.dense_log_prob``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import settings
from .kernels import DiagNoise, LazyKernelMatrix


class DenseCovariance:
    """Dense (M x M) covariance produced lazily by a builder, with a cheap diagonal; used for predictive MVNs."""

    def __init__(self, diag, builder=None, added_diag: Optional[torch.Tensor] = None, n: Optional[int] = None):
        # ``diag``: the diagonal, or a callable producing it on first use (a prediction asked for its mean only never pays the
        # O(M N^2) product behind the variance); ``n``: the size, needed while the diagonal does not exist yet
        self._diag, self._builder, self._added = diag, builder, added_diag
        self._n = n if n is not None else diag.shape[0]
        self._dense = None

    @property
    def shape(self):
        return torch.Size([self._n, self._n])

    def _diagonal(self) -> torch.Tensor:
        if callable(self._diag):
            self._diag = self._diag()
        return self._diag

    def diag(self):
        d = self._diagonal()
        return d if self._added is None else d + self._added

    def add_diag_vector(self, v: torch.Tensor) -> "DenseCovariance":
        added = v if self._added is None else self._added + v
        out = DenseCovariance(self._diagonal(), self._builder, added, n=self._n)
        out._dense = self._dense
        return out

    def evaluate(self) -> torch.Tensor:
        if self._dense is None:
            if self._builder is None:
                raise RuntimeError("this covariance only carries its diagonal")
            self._dense = self._builder()
        if self._added is None:
            return self._dense
        out = self._dense.clone()
        out.diagonal().add_(self._added)
        return out

    to_dense = evaluate

    def __add__(self, other):
        if isinstance(other, DiagNoise):
            return self.add_diag_vector(other.diag())
        return NotImplemented


class MultivariateNormal:
    def __init__(self, mean: torch.Tensor, covariance_matrix):
        self.loc = mean
        self._covar = covariance_matrix

    # -- accessors -------------------------------------------------------------------------------
    @property
    def mean(self):
        return self.loc

    @property
    def lazy_covariance_matrix(self):
        return self._covar

    @property
    def covariance_matrix(self):
        return self._covar if torch.is_tensor(self._covar) else self._covar.evaluate()

    @property
    def variance(self):
        d = self._covar.diagonal() if torch.is_tensor(self._covar) else self._covar.diag()
        return d.clamp_min(settings.min_variance.value())

    @property
    def stddev(self):
        return self.variance.sqrt()

    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    def confidence_region(self):
        s2 = self.stddev * 2
        return self.loc - s2, self.loc + s2

    # -- the hot operator ------------------------------------------------------------------------
    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        cov = self._covar
        if isinstance(cov, LazyKernelMatrix):
            if not cov.is_square:
                raise RuntimeError("log_prob needs a square covariance")
            if cov.tau is None:
                raise RuntimeError("log_prob of a noise-free kernel matrix: apply the likelihood first")
            from ..linalg import exact_mll

            return exact_mll(cov.U1, cov.spec, cov.tau, self.loc, value, cov.grp, cov.n_grad_dims)
        from ..linalg import dense_log_prob

        dense = cov if torch.is_tensor(cov) else cov.evaluate()
        return dense_log_prob(dense, (value - self.loc).to(torch.float64))

    def __repr__(self):
        return f"MultivariateNormal(loc: {tuple(self.loc.shape)})"
