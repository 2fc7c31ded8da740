"""Kernel operator surface (gpytorch ``Kernel`` protocol subset) and the lazy covariance it returns.

Every stationary kernel here contributes a per-feature weight vector w so that the whole stack — ScaleKernel,
ProductKernel, RBF / Rough_RBF factors (models/gp_plus.py:219-303, models/gpregression.py:108-111) — collapses to
    K_ij = sf2 * exp(-sum_d w_d (u_id-u_jd)^2)          
and is evaluated by ONE fused HIP tile kernel (``gpp_kernel_build``) instead of gpytorch's distance GEMM + ~6 N^2
elementwise passes.  ``kernel(x)`` returns a :class:`LazyKernelMatrix` (the analogue of gpytorch's
LazyEvaluatedKernelTensor); nothing N x N exists until ``evaluate()`` or ``log_prob`` asks for it.
"""
from __future__ import annotations

import threading
from typing import List, Optional

import torch

from ..backend import KIND_MATERN32, KIND_MATERN52, KIND_RBF
from ..linalg import KernelSpec, cross_kernel, dense_kernel
from .module import Interval, Module, Positive


#: what the reference's ``forward(x1, x2, diag, **params)`` can see of its call and the weight protocol cannot: whether an
#: input requires grad / diag was asked (kernels/Rough_RBF.py:19-26 switches formula on it).  Set by ``Kernel.__call__``
#: around ``spec()`` for the calling host thread.
_call = threading.local()


def call_needs_branch1() -> bool:
    return getattr(_call, "branch1", False)


class LazyKernelMatrix:
    """sf2*k(U1,U2;w) (+ diag(tau[grp]) when square) held symbolically."""

    def __init__(self, U1: torch.Tensor, U2: Optional[torch.Tensor], spec: KernelSpec,
                 tau: Optional[torch.Tensor] = None, grp: Optional[torch.Tensor] = None, n_grad_dims: Optional[int] = None):
        self.U1, self.U2, self.spec, self.tau, self.grp = U1, U2, spec, tau, grp
        self.n_grad_dims = n_grad_dims

    @property
    def is_square(self) -> bool:
        return self.U2 is None

    @property
    def shape(self):
        return torch.Size([self.U1.shape[0], (self.U1 if self.U2 is None else self.U2).shape[0]])

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    @property
    def device(self):
        return self.U1.device

    @property
    def dtype(self):
        return torch.float64

    def add_diag(self, tau: torch.Tensor, grp: Optional[torch.Tensor] = None) -> "LazyKernelMatrix":
        """K + diag(tau[grp]) — what ``likelihood(mvn)`` does (likelihoods_noise/multifidelity.py:63-67)."""
        if not self.is_square:
            raise RuntimeError("diagonal noise needs a square covariance")
        if self.tau is not None:
            raise RuntimeError("noise was already added to this covariance")
        return LazyKernelMatrix(self.U1, None, self.spec, tau, grp, self.n_grad_dims)

    def noise_vector(self) -> Optional[torch.Tensor]:
        if self.tau is None:
            return None
        t = self.tau.reshape(-1)
        if self.grp is None:
            return t[0].expand(self.U1.shape[0])
        return t[self.grp.long()]

    def diag(self) -> torch.Tensor:
        n = self.U1.shape[0]
        if self.spec.kind != KIND_RBF and not self.is_square:
            raise NotImplementedError
        d = self.spec.sf2.reshape(()).to(torch.float64).expand(n)
        nv = self.noise_vector()
        return d if nv is None else d + nv.to(d)

    diagonal = diag

    def evaluate(self) -> torch.Tensor:
        """Dense matrix (no autograd): ``covar_x.evaluate()`` of models/gp_plus.py:474."""
        if self.is_square:
            return dense_kernel(self.U1, self.spec, self.tau, self.grp)
        return cross_kernel(self.U1, self.U2, self.spec)

    to_dense = evaluate

    def __add__(self, other):
        if isinstance(other, DiagNoise):
            return self.add_diag(other.tau, other.grp)
        return NotImplemented


class DiagNoise:
    """diag(tau[grp]) produced by a likelihood's noise model."""

    def __init__(self, tau: torch.Tensor, grp: Optional[torch.Tensor], n: int):
        self.tau, self.grp, self.n = tau, grp, n

    def diag(self):
        t = self.tau.reshape(-1)
        return t[0].expand(self.n) if self.grp is None else t[self.grp.long()]


class Kernel(Module):
    """gpytorch.kernels.Kernel subset: ``ard_num_dims``, ``active_dims``, ``lengthscale`` (+constraint, raw parameter
    of shape (1, ard_num_dims)), ``k1 * k2`` -> ProductKernel, ``kernel(x1[, x2])`` -> lazy matrix."""

    has_lengthscale = False

    def __init__(self, ard_num_dims: Optional[int] = None, active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint: Optional[Interval] = None, **kwargs):
        super().__init__()
        if active_dims is not None and not torch.is_tensor(active_dims):
            active_dims = torch.tensor(active_dims, dtype=torch.long)
        self.register_buffer("active_dims", active_dims)
        self.ard_num_dims = ard_num_dims
        if self.has_lengthscale:
            n = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(1, n)))
            self.register_constraint("raw_lengthscale", lengthscale_constraint if lengthscale_constraint is not None else Positive())
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda m: m.lengthscale,
                                    lambda m, v: m._set_lengthscale(v))

    @property
    def lengthscale(self):
        if not self.has_lengthscale:
            return None
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_lengthscale)
        self.initialize(raw_lengthscale=self.raw_lengthscale_constraint.inverse_transform(value.to(self.raw_lengthscale)))

    # -- weighted-distance protocol ----------------------------------------------------------------
    def _dims(self, D: int) -> torch.Tensor:
        return torch.arange(D, device=self.raw_lengthscale.device if self.has_lengthscale else None) \
            if self.active_dims is None else self.active_dims

    def feature_weights(self, D: int) -> torch.Tensor:
        """This kernel's w (length D, zero outside active_dims)."""
        raise NotImplementedError

    def kind_and_split(self, D: int):
        return KIND_RBF, 0

    def spec(self, D: int, device, sf2: Optional[torch.Tensor] = None) -> KernelSpec:
        w = self.feature_weights(D).to(device=device, dtype=torch.float64)
        kind, d_split = self.kind_and_split(D)
        if sf2 is None:
            sf2 = torch.ones((), dtype=torch.float64, device=device)
        return KernelSpec(w, sf2.to(device=device, dtype=torch.float64).reshape(()), kind, d_split)

    def _scatter(self, values: torch.Tensor, D: int) -> torch.Tensor:
        dims = self._dims(D).to(values.device)
        vals = values.reshape(-1).to(torch.float64)
        if vals.numel() == 1 and dims.numel() > 1:
            vals = vals.expand(dims.numel())
        w = torch.zeros(D, dtype=torch.float64, device=values.device)
        return w.index_add(0, dims, vals)

    def forward(self, x1, x2=None, diag=False, **params):
        return self.__call__(x1, x2, diag=diag, **params)

    def __call__(self, x1, x2=None, diag: bool = False, **params):
        if x1.dim() == 1:
            x1 = x1.unsqueeze(-1)
        if x2 is not None and x2.dim() == 1:
            x2 = x2.unsqueeze(-1)
        prev = getattr(_call, "branch1", False)
        _call.branch1 = bool(x1.requires_grad or (x2 is not None and x2.requires_grad) or diag
                             or params.get("last_dim_is_batch", False))
        try:
            lazy = LazyKernelMatrix(x1, x2, self.spec(x1.shape[-1], x1.device))
        finally:
            _call.branch1 = prev
        return lazy.diag() if diag else lazy

    def __mul__(self, other):
        kernels: List[Kernel] = []
        kernels += list(self.kernels) if isinstance(self, ProductKernel) else [self]
        kernels += list(other.kernels) if isinstance(other, ProductKernel) else [other]
        return ProductKernel(*kernels)


class RBFKernel(Kernel):
    """gpytorch RBFKernel: exp(-0.5 ||(x-x')/l||^2)  =>  w_d = 1/(2 l_d^2)  (models/gp_plus.py:223-253)."""
    has_lengthscale = True

    def feature_weights(self, D):
        return self._scatter(0.5 / self.lengthscale.pow(2), D)


class MaternKernel(Kernel):
    """gpytorch MaternKernel nu in {1.5, 2.5} (kernels/matern.py:4-8): distance r = ||(x-x')/l||; the tile kernel
    receives 2 w_d = 1/l_d^2 through the same w convention (r^2 = 2 sum w_d dx^2)."""
    has_lengthscale = True

    def __init__(self, nu: float = 2.5, **kwargs):
        if nu not in (1.5, 2.5):
            raise RuntimeError("nu expected to be 1.5 or 2.5")
        super().__init__(**kwargs)
        self.nu = nu

    def feature_weights(self, D):
        return self._scatter(0.5 / self.lengthscale.pow(2), D)

    def kind_and_split(self, D):
        return (KIND_MATERN32 if self.nu == 1.5 else KIND_MATERN52), 0


class ProductKernel(Kernel):
    """Product of stationary factors = sum of their weight vectors (flattened like gpytorch: ``kernels.{i}``)."""

    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    def feature_weights(self, D):
        w = None
        for k in self.kernels:
            wk = k.feature_weights(D)
            w = wk if w is None else w + wk.to(w.device)
        return w

    def kind_and_split(self, D):
        kinds = [k.kind_and_split(D) for k in self.kernels]
        non_rbf = [(i, ks) for i, ks in enumerate(kinds) if ks[0] != KIND_RBF]
        if not non_rbf:
            return KIND_RBF, 0
        if len(non_rbf) > 1:
            raise NotImplementedError("at most one Matern factor per product")
        i, (kind, _) = non_rbf[0]
        mat_dims = self.kernels[i]._dims(D)
        d_split = int(mat_dims.min())
        rbf_dims = [int(d) for j, k in enumerate(self.kernels) if j != i for d in k._dims(D)]
        if rbf_dims and max(rbf_dims) >= d_split:
            raise NotImplementedError("RBF factors must act on the leading dims, the Matern factor on the trailing dims")
        return kind, d_split


class ScaleKernel(Kernel):
    """gpytorch ScaleKernel: outputscale * base_kernel (models/gpregression.py:108-111)."""

    def __init__(self, base_kernel: Kernel, outputscale_prior=None, outputscale_constraint: Optional[Interval] = None, **kwargs):
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(())))
        self.register_constraint("raw_outputscale", outputscale_constraint if outputscale_constraint is not None else Positive())
        if outputscale_prior is not None:
            self.register_prior("outputscale_prior", outputscale_prior, lambda m: m.outputscale, lambda m, v: m._set_outputscale(v))

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        self._set_outputscale(value)

    def _set_outputscale(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value)
        value = value.to(self.raw_outputscale)
        self.initialize(raw_outputscale=self.raw_outputscale_constraint.inverse_transform(value).reshape(()))

    def feature_weights(self, D):
        return self.base_kernel.feature_weights(D)

    def kind_and_split(self, D):
        return self.base_kernel.kind_and_split(D)

    def spec(self, D, device, sf2=None):
        os_ = self.outputscale if sf2 is None else sf2 * self.outputscale
        return self.base_kernel.spec(D, device, os_)
