"""Minimal stand-ins for the gpytorch protocol GP+ is written against: ``Module`` with priors,
constraints and ``initialize``; ``Interval``/``GreaterThan``/``Positive`` constraints; ``Prior`` base with the two
torch.distributions priors the path uses.  gpytorch itself is not a dependency.

Call sites in the reference that rely on this protocol: ``register_prior`` models/gpregression.py:84,98-100,113-115,
models/gp_plus.py:274-295,1235-1247; ``named_priors`` models/gpregression.py:171; ``initialize`` models/gp_plus.py:225;
constraints models/gpregression.py:59,93-95,108-111, models/gp_plus.py:243-272.
"""
from __future__ import annotations

import math
from typing import Callable, Iterator, Optional, Tuple, Union

import torch
from torch import nn


# ------------------------------------------------------------------------------------------------
# constraints: constrained = transform(raw) + lower_bound   (gpytorch.constraints.Interval semantics for
# a one-sided bound with a user transform, which is the only form the reference uses)
# ------------------------------------------------------------------------------------------------
class Interval(nn.Module):
    def __init__(self, lower_bound=0.0, upper_bound=math.inf, transform: Callable = torch.nn.functional.softplus,
                 inv_transform: Optional[Callable] = None, initial_value=None):
        super().__init__()
        self.register_buffer("lower_bound", torch.as_tensor(float(lower_bound)))
        self.register_buffer("upper_bound", torch.as_tensor(float(upper_bound)))
        self._transform = transform
        self._inv_transform = inv_transform
        self._initial_value = initial_value

    @property
    def enforced(self) -> bool:
        return self._transform is not None

    @property
    def initial_value(self):
        return self._initial_value

    def transform(self, tensor: torch.Tensor) -> torch.Tensor:
        if not self.enforced:
            return tensor
        return self._transform(tensor) + self.lower_bound.to(tensor)

    def inverse_transform(self, transformed: torch.Tensor) -> torch.Tensor:
        if not self.enforced:
            return transformed
        if self._inv_transform is None:
            raise RuntimeError("constraint has no inverse transform")
        return self._inv_transform(transformed - self.lower_bound.to(transformed))

    def check(self, tensor) -> bool:
        return bool(torch.all(tensor <= self.upper_bound) and torch.all(tensor >= self.lower_bound))


class GreaterThan(Interval):
    def __init__(self, lower_bound, transform=torch.nn.functional.softplus, inv_transform=None, initial_value=None):
        if inv_transform is None and transform is torch.nn.functional.softplus:
            inv_transform = lambda x: x + torch.log(-torch.expm1(-x))  # noqa: E731
        super().__init__(lower_bound, math.inf, transform, inv_transform, initial_value)


class Positive(GreaterThan):
    def __init__(self, transform=torch.nn.functional.softplus, inv_transform=None, initial_value=None):
        super().__init__(0.0, transform, inv_transform, initial_value)


# ------------------------------------------------------------------------------------------------
# priors
# ------------------------------------------------------------------------------------------------
class Prior(nn.Module):
    """log_prob / sample / expand, the three methods the fit drivers use (optim/mll_torch.py:116 through the MLL,
    models/gpregression.py:168-174)."""

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError

    def rsample(self, sample_shape=torch.Size([])) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError

    def sample(self, sample_shape=torch.Size([])) -> torch.Tensor:
        with torch.no_grad():
            return self.rsample(sample_shape)

    def expand(self, batch_shape):  # pragma: no cover - interface
        raise NotImplementedError


_HALF_LOG_2PI = 0.5 * __import__("math").log(2 * __import__("math").pi)


class NormalPrior(Prior):
    """gpytorch.priors.NormalPrior (models/gp_plus.py:279-295,495,1247)."""

    def __init__(self, loc, scale):
        super().__init__()
        self.register_buffer("loc", torch.as_tensor(loc, dtype=torch.get_default_dtype()))
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.get_default_dtype()))

    def _dist(self, like=None):
        loc, scale = self.loc, self.scale
        if like is not None:
            loc, scale = loc.to(like), scale.to(like)
        return torch.distributions.Normal(loc, scale)

    def log_prob(self, x):
        # torch.distributions.Normal.log_prob written out: building the distribution object validates its arguments
        # with a device sync per tensor, several times per evaluation of a small model
        loc, scale = self.loc.to(x), self.scale.to(x)
        return -((x - loc) ** 2) / (2 * scale ** 2) - scale.log() - _HALF_LOG_2PI

    def rsample(self, sample_shape=torch.Size([])):
        return self._dist().rsample(sample_shape)

    def expand(self, batch_shape):
        return NormalPrior(self.loc.expand(torch.Size(batch_shape)), self.scale.expand(torch.Size(batch_shape)))


class LogNormalPrior(Prior):
    """gpytorch.priors.LogNormalPrior (models/gpregression.py:113-115)."""

    def __init__(self, loc, scale):
        super().__init__()
        self.register_buffer("loc", torch.as_tensor(loc, dtype=torch.get_default_dtype()))
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.get_default_dtype()))

    def _dist(self, like=None):
        loc, scale = self.loc, self.scale
        if like is not None:
            loc, scale = loc.to(like), scale.to(like)
        return torch.distributions.LogNormal(loc, scale)

    def log_prob(self, x):
        # TransformedDistribution(Normal, ExpTransform).log_prob written out (see NormalPrior.log_prob)
        loc, scale = self.loc.to(x), self.scale.to(x)
        lx = x.log()
        return -((lx - loc) ** 2) / (2 * scale ** 2) - scale.log() - _HALF_LOG_2PI - lx

    def rsample(self, sample_shape=torch.Size([])):
        return self._dist().rsample(sample_shape)

    def expand(self, batch_shape):
        return LogNormalPrior(self.loc.expand(torch.Size(batch_shape)), self.scale.expand(torch.Size(batch_shape)))


# ------------------------------------------------------------------------------------------------
# Module
# ------------------------------------------------------------------------------------------------
def _named_priors(module: nn.Module, memo: set, prefix: str):
    """Depth-first over ALL nn.Module children (plain containers such as ModuleList included), like gpytorch."""
    for name, (prior, closure, inv_closure) in getattr(module, "_priors", {}).items():
        if prior is not None and prior not in memo:
            memo.add(prior)
            yield (prefix + ("." if prefix else "") + name, module, prior, closure, inv_closure)
    for mname, child in module.named_children():
        if isinstance(child, Prior):
            continue
        yield from _named_priors(child, memo, prefix + ("." if prefix else "") + mname)


class Module(nn.Module):
    """gpytorch.Module subset: priors, constraints, initialize."""

    def __init__(self):
        super().__init__()
        self._priors = {}
        self._constraints_names = {}

    def __call__(self, *inputs, **kwargs):
        return self.forward(*inputs, **kwargs)

    # -- constraints -----------------------------------------------------------------------------
    def register_constraint(self, param_name: str, constraint: Interval) -> None:
        if param_name not in self._parameters:
            raise RuntimeError(f"no parameter {param_name} to constrain")
        self.add_module(param_name + "_constraint", constraint)

    def constraint_for_parameter_name(self, param_name: str) -> Optional[Interval]:
        return self._modules.get(param_name + "_constraint")

    # -- priors ----------------------------------------------------------------------------------
    def register_prior(self, name: str, prior: Prior, param_or_closure: Union[str, Callable],
                       setting_closure: Optional[Callable] = None) -> None:
        if isinstance(param_or_closure, str):
            pname = param_or_closure
            if pname not in self._parameters and not hasattr(self, pname):
                raise AttributeError(f"Unknown parameter {pname} for {type(self).__name__}")

            def closure(module, _n=pname):
                return getattr(module, _n)

            if setting_closure is not None:
                raise RuntimeError("a setting closure needs a closure, not a parameter name")

            def setting_closure(module, val, _n=pname):  # noqa: E306
                return module.initialize(**{_n: val})
        else:
            closure = param_or_closure
        self.add_module(name, prior)
        self._priors[name] = (prior, closure, setting_closure)

    def named_priors(self, memo=None, prefix: str = "") -> Iterator[Tuple[str, "Module", Prior, Callable, Callable]]:
        if memo is None:
            memo = set()
        yield from _named_priors(self, memo, prefix)

    # -- initialize ------------------------------------------------------------------------------
    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            if isinstance(val, (int, float)):
                val = float(val)
            if "." in name:
                head, rest = name.split(".", 1)
                getattr(self, head).initialize(**{rest: val})
            elif name in self._parameters:
                param = self._parameters[name]
                with torch.no_grad():
                    if torch.is_tensor(val):
                        param.copy_(val.to(param).expand_as(param) if val.numel() in (1, param.numel()) and val.shape != param.shape
                                    else val.to(param).reshape(param.shape))
                    else:
                        param.fill_(val)
            elif hasattr(type(self), name) or hasattr(self, name):
                setattr(self, name, val)
            else:
                raise AttributeError(f"Unknown parameter {name} for {type(self).__name__}")
        return self
