"""Operator surface of the reference's ``kernels`` package (kernels/__init__.py:1-6): same names, evaluated by the
fused HIP tile kernel through the weighted-distance protocol of ``gpcore.kernels``."""
from ..gpcore.kernels import Kernel, MaternKernel, ProductKernel, RBFKernel, ScaleKernel  # noqa: F401
from .matern import Matern32Kernel, Matern52Kernel  # noqa: F401
from .Rough_RBF import Rough_RBF  # noqa: F401
from .wighted_RBF import wighted_RBF  # noqa: F401
