"""Own minimal implementation of the gpytorch protocol GP+ is written against (gpytorch is not a dependency)."""
from .. import settings  # noqa: F401
from . import metrics  # noqa: F401
from ..errors import NanError, NotPSDError  # noqa: F401
from .module import GreaterThan, Interval, LogNormalPrior, Module, NormalPrior, Positive, Prior  # noqa: F401
from .kernels import (Kernel, LazyKernelMatrix, MaternKernel, ProductKernel, RBFKernel, ScaleKernel)  # noqa: F401
from .distributions import DenseCovariance, MultivariateNormal  # noqa: F401
from .means import ConstantMean, Mean, ZeroMean  # noqa: F401
from .likelihoods import GaussianLikelihood, HomoskedasticNoise, _GaussianLikelihoodBase  # noqa: F401
from .models import ExactGP, GP  # noqa: F401
from .mlls import ExactMarginalLogLikelihood  # noqa: F401
