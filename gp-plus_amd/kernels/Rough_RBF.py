"""Standalone ``Rough_RBF`` (reference: kernels/Rough_RBF.py:6-7,18-32): inputs scaled by sqrt(lengthscale), the
file's own ``postprocess_rbf`` = exp(-dist^2)  =>  k = exp(-sum_d l_d (x_d-x'_d)^2), i.e. w_d = l_d.
(Inside GP_Plus the name 'Rough_RBF' is swapped for gpytorch's RBFKernel with the 2^-1/2 10^(-omega/2) transform,
models/gp_plus.py:229-230,248-253; that path uses gpcore.RBFKernel.)  The reference's no-grad 1-D branch
(RBFCovariance, Rough_RBF.py:33-40) silently switches to exp(-d^2/2l^2); this implementation uses the branch-1 formula
everywhere."""
from ..gpcore.kernels import Kernel


class Rough_RBF(Kernel):
    has_lengthscale = True

    def feature_weights(self, D):
        return self._scatter(self.lengthscale, D)
