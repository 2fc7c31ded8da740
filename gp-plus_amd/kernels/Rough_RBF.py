"""Standalone ``Rough_RBF`` (reference: kernels/Rough_RBF.py:18-40), both branches of its ``forward``:

  * an input requires grad, ``ard_num_dims > 1``, ``diag`` or ``last_dim_is_batch`` (:19-32): inputs scaled by
    sqrt(lengthscale) and the file's own ``postprocess_rbf`` = exp(-dist^2) (:6-7)  =>  k = exp(-sum_d l_d (x_d-x'_d)^2),
    i.e. w_d = l_d;
  * otherwise (:33-40): gpytorch's ``RBFCovariance``  =>  k = exp(-||x-x'||^2 / (2 l^2)), i.e. w = 1 / (2 l^2).

(Inside GP_Plus the name 'Rough_RBF' is swapped for gpytorch's RBFKernel with the 2^-1/2 10^(-omega/2) transform,
models/gp_plus.py:229-230,248-253; that path uses gpcore.RBFKernel.  This class is what
``GPR(correlation_kernel='Rough_RBF')`` instantiates, models/gpregression.py:89-102.)"""
from ..gpcore.kernels import Kernel, call_needs_branch1


class Rough_RBF(Kernel):
    has_lengthscale = True

    def feature_weights(self, D):
        if call_needs_branch1() or (self.ard_num_dims is not None and self.ard_num_dims > 1):
            return self._scatter(self.lengthscale, D)
        return self._scatter(0.5 / self.lengthscale.pow(2), D)
