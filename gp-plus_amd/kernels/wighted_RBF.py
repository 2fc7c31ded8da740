"""``wighted_RBF``: the reference's class (kernels/wighted_RBF.py:18-49) is an unfinished stub whose ARD/grad branch
returns an all-ones matrix (valid only for 1-D input).  What BASELINE config 3 and Examples/02 actually execute is
the product kernel RBF(z; l=1) x RBF(x_quant; l(omega)) = exp(-sum_d w_d dx_d^2) with w = [1/2, 1/2, 10^omega...]
(models/gp_plus.py:219-303); this class exposes exactly that weighted form as a single operator:
w_d = lengthscale_d on its active dims (same convention as Rough_RBF), optionally preceded by fixed weights.
The deviation from the stub is documented in DESIGN.md."""
import torch

from ..gpcore.kernels import Kernel


class wighted_RBF(Kernel):
    has_lengthscale = True

    def __init__(self, fixed_weights=None, **kwargs):
        super().__init__(**kwargs)
        self.register_buffer("fixed_weights", None if fixed_weights is None else torch.as_tensor(fixed_weights, dtype=torch.float64))

    def feature_weights(self, D):
        w = self._scatter(self.lengthscale, D)
        if self.fixed_weights is not None:
            fw = torch.zeros(D, dtype=torch.float64, device=w.device)
            fw[: self.fixed_weights.numel()] = self.fixed_weights.to(w.device)
            w = w + fw
        return w
