from .mll_torch import fit_model_torch  # noqa: F401
from .mll_scipy import fit_model_scipy, MLLObjective, marginal_log_likelihood  # noqa: F401
from .mll_parallel import fit_restarts_parallel, fit_scipy_parallel, split_restarts  # noqa: F401
from .mll_noise_continuation import fit_model_continuation, loocv_rrmse  # noqa: F401
from .mll_batched import BatchedObjective, fit_model_torch_batched  # noqa: F401
