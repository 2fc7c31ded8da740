from .mll_torch import fit_model_torch  # noqa: F401
