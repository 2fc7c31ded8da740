"""gpytorch.utils.errors stand-ins (caught by optim/mll_scipy.py:18,233 in the reference)."""


class NotPSDError(RuntimeError):
    pass


class NanError(RuntimeError):
    pass
