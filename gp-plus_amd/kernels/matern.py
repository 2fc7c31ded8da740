"""kernels/matern.py:4-8 of the reference: MaternKernel with nu fixed to 1.5 / 2.5."""
from ..gpcore.kernels import MaternKernel


class Matern32Kernel(MaternKernel):
    def __init__(self, **kwargs):
        super().__init__(nu=1.5, **kwargs)


class Matern52Kernel(MaternKernel):
    def __init__(self, **kwargs):
        super().__init__(nu=2.5, **kwargs)
