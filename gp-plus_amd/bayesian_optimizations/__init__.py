from .AFs import AF_EI, AF_HF, AF_HF_Engineering, AF_LF, AF_LF_Engineering  # noqa: F401
from .BO_GP_plus import BO  # noqa: F401
