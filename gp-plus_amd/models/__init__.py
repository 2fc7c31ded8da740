from .gpregression import GPR  # noqa: F401
from .gp_plus import GP_Plus  # noqa: F401
