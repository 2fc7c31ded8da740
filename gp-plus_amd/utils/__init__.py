from .set_seed import set_seed  # noqa: F401
from .data_type_check import data_type_check  # noqa: F401
from .transforms import softplus, inv_softplus  # noqa: F401
