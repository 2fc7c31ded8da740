from .horseshoe import LogHalfHorseshoePrior  # noqa: F401
from .mollified_uniform import MollifiedUniformPrior  # noqa: F401
