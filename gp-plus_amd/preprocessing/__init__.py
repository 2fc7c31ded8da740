from .normalizeX import standard  # noqa: F401
from .numericlevels import setlevels  # noqa: F401
from .split import train_test_split_normalizeX  # noqa: F401
