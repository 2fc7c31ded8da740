"""utils/set_seed.py:6-15 of the reference."""
import random

import numpy as np
import torch


def set_seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
