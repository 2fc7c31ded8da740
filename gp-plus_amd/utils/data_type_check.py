"""utils/data_type_check.py:3-11 of the reference."""
import numpy as np
import torch


def data_type_check(data):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        print("Warning: Data type was numpy.ndarray. GP+ made it a torch tensor to be able to continue.")
        return torch.from_numpy(data)
    print(f"Warning: Data type was {type(data)}. GP+ made it a torch tensor to be able to continue.")
    return torch.tensor(data)
