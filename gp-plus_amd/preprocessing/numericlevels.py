"""``setlevels``: map the values of categorical columns to 0..L-1 in sorted order (reference:
preprocessing/numericlevels.py:5-53)."""
import numpy as np
import torch

from ..utils import data_type_check


def setlevels(X, qual_index=None, return_label=False):
    if qual_index == []:
        return X
    X = data_type_check(X)
    if not isinstance(X, torch.Tensor):
        raise TypeError("X must be a PyTorch tensor or a NumPy array.")
    temp = X.detach().cpu().clone().numpy()
    labels = []
    if temp.ndim > 1:
        if qual_index is None:
            qual_index = list(range(temp.shape[-1]))
        for j in qual_index:
            levels, inverse = np.unique(temp[..., j], return_inverse=True)
            labels.append(levels.tolist())
            temp[..., j] = inverse.reshape(temp[..., j].shape)
    else:
        levels, inverse = np.unique(temp, return_inverse=True)
        labels.append(levels.tolist())
        temp = inverse
    out = torch.from_numpy(np.asarray(temp))
    return (out, labels) if return_label else out
