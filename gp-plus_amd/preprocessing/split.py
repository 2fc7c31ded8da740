"""``train_test_split_normalizeX`` (reference: preprocessing/split.py:7-48): sklearn split, then ``standard``."""
import torch
from sklearn.model_selection import train_test_split

from .normalizeX import standard
from .numericlevels import setlevels


def train_test_split_normalizeX(X, y, test_size=None, shuffle=True, stratify=None, qual_dict={}, random_state=1,
                                return_mean_std=False, set_levels=False):
    qual_index = list(qual_dict.keys())
    if set_levels:
        X = setlevels(X, qual_index=qual_index)
    Xtrain, Xtest, ytrain, ytest = train_test_split(X, y, test_size=test_size, shuffle=shuffle,
                                                    random_state=random_state, stratify=stratify)
    Xtrain, Xtest, mean_train, std_train = standard(Xtrain=Xtrain, qual_index=qual_dict, Xtest=Xtest)
    out = [v if isinstance(v, torch.Tensor) else torch.tensor(v) for v in (Xtrain, Xtest, ytrain, ytest)]
    if return_mean_std:
        return (*out, mean_train, std_train)
    return tuple(out)
