"""``standard``: z-score the quantitative columns with the POPULATION std, NaN-aware (reference:
preprocessing/normalizeX.py:8-72)."""
import torch


def compute_mean_std(tensor):
    means = torch.nanmean(tensor, dim=0)
    mask = ~torch.isnan(tensor)
    diffs = tensor - means
    diffs[~mask] = 0
    sum_sq = torch.sum(diffs ** 2, dim=0)
    count = mask.sum(dim=0)
    zero = count == 0
    count[zero] = 1
    stds = torch.sqrt(sum_sq / count)
    stds[zero] = 0
    return means, stds


def standard(Xtrain, qual_index, Xtest=None):
    quant_index = [i for i in range(Xtrain.shape[1]) if i not in qual_index.keys()]
    if not isinstance(Xtrain, torch.Tensor):
        Xtrain = torch.tensor(Xtrain)
    if Xtest is not None and not isinstance(Xtest, torch.Tensor):
        Xtest = torch.tensor(Xtest)
    if len(quant_index) == 0:
        return Xtrain
    temp = Xtrain[..., quant_index]
    mean_xtrain, std_xtrain = compute_mean_std(temp)
    if torch.isnan(temp).any():
        print("Warning: There are NaN values in the data. Mean and standard deviation were calculated excluding these values.")
    Xtrain[..., quant_index] = (temp - mean_xtrain) / std_xtrain
    if Xtest is None:
        return Xtrain, mean_xtrain, std_xtrain
    Xtest[..., quant_index] = (Xtest[..., quant_index] - mean_xtrain) / std_xtrain
    return Xtrain, Xtest, mean_xtrain, std_xtrain
