"""gpytorch.metrics subset used by GP_Plus.evaluation (models/gp_plus.py:900-908)."""
import torch


def negative_log_predictive_density(pred_dist, test_y):
    return -pred_dist.log_prob(test_y) / test_y.shape[-1]


def mean_squared_error(pred_dist, test_y, squared: bool = True):
    res = torch.square(pred_dist.mean - test_y).mean(dim=-1)
    return res if squared else res.sqrt()


def mean_absolute_error(pred_dist, test_y):
    return torch.abs(pred_dist.mean - test_y).mean(dim=-1)
