"""Adam fit driver (reference: optim/mll_torch.py:56-141): ``(num_restarts+1) x num_iter`` evaluations of
``loss = -mll(model(*train_inputs), y); loss.backward(); optimizer.step()``, best-state tracking, restarts from prior
samples.  Every evaluation runs on the HIP back end through ``ExactMarginalLogLikelihood`` -> ``log_prob``."""
import math
from copy import deepcopy
from typing import List, Optional

import torch
from tqdm import tqdm

from ..gpcore.mlls import ExactMarginalLogLikelihood


def fit_model_torch(model, model_param_groups: Optional[List] = None, lr_default: float = 0.01, num_iter: int = 100,
                    num_restarts: int = 0, break_steps: int = 50, verbose: bool = True) -> float:
    """Optimize the log-posterior of a GP+ model with ``torch.optim.Adam``.

    :returns: ``(f_inc, loss_hist_total)`` — best (negative, per-datum) log-posterior found and the loss histories.
    """
    model.train()
    mll = ExactMarginalLogLikelihood(model.likelihood, model)
    f_inc = math.inf
    current_state_dict = model.state_dict()
    loss_hist_total = []

    for i in range(num_restarts + 1):
        optimizer = torch.optim.Adam(model.parameters() if model_param_groups is None else model_param_groups, lr=lr_default)
        loss_hist = []
        epochs_iter = tqdm(range(num_iter), desc='Epoch', position=0, leave=True, disable=not verbose)
        for j in epochs_iter:
            optimizer.zero_grad()
            output = model(*model.train_inputs)
            loss = -mll(output, model.train_targets)
            loss.backward()
            optimizer.step()

            acc_loss = loss.item()
            if verbose:
                epochs_iter.set_description(f'Epoch {j} - loss {acc_loss:.4f}')
            loss_hist.append(acc_loss)
            # reference early stop (optim/mll_torch.py:126-128): first reachable at j = 2*break_steps
            if j > break_steps and j % break_steps == 0:
                if (torch.mean(torch.Tensor(loss_hist)[j - break_steps:j]) - loss_hist[j]) <= 0:
                    break
        loss_hist_total.append(loss_hist)

        if loss.item() < f_inc:
            current_state_dict = deepcopy(model.state_dict())
            f_inc = loss.item()
        if i < num_restarts:
            model.reset_parameters()

    model.load_state_dict(current_state_dict)
    return f_inc, loss_hist_total
