"""Adam fit driver (reference: optim/mll_torch.py:56-141): ``(num_restarts+1) x num_iter`` evaluations of
``loss = -mll(model(*train_inputs), y); loss.backward(); optimizer.step()``, best-state tracking, restarts from prior
samples.  Every evaluation runs on the HIP back end through ``ExactMarginalLogLikelihood`` -> ``log_prob``."""
import math
from copy import deepcopy
from typing import List, Optional

import torch
from tqdm import tqdm

from ..gpcore.mlls import ExactMarginalLogLikelihood


def _plateaued(history: List[float], j: int, window: int) -> bool:
    """The reference's early stop (optim/mll_torch.py:126-128): at every ``window``-th iteration after the first
    ``window`` ones, stop when the mean of the last ``window`` losses is not above the current one (float32 mean, as
    there).  First reachable at j = 2 * window."""
    if j <= window or j % window != 0:
        return False
    recent = torch.Tensor(history)[j - window:j]
    return bool((torch.mean(recent) - history[j]) <= 0)


def _adam_run(model, mll, params, lr: float, num_iter: int, break_steps: int, verbose: bool) -> List[float]:
    """One start point: up to ``num_iter`` Adam steps on ``-mll`` (optim/mll_torch.py:99-128).  Returns the losses seen
    BEFORE each step; the last entry is what the restart is judged by."""
    optimizer = torch.optim.Adam(params, lr=lr)
    history: List[float] = []
    bar = tqdm(range(num_iter), desc='Epoch', position=0, leave=True, disable=not verbose)
    for j in bar:
        optimizer.zero_grad()
        loss = -mll(model(*model.train_inputs), model.train_targets)
        loss.backward()
        optimizer.step()
        history.append(loss.item())
        if verbose:
            bar.set_description(f'Epoch {j} - loss {history[-1]:.4f}')
        if _plateaued(history, j, break_steps):
            break
    return history


def fit_model_torch(model, model_param_groups: Optional[List] = None, lr_default: float = 0.01, num_iter: int = 100,
                    num_restarts: int = 0, break_steps: int = 50, verbose: bool = True) -> float:
    """Optimize the log-posterior of a GP+ model with ``torch.optim.Adam`` (optim/mll_torch.py:56-141).

    :returns: ``(f_inc, loss_hist_total)`` — best (negative, per-datum) log-posterior found and the loss histories.
    """
    model.train()
    mll = ExactMarginalLogLikelihood(model.likelihood, model)
    best_loss, best_state = math.inf, model.state_dict()
    histories = []
    for restart in range(num_restarts + 1):
        params = model.parameters() if model_param_groups is None else model_param_groups
        history = _adam_run(model, mll, params, lr_default, num_iter, break_steps, verbose)
        histories.append(history)
        if history and history[-1] < best_loss:  # strict, as in the reference: ties keep the earlier start
            best_loss, best_state = history[-1], deepcopy(model.state_dict())
        if restart < num_restarts:
            model.reset_parameters()  # next start point: a draw from the priors (models/gpregression.py:168-174)
    model.load_state_dict(best_state)
    return best_loss, histories
