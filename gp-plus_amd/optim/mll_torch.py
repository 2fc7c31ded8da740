"""Adam fit driver (reference: optim/mll_torch.py:56-141): ``(num_restarts+1) x num_iter`` evaluations of
``loss = -mll(model(*train_inputs), y); loss.backward(); optimizer.step()``, best-state tracking, restarts from prior
samples.  Every evaluation runs on the HIP back end through ``ExactMarginalLogLikelihood`` -> ``log_prob``."""
import math
from copy import deepcopy
from typing import List, Optional

import torch
from tqdm import tqdm

from ..backend import INFO_PANEL_TIMEOUT
from ..gpcore.mlls import ExactMarginalLogLikelihood


def _plateaued(history: List[float], j: int, window: int) -> bool:
    """The reference's early stop (optim/mll_torch.py:126-128): at every ``window``-th iteration after the first
    ``window`` ones, stop when the mean of the last ``window`` losses is not above the current one (float32 mean, as
    there).  First reachable at j = 2 * window."""
    if j <= window or j % window != 0:
        return False
    recent = torch.Tensor(history)[j - window:j]
    return bool((torch.mean(recent) - history[j]) <= 0)


def _graphed_step(model, mll, params, evaluations: int):
    """The replayed form of one evaluation (gp-plus_amd/graphed.py::GraphedLossAndGrad), or None where it does not apply: a GPU
    model below the look-ahead factorisation's size (its internal streams do not belong in a graph, and one evaluation hides the
    host there anyway), enough evaluations to repay three warm-up evaluations and the capture, ``settings.graphed_objective`` on."""
    from .. import settings
    from ..linalg import LOOKAHEAD_MIN_N

    plist = [p for p in params if p.requires_grad]
    if not plist or not settings.graphed_objective.value() or evaluations <= 8:
        return None
    dev, n = plist[0].device, int(model.train_targets.shape[0])
    if dev.type != "cuda" or n >= LOOKAHEAD_MIN_N or settings.sharded_evaluation.value() is not None:
        return None
    from ..graphed import GraphedLossAndGrad

    def closure():
        return -mll(model(*model.train_inputs), model.train_targets)

    from .._lib import GppError
    from ..errors import NanError, NotPSDError

    try:
        return GraphedLossAndGrad(closure, plist, n, dev)
    except (NotPSDError, NanError):
        # the eager warm-up evaluations failed at THIS start point (indefinite covariance, NaN inputs): not a capture problem — the
        # eager loop meets the same point and raises the reference's own error with its own message
        return None
    except GppError:
        raise  # a genuine library error is never "could not be captured"
    except RuntimeError as exc:  # a capture this stack refuses: the eager loop is the same computation — but say so, once
        import warnings

        warnings.warn(f"fit_model_torch: the evaluation could not be captured as a HIP graph ({exc}); running it eagerly",
                      RuntimeWarning)
        return None


def _adam_run(model, mll, params, lr: float, num_iter: int, break_steps: int, verbose: bool, graphed=None) -> List[float]:
    """One start point: up to ``num_iter`` Adam steps on ``-mll`` (optim/mll_torch.py:99-128).  Returns the losses seen
    BEFORE each step; the last entry is what the restart is judged by.  ``graphed``: the evaluation as a replayed HIP graph
    (Adam itself stays outside it, so its arithmetic is the eager one); an iteration the graph hands back — factorisation status
    not zero, a non-finite number — is evaluated eagerly with the jitter policy and the exceptions of the reference."""
    optimizer = torch.optim.Adam(params, lr=lr)
    history: List[float] = []
    bar = tqdm(range(num_iter), desc='Epoch', position=0, leave=True, disable=not verbose)
    for j in bar:
        value = graphed.step() if graphed is not None and not graphed.dead else None
        if value is None:
            if graphed is not None and not graphed.dead and graphed.last_status >= INFO_PANEL_TIMEOUT:
                # a cooperative launch inside the captured evaluation timed out (another tenant on the GPU): the eager evaluation
                # below switches that path off for the context, but the CAPTURE still contains it — every replay would pay the
                # time-out again.  The rest of this start point runs eagerly; the driver re-captures before the next one.
                graphed.dead = True
            optimizer.zero_grad()
            loss = -mll(model(*model.train_inputs), model.train_targets)
            loss.backward()
            value = None
        optimizer.step()
        history.append(loss.item() if value is None else value)
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
    # the evaluation as one replayed HIP graph at the examples' sizes (the reference's notebooks and BO loop call this function
    # directly): captured once per fit, the restarts only change the parameters' values
    graphed = _graphed_step(model, mll, list(model.parameters()), (num_restarts + 1) * num_iter) if model_param_groups is None else None
    fit_model_torch.last_graph = None
    for restart in range(num_restarts + 1):
        params = model.parameters() if model_param_groups is None else model_param_groups
        history = _adam_run(model, mll, params, lr_default, num_iter, break_steps, verbose, graphed)
        histories.append(history)
        if history and history[-1] < best_loss:  # strict, as in the reference: ties keep the earlier start
            best_loss, best_state = history[-1], deepcopy(model.state_dict())
        if restart < num_restarts:
            model.reset_parameters()  # next start point: a draw from the priors (models/gpregression.py:168-174)
            if graphed is not None and graphed.dead:  # (see _adam_run) capture again, now without the launch that timed out
                counts = (graphed.replays, graphed.declined)
                for p in model.parameters():
                    p.grad = None
                graphed = _graphed_step(model, mll, list(model.parameters()), (num_restarts - restart) * num_iter)
                if graphed is not None:
                    graphed.replays, graphed.declined = graphed.replays + counts[0], graphed.declined + counts[1]
    if graphed is not None:  # only the counters outlive the fit: the graph and its memory pool are released here
        from types import SimpleNamespace
        fit_model_torch.last_graph = SimpleNamespace(replays=graphed.replays, declined=graphed.declined)
        for p in model.parameters():
            p.grad = None  # (they point into the graph's pool)
        graphed = None
    model.load_state_dict(best_state)
    return best_loss, histories
