"""Restart-parallel Adam fit on ONE GPU: all ``num_restarts + 1`` runs of ``optim/mll_torch.py:99-141`` advance together,
every iteration being one batched evaluation (``batched.BatchedMLLFunction``: B problems per kernel launch).

The reference runs its restarts one after the other; for the data sizes of its examples (N = 100 ... 500) a single
evaluation leaves an MI355X almost empty, so evaluating the 5 ... 65 parameter sets together costs about as much as
evaluating one.  Semantics kept from ``fit_model_torch``: run 0 starts from the model's current parameters, run i >= 1
from the i-th ``model.reset_parameters()`` sample (same RNG order); Adam(lr) per run (Adam is element-wise, so one
optimizer over the stacked parameters IS B independent optimizers); the early stop of :126-128 per run (a stopped run
is frozen); the winner is the run with the smallest LAST loss and its parameters are loaded into the model.

Everything before and after the O(N^3) part — constraints and transforms, the latent map of categorical inputs, mean
functions, priors — is the model's own code, vectorised over the runs with ``torch.func.functional_call`` + ``torch.vmap``
(no second implementation of the model).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from copy import deepcopy
from typing import Dict, List, Tuple

import torch
from torch.func import functional_call

from ..batched import batched_mll
from ..gpcore.kernels import LazyKernelMatrix
from ..gpcore.mlls import ExactMarginalLogLikelihood

__all__ = ["BatchedObjective", "fit_model_torch_batched", "BATCHED_MAX_N"]

#: largest N the batched kernels take (gpp_api.hip: blocks up to GPP_BLK_MAX = 6144 rows are factored in leaf steps)
BATCHED_MAX_N = 6144


class BatchedObjective:
    """``loss()`` -> (B,) tensor of ``-(log p(y) + log priors) / N`` for the B stacked parameter sets ``self.theta``."""

    def __init__(self, model, B: int):
        self.model, self.B = model, B
        model.train()
        self.mll = ExactMarginalLogLikelihood(model.likelihood, model)
        self.names = [n for n, p in model.named_parameters() if p.requires_grad]
        self.fixed = {n: p.detach() for n, p in model.named_parameters() if not p.requires_grad}
        self.buffers = {n: b for n, b in model.named_buffers()}
        self.theta: "OrderedDict[str, torch.nn.Parameter]" = OrderedDict(
            (n, torch.nn.Parameter(p.detach().unsqueeze(0).repeat(B, *([1] * p.dim())).clone()))
            for n, p in model.named_parameters() if p.requires_grad)
        self._static = None

    # -- parameter plumbing ---------------------------------------------------------------------------------
    def set_row(self, b: int, state: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for n in self.names:
                self.theta[n][b].copy_(state[n].to(self.theta[n]))

    def row(self, b: int) -> Dict[str, torch.Tensor]:
        return {n: self.theta[n][b].detach().clone() for n in self.names}

    def sample_restarts(self) -> None:
        """Row 0 = the model's current parameters, row b >= 1 = the b-th ``reset_parameters()`` sample (the reference
        resets after every run, optim/mll_torch.py:138-139: same samples, same order)."""
        model = self.model
        start = deepcopy(model.state_dict())
        cur = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
        self.set_row(0, cur)
        for b in range(1, self.B):
            model.reset_parameters()
            self.set_row(b, {n: p.detach() for n, p in model.named_parameters() if p.requires_grad})
        model.load_state_dict(start)

    # -- evaluation -------------------------------------------------------------------------------------------
    def _pre(self, params: Dict[str, torch.Tensor]):
        """One parameter set -> everything the kernels need, through the model's own forward / likelihood / priors."""
        model = self.model
        full = dict(self.fixed)
        full.update(params)
        full.update(self.buffers)
        x = model.train_inputs[0]

        def run(m):
            out = torch.nn.Module.__call__(m, x)            # GP_Plus.forward: features, mean, lazy covariance
            cov = out.lazy_covariance_matrix
            if not isinstance(cov, LazyKernelMatrix):
                raise RuntimeError("the batched fit needs the model's forward to return a lazy kernel covariance")
            noisy = m.likelihood(out).lazy_covariance_matrix
            prior = out.mean.new_zeros(())
            for _, module, pr, closure, _ in m.named_priors():
                prior = prior + pr.log_prob(closure(module)).sum().to(prior)
            static = (noisy.grp, cov.spec.kind, cov.spec.d_split)
            return (cov.U1, cov.spec.w, cov.spec.sf2.reshape(()), noisy.tau.reshape(-1), out.mean, prior), static

        class _Shim(torch.nn.Module):  # functional_call needs a module whose forward does the work
            def __init__(s, inner):
                super().__init__()
                s.inner = inner

            def forward(s):
                tensors, static = run(s.inner)
                self._static = static
                return tensors

        shim = _Shim(model)
        return functional_call(shim, {"inner." + k: v for k, v in full.items()}, ())

    def loss(self) -> torch.Tensor:
        N = self.model.train_targets.shape[0]
        U, w, sf2, tau, mean, prior = torch.vmap(self._pre, in_dims=(0,), randomness="error")(dict(self.theta))
        grp, kind, d_split = self._static
        # leading feature columns produced by the (trainable) latent map of the categorical inputs: they differ from run to
        # run and receive gradients; without them every run sees the same features.  (Inside vmap ``requires_grad`` of
        # the features is not visible to the model's forward, so the count is taken from the model here.)
        dz = int(self.model._features(self.model.train_inputs[0])[1]) if hasattr(self.model, "_features") else 0
        learn_U = U.requires_grad and dz > 0
        if not learn_U:
            U = U[0]  # identical features for every run: share them
        mll = batched_mll(U, w, sf2, tau, mean, self.model.train_targets, grp, kind, d_split, dz if learn_U else 0)
        return -(mll + prior) / N


class _GraphedLossAndGrad:
    """loss (B,) and its gradients for the stacked parameters as ONE replayed HIP graph (``torch.cuda.CUDAGraph``).

    A batched evaluation at the examples' sizes is ~1 ms of GPU work issued by ~1.7 ms of Python (vmap over the model's own
    forward, ~40 library calls, autograd); the launches depend on shapes only.  The graph reads the stacked parameters (updated
    in place by Adam) and the ``active`` mask (updated in place by the driver) and leaves loss and gradients in fixed buffers.
    Nothing may wait for the host inside a capture, so the factorisation makes its no-jitter attempt only
    (``batched._factor_batched``); ``step()`` returns False when any element's status is not zero and the driver then evaluates
    that iteration eagerly (per-element jitter retries) — same numbers either way."""

    def __init__(self, obj: BatchedObjective, params: List[torch.nn.Parameter], active: torch.Tensor):
        from ..backend import get_context
        from ..batched import get_batched_workspace

        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("graph replay needs a GPU device")
        self.params = params
        N = int(obj.model.train_targets.shape[0])
        self.ws = get_batched_workspace(get_context(dev), obj.B, N)  # held: dropped from the cache when another (B, N) is asked for
        self.status_host = torch.zeros(obj.B, dtype=torch.int32).pin_memory()
        self.done = torch.cuda.Event()

        def body():
            loss = obj.loss()
            grads = torch.autograd.grad(torch.nansum(torch.where(active, loss, torch.zeros_like(loss))), params, allow_unused=True)
            return loss.detach(), grads  # (None for a parameter the loss does not depend on, as ``backward`` leaves its .grad)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        from ..graphed import capture_without_gc

        with capture_without_gc(), torch.cuda.graph(self.graph):
            self.loss, self.grads = body()
        self._scratch = get_context(dev)._ws
        self.replays = self.declined = 0

    def step(self) -> bool:
        self.ws.epoch += 1
        self.graph.replay()
        self.status_host.copy_(self.ws.info, non_blocking=True)
        self.done.record()
        self.done.synchronize()
        self.replays += 1
        if bool((self.status_host != 0).any()):
            self.declined += 1
            return False
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return True


def fit_model_torch_batched(model, lr_default: float = 0.01, num_iter: int = 100, num_restarts: int = 0,
                            break_steps: int = 50, verbose: bool = False) -> Tuple[float, List[List[float]]]:
    """Drop-in for ``fit_model_torch`` (same return value) that advances all restarts together."""
    B = num_restarts + 1
    N = int(model.train_targets.shape[0])
    if N > BATCHED_MAX_N:
        # gpp_potrf_batched factors each problem right-looking in leaf steps and rejects larger matrices; at these sizes
        # one evaluation fills the GPU by itself, so the sequential driver loses nothing
        from .mll_torch import fit_model_torch
        return fit_model_torch(model, None, lr_default, num_iter, num_restarts, break_steps, verbose=verbose)
    obj = BatchedObjective(model, B)
    obj.sample_restarts()
    params = list(obj.theta.values())
    opt = torch.optim.Adam(params, lr=lr_default)
    dev = params[0].device
    active = torch.ones(B, dtype=torch.bool, device=dev)
    H = torch.full((num_iter, B), math.nan, dtype=torch.float64, device=dev)  # loss histories stay on the GPU
    last = torch.full((B,), math.inf, dtype=torch.float64, device=dev)
    done_at = num_iter
    graphed = None
    from .. import settings
    if settings.graphed_objective.value() and dev.type == "cuda" and num_iter > 8:
        try:
            graphed = _GraphedLossAndGrad(obj, params, active)
        except RuntimeError as exc:  # a capture the stack refuses: the eager loop below is the same computation — but say so
            import warnings

            warnings.warn(f"fit_model_torch_batched: the step could not be captured as a HIP graph ({exc}); running it eagerly",
                          RuntimeWarning)
            graphed = None
    fit_model_torch_batched.last_graph = None  # (for tests and tools: the finished fit's counters, set below)
    for j in range(num_iter):
        if graphed is not None and graphed.step():
            loss = graphed.loss
        else:
            opt.zero_grad(set_to_none=True)
            loss = obj.loss()
            torch.nansum(torch.where(active, loss, torch.zeros_like(loss))).backward()
        before = [p.detach().clone() for p in params]
        opt.step()
        with torch.no_grad():
            # a stopped run keeps the parameters it stopped with; a run whose covariance was not positive definite (NaN
            # loss, zero gradient through nansum) does not move on Adam's momentum either
            frozen = ~active | ~torch.isfinite(loss.detach())
            for p, old in zip(params, before):
                p.copy_(torch.where(frozen.reshape(-1, *([1] * (p.dim() - 1))), old, p))
            lv = loss.detach()
            lv = torch.where(torch.isfinite(lv), lv, torch.full_like(lv, math.inf))
            H[j] = torch.where(active, lv, torch.full_like(lv, math.nan))
            last = torch.where(active, lv, last)
            if j > break_steps and j % break_steps == 0:
                # reference :126-128, per run: stop when the mean of the previous window is not above the current loss.
                # The reference forms that mean from ``torch.Tensor(loss_hist)`` on the host, i.e. in float32, which is what
                # ends a run on a plateau (differences below ~1e-7 relative round to zero): the same expression on the
                # same device here, so the runs stop where the sequential driver stops them.
                Hc = H[j - break_steps:j + 1].cpu()
                stop_l = [bool((torch.mean(torch.Tensor(Hc[:break_steps, b_].tolist())) - Hc[break_steps, b_].item()) <= 0)
                          for b_ in range(B)]
                stop = active & torch.tensor(stop_l, device=dev)
                active.copy_(active & ~stop)  # (in place: the replayed graph reads this very tensor)
                if verbose:
                    print(f"iter {j}: best loss {float(last.min()):.4f}, {int(active.sum())} of {B} runs active")
                if not bool(active.any()):
                    done_at = j + 1
                    break
    if graphed is not None:
        # only the counters outlive the fit: the graph and its private memory pool are released here
        from types import SimpleNamespace
        fit_model_torch_batched.last_graph = SimpleNamespace(replays=graphed.replays, declined=graphed.declined)
        graphed = None
    Hc = H[:done_at].cpu()
    hist = [[v for v in Hc[:, b_].tolist() if not math.isnan(v)] for b_ in range(B)]
    best = int(torch.argmin(last).item())
    f_inc = float(last[best].item())
    if math.isfinite(f_inc):
        state = deepcopy(model.state_dict())
        state.update(obj.row(best))
        model.load_state_dict(state)
    return f_inc, hist
