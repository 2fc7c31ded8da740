"""Objective + gradient as ONE replayed HIP graph (hipGraph through ``torch.cuda.CUDAGraph``).

At the sizes of the reference's examples (N = 100 ... 2000) one evaluation of ``optim/mll_scipy.py:37-60,101-127`` — model
forward, ~40 library launches, priors, autograd backward, parameter transforms — is ~100 short kernels issued by ~2 ms of
Python, and the L-BFGS loop of ``fit_model_scipy`` runs thousands of them one after the other.  The launches do not depend on
the parameter VALUES, only on shapes: they are captured once, with the parameters read from one flat device vector, and every
later evaluation is a copy of theta, one graph launch and one read-back of (objective, gradient, factorisation status).

Nothing may wait for the host inside a capture, so the factorisation runs its no-jitter attempt only and leaves its status on the
device (``linalg._factor``); a replay whose status is not zero — or whose objective is not finite — returns ``None`` and the caller
evaluates that point eagerly (jitter retries, NotPSDError / NanError).  Limited to the single-stream factorisation (N < 3840):
above that one evaluation is long enough to hide the host, and the look-ahead driver's internal streams do not belong in a graph.
"""
from __future__ import annotations

import contextlib
import gc

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from .linalg import LOOKAHEAD_MIN_N, get_context, get_workspace


@contextlib.contextmanager
def capture_without_gc():
    """Around ``torch.cuda.graph``: no garbage collection WHILE a stream is capturing.  ``torch.cuda.graph`` collects once when it is
    entered, but a collection triggered during the capture can still finalise an older ``CUDAGraph`` (they sit in reference cycles
    with their closures): its destructor releases a memory pool, HIP refuses that while a stream is capturing, and an error thrown
    from a destructor aborts the process ("Fatal Python error: Aborted ... Garbage-collecting" inside a capture: seen once in about
    ten full GPU test runs)."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

__all__ = ["GraphedObjective", "GraphedLossAndGrad"]


class GraphedObjective:
    """``closure()`` -> scalar objective of ``params``; ``evaluate(theta)`` -> (value, gradient) as numpy, or None."""

    def __init__(self, closure: Callable[[], torch.Tensor], params: List[torch.nn.Parameter], n_points: int, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("graph replay needs a GPU device")
        if n_points >= LOOKAHEAD_MIN_N:
            raise RuntimeError(f"graph replay is limited to N < {LOOKAHEAD_MIN_N}")
        self.params, self.device = params, device
        self.sizes = [p.numel() for p in params]
        n = sum(self.sizes)
        self.n = n
        self.theta = torch.zeros(n, dtype=torch.float64, device=device)            # static input of the graph
        self.theta_host = torch.zeros(n, dtype=torch.float64).pin_memory()
        self.out_host = torch.zeros(n + 2, dtype=torch.float64).pin_memory()
        self.done = torch.cuda.Event()
        # the evaluation workspace the captured launches write to: held here, because linalg.get_workspace drops a size when
        # another one is asked for and a replay must never write into memory that has been handed to someone else
        self.gctx = get_context(device)
        self.ws = get_workspace(self.gctx, n_points)
        self.status = self.ws.info

        def body():
            with torch.no_grad():  # scatter theta into the parameters (their storage is the graph's own input)
                i = 0
                for p, k in zip(params, self.sizes):
                    p.copy_(self.theta[i:i + k].view(p.shape))
                    i += k
            value = closure()
            grads = torch.autograd.grad(value, params)
            return torch.cat([value.detach().reshape(1).double()] + [g.reshape(-1).double() for g in grads]
                             + [self.status.reshape(1).double()])

        with torch.no_grad():
            self.theta.copy_(torch.cat([p.detach().reshape(-1).double() for p in params]))
        # warm-up on a side stream (allocations, lazily created workspaces, one-time checks), then the capture
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with capture_without_gc(), torch.cuda.graph(self.graph):
            self.out = body()
        self._lib_scratch = self.gctx._ws  # (same reason: the library's scratch buffer is replaced when a larger one is needed)
        self.last_status = 0
        self.replays = 0   # evaluations asked of the graph
        self.declined = 0  # ... of which it handed back to the eager path (status != 0 or non-finite numbers)

    def evaluate(self, theta: np.ndarray) -> Optional[Tuple[float, np.ndarray]]:
        self.theta_host.copy_(torch.from_numpy(np.ascontiguousarray(theta, dtype=np.float64)))
        self.theta.copy_(self.theta_host, non_blocking=True)
        self.ws.epoch += 1  # the factors in the workspace are overwritten: prediction caches living there are stale
        self.graph.replay()
        self.out_host.copy_(self.out, non_blocking=True)
        self.done.record(torch.cuda.current_stream(self.device))
        self.done.synchronize()
        self.replays += 1
        res = self.out_host.numpy()
        value, status = float(res[0]), res[-1]
        self.last_status = int(status) if np.isfinite(status) else -1
        if status != 0.0 or not np.isfinite(value) or not np.all(np.isfinite(res[1:-1])):
            self.declined += 1
            return None
        return value, res[1:-1].copy()


class GraphedLossAndGrad:
    """``closure()`` -> scalar loss of ``params`` as ONE replayed HIP graph that reads the parameters IN PLACE (an optimizer updates
    their storage between replays) and leaves the loss, its gradients and the factorisation status in fixed buffers: the
    sequential Adam driver's evaluation (reference optim/mll_torch.py:110-118: forward, ``-mll``, ``backward``) at the sizes of the
    reference's examples, where one evaluation is ~100 short launches issued by 1.6-1.9 ms of Python and replays in ~0.6 ms.
    ``step()`` returns the loss as a float and binds the gradient buffers to ``p.grad`` — or returns None (status not zero, or a
    non-finite number) and the caller evaluates that iteration eagerly: jitter retries, NotPSDError / NanError as without the
    graph.  Same kernels on the same data as the eager evaluation: bitwise the same numbers."""

    def __init__(self, closure: Callable[[], torch.Tensor], params: List[torch.nn.Parameter], n_points: int, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("graph replay needs a GPU device")
        if n_points >= LOOKAHEAD_MIN_N:
            raise RuntimeError(f"graph replay is limited to N < {LOOKAHEAD_MIN_N}")
        self.params, self.device = params, device
        self.gctx = get_context(device)
        self.ws = get_workspace(self.gctx, n_points)  # held: see GraphedObjective
        self.head_host = torch.zeros(2, dtype=torch.float64).pin_memory()
        self.done = torch.cuda.Event()

        def body():
            value = closure()
            grads = torch.autograd.grad(value, params, allow_unused=True)
            finite = torch.isfinite(value.detach().double().reshape(1))
            for g in grads:
                if g is not None:
                    finite = finite & torch.isfinite(g.detach()).all().reshape(1)
            # [loss, status]: the status word, or -1 when a number is not finite
            head = torch.cat([value.detach().reshape(1).double(),
                              torch.where(finite, self.ws.info.reshape(1).double(), torch.full((1,), -1.0, dtype=torch.float64, device=device))])
            return head, grads

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                body()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with capture_without_gc(), torch.cuda.graph(self.graph):
            self.head, self.grads = body()
        self._lib_scratch = self.gctx._ws
        self.replays = self.declined = 0
        self.last_status = 0  # status word of the last replay (-1: a non-finite number)
        self.dead = False     # set by the driver when the captured launches must not be replayed any more (a time-out status)

    def step(self) -> Optional[float]:
        self.ws.epoch += 1
        self.graph.replay()
        self.head_host.copy_(self.head, non_blocking=True)
        self.done.record(torch.cuda.current_stream(self.device))
        self.done.synchronize()
        self.replays += 1
        value, status = float(self.head_host[0]), float(self.head_host[1])
        self.last_status = int(status)
        if status != 0.0:
            self.declined += 1
            return None
        for p, g in zip(self.params, self.grads):
            p.grad = g
        return value
