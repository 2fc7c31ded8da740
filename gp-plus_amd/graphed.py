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

__all__ = ["GraphedObjective", "GraphedLossAndGrad", "GraphedSegment"]

import os as _os

#: capture_error_mode of every capture here (experiment knob GPP_CAPTURE_MODE: global / thread_local / relaxed)
_CAPTURE_MODE = _os.environ.get("GPP_CAPTURE_MODE", "global")


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
        with capture_without_gc(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
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
        with capture_without_gc(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
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


# ---------------------------------------------------------------------------------------------------
# The HOST segments of an evaluation as graphs (round 6; N >= 3840, where the whole evaluation cannot be one graph)
# ---------------------------------------------------------------------------------------------------
class GraphedSegment:
    """A piece of the model's OWN code — ``fn()`` -> tuple of tensors, a function of ``params`` through ordinary PyTorch ops — as two
    replayed HIP graphs: its forward, and its backward (``torch.autograd.grad`` of the outputs w.r.t. the parameters for given output
    gradients), tied into autograd by one ``torch.autograd.Function`` node.

    Why: above N = 3840 the factorisation runs on the library's internal streams and does not belong in a graph, so an evaluation
    through the plain API (``model(*x)``, ``-mll(...)``, ``backward()``: optim/mll_torch.py:114-117) issued ~170 element-wise kernels of
    3-4 us for the parameter transforms (models/gp_plus.py:243-295), the priors (priors/horseshoe.py:63-66, gpregression.py:84-115), the
    manifold map and their backward, one Python call each: 0.4 ms of device time and a 0.4 ms gap in front of the next covariance build
    at C3 (profiles/r05_hbm_probe.txt), the whole cost of an evaluation at small N.  The kernels are the same ones, on the same data,
    in the same order — the numbers are bitwise those of the eager evaluation — but the host issues four graph launches instead.
    The outputs live in the graph's static buffers: they are valid until the next replay (the next evaluation of the same model)."""

    def __init__(self, fn: Callable[[], tuple], params: List[torch.nn.Parameter], device, module: Optional[torch.nn.Module] = None):
        """``module`` (whose parameters ``fn`` reads; ``params`` is then ignored): the captured code runs on SHADOW leaves — copies of the
        module's parameters swapped in for the capture (torch.nn.utils.stateless), refreshed by copies that are part of the forward
        graph.  The real parameters then appear in ONE place only, as inputs of the segment's autograd node in the caller's own
        graph, so their AccumulateGrad nodes are created on the caller's stream.  Without it (round 6's first form) the captured
        autograd graph — which must stay alive — held the real parameters' AccumulateGrad nodes on the CAPTURE stream, every later
        backward accumulated p.grad there, and a second active stream perturbs the factorisation's streams: C2 +1.9 ms."""
        device = torch.device(device)
        self.device = device
        if module is not None:
            from torch.nn.utils import stateless

            named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
            self.params = [p for _, p in named]
            self.leaves = [p.detach().clone().requires_grad_(True) for p in self.params]
            self.names = [n for n, _ in named]

            def run():
                with stateless._reparametrize_module(module, dict(zip(self.names, self.leaves))):
                    return fn()
        else:
            self.params = [p for p in params if p.requires_grad]
            self.leaves = self.params
            run = fn

        def grads_of(outs, gouts):
            diff = [(o, g) for o, g in zip(outs, gouts) if g is not None]
            if not diff or not self.leaves:
                return [None] * len(self.leaves)
            return list(torch.autograd.grad([o for o, _ in diff], self.leaves, [g for _, g in diff], allow_unused=True))

        def refresh():
            if self.leaves is not self.params:
                with torch.no_grad():
                    for leaf, p in zip(self.leaves, self.params):
                        leaf.copy_(p)

        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                refresh()
                outs = run()
                used = grads_of(outs, [torch.zeros_like(o) if o.requires_grad else None for o in outs])
                # Only the parameters the segment really depends on are inputs of its autograd node.  (Not a nicety: with a
                # parameter among the node's inputs that the segment does not use, the NEXT graph capture in the process — while
                # such a node is alive — dies in hipStreamEndCapture on this stack (ROCm 7.2 / PyTorch 2.10); minimal reproducer
                # tools/attic/dev/segment_probe.py with DISJOINT=1.)
                keep = [g is not None for g in used]
                shadowed = self.leaves is not self.params
                self.params = [p for p, k in zip(self.params, keep) if k]
                self.leaves = [p for p, k in zip(self.leaves, keep) if k] if shadowed else self.params
                if shadowed:
                    self.names = [n for n, k in zip(self.names, keep) if k]
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.fwd = torch.cuda.CUDAGraph()
        # Both captures on the stream of the warm-up passes: the leaves' AccumulateGrad nodes were created there, and autograd hands a
        # gradient to such a node on the NODE's stream — captured from another stream, the backward forks onto it (PyTorch warns:
        # "The AccumulateGrad node's stream does not match ..."), the graph gets a second branch, and a replay then occupies a second
        # hardware queue beside the caller's: what perturbs the factorisation's CU-masked streams (settings.graphed_segments' note).
        cap = side if _os.environ.get("GPP_SEGMENT_CAPTURE_STREAM", "side") == "side" else None
        with capture_without_gc(), torch.cuda.graph(self.fwd, stream=cap, capture_error_mode=_CAPTURE_MODE):
            refresh()  # (part of the graph: the shadow leaves take the parameters' current values at every replay)
            self.outs = tuple(run())
        self.gouts = [torch.zeros_like(o) if o.requires_grad else None for o in self.outs]
        self.bwd = torch.cuda.CUDAGraph()
        with capture_without_gc(), torch.cuda.graph(self.bwd, pool=self.fwd.pool(), stream=cap, capture_error_mode=_CAPTURE_MODE):
            self.grads = grads_of(self.outs, self.gouts)
        # (The captured autograd graph stays alive: the backward graph replays into the activations it holds.)
        self.replays = 0

    def __call__(self) -> tuple:
        """The outputs of this evaluation (autograd-connected to the parameters through ONE node)."""
        return _SegmentFunction.apply(self, *self.params)


class _SegmentFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg: GraphedSegment, *params):
        seg.fwd.replay()
        seg.replays += 1
        ctx.seg = seg
        outs = tuple(o.detach() for o in seg.outs)
        ctx.mark_non_differentiable(*[o for o, g in zip(outs, seg.gouts) if g is None])
        return outs

    @staticmethod
    def backward(ctx, *gs):
        seg = ctx.seg
        for static, g in zip(seg.gouts, gs):
            if static is None:
                continue
            if g is None:
                static.zero_()
            else:
                static.copy_(g)
        seg.bwd.replay()
        return (None,) + tuple(None if g is None else g.detach() for g in seg.grads)


def segment_key(params, *tensors) -> tuple:
    """What a captured segment depends on besides the parameters' VALUES: their storage, shape, dtype and requires_grad flags, and
    the identity of the data tensors it reads."""
    return (tuple((p.data_ptr(), tuple(p.shape), p.dtype, p.requires_grad) for p in params),
            tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in tensors))


def segments_apply(n_points: int, device) -> bool:
    """Graphed host segments apply to evaluations the whole-evaluation graphs do not cover (N >= 3840), on a GPU, with autograd on,
    outside any capture — when ``settings.graphed_segments`` is on (default off: measured slower, see there)."""
    from . import settings

    device = torch.device(device)
    return (settings.graphed_segments.value() and device.type == "cuda" and n_points >= LOOKAHEAD_MIN_N and torch.is_grad_enabled()
            and not torch.cuda.is_current_stream_capturing())
