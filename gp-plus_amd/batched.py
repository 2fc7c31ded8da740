"""Batched evaluation: B independent exact-GP log-likelihoods of the SAME data size in every kernel launch.

The reference fits a model by running its restarts one after the other (optim/mll_torch.py:99-141: ``num_restarts + 1``
Adam runs of 100 evaluations each; optim/mll_scipy.py:281-295 farms L-BFGS starts out to joblib workers).  For the sizes
its examples use (N = 100 ... 500) one evaluation is a chain of a few dozen latency-bound launches that leave the GPU
almost empty, so the MI355X-native form of "restart parallelism" is to evaluate ALL restarts in the same launches: every
kernel of ``csrc/`` takes a batch dimension (``gpp_*_batched``), and 64 restarts cost the latency chain of one.

``BatchedMLLFunction`` is ``linalg.ExactMLLFunction`` with a leading batch dimension on every argument:
    U (N, D) shared or (B, N, D);  w (B, D);  sf2 (B,);  tau (B, S);  mean (B, N);  y (N,) or (B, N)  ->  mll (B,)
Elements whose covariance is not positive definite after the jitter schedule (gpytorch's 1e-8 * 10^i) return NaN and
zero gradients instead of raising: one bad restart must not stop the others (the drivers score it +inf).
"""
from __future__ import annotations

import threading
from typing import Dict, Optional, Tuple

import torch

from .backend import KIND_RBF, UPLO_UPPER, GppContext, get_context
from . import settings

__all__ = ["BatchedWorkspace", "BatchedMLLFunction", "batched_mll"]


class BatchedWorkspace:
    def __init__(self, ctx: GppContext, B: int, N: int):
        dev = ctx.device
        self.B, self.N = B, N
        self.A, self.Li, self.Ki = (ctx.batched_buffer(B, N) for _ in range(3))
        self.r, self.z, self.alpha = (ctx.batched_vector(B, N) for _ in range(3))
        self.out3 = torch.empty(B, 3, dtype=torch.float64, device=dev)
        self.info = torch.zeros(B, dtype=torch.int32, device=dev)
        self.info_host = torch.zeros(B, dtype=torch.int32).pin_memory()
        self.info_event = torch.cuda.Event()
        self.epoch = 0


_workspaces: Dict[Tuple[int, int, int], BatchedWorkspace] = {}
_lock = threading.Lock()


def get_batched_workspace(ctx: GppContext, B: int, N: int) -> BatchedWorkspace:
    key = (ctx.index, B, N)
    ws = _workspaces.get(key)
    if ws is None:
        with _lock:
            for k in [k for k in _workspaces if k[0] == ctx.index]:
                del _workspaces[k]
            ws = BatchedWorkspace(ctx, B, N)
            _workspaces[key] = ws
    return ws


def _f64(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float64).contiguous()


def _factor_batched(gctx: GppContext, ws: BatchedWorkspace, U, w, sf2, tau, grp, kind, d_split, after=None) -> torch.Tensor:
    """Build + factor all elements; failing ones are retried with gpytorch's jitter schedule added to THEIR noise.
    Returns the boolean mask (B,) of elements that are positive definite in the end.  ``after()`` enqueues the rest of
    the evaluation before the host waits for the status words (one wait per attempt, covering the factorisation only)."""
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture (optim/mll_batched.py) nothing may wait for the host: ONE attempt without jitter; the status
        # words stay on the device and the owner of the graph re-runs the step eagerly when any of them is not zero
        gctx.kernel_build_batched(U, w, sf2, tau, grp, ws.A, kind=kind, d_split=d_split, uplo=UPLO_UPPER)
        gctx.potrf_batched(ws.A, ws.Li, ws.info)
        if after is not None:
            after()
        return ws.info == 0
    jitters = [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
    extra = torch.zeros(ws.B, 1, dtype=torch.float64, device=U.device)
    ok = None
    for attempt in range(len(jitters) + 1):
        gctx.kernel_build_batched(U, w, sf2, tau + extra, grp, ws.A, kind=kind, d_split=d_split, uplo=UPLO_UPPER)
        gctx.potrf_batched(ws.A, ws.Li, ws.info)
        ok = ws.info == 0
        ws.info_host.copy_(ws.info, non_blocking=True)
        ws.info_event.record()
        if after is not None:
            after()
        ws.info_event.synchronize()
        if bool((ws.info_host == 0).all()) or attempt == len(jitters):
            break
        extra = torch.where(ok.unsqueeze(1), extra, torch.full_like(extra, jitters[attempt]))
    return ok


class BatchedMLLFunction(torch.autograd.Function):
    """B independent evaluations per launch.  As in linalg.ExactMLLFunction the gradients are produced in ``forward``
    (when any input needs one), ahead of the host's wait for the factorisation status; ``backward`` only scales them."""

    @staticmethod
    def forward(ctx, U, w, sf2, tau, mean, y, grp, kind, d_split, dU):
        dev = w.device
        gctx = get_context(dev)
        B, D = w.shape
        N = U.shape[-2]
        Ud, wd, sd, td = _f64(U, dev), _f64(w, dev), _f64(sf2.reshape(B), dev), _f64(tau.reshape(B, -1), dev)
        S = td.shape[1]
        if grp is not None and grp.dtype != torch.int32:
            grp = grp.to(torch.int32)
        ws = get_batched_workspace(gctx, B, N)
        ws.epoch += 1
        torch.sub(_f64(y, dev).expand(B, N), _f64(mean, dev).expand(B, N), out=ws.r)
        need = ctx.needs_input_grad
        need_grad = any(need[:6])
        need_U = need[0] and dU > 0
        g_w = g_s = g_t = g_Ud = None
        if need_grad:
            g_w = torch.empty(B, D, dtype=torch.float64, device=dev)
            g_s = torch.empty(B, dtype=torch.float64, device=dev)
            g_t = torch.empty(B, S, dtype=torch.float64, device=dev)
            g_Ud = torch.empty(B, N, dU, dtype=torch.float64, device=dev) if need_U else None

        def rest():
            gctx.trtri_batched(ws.A, ws.Li, ws.Ki)
            gctx.mll_reduce_batched(ws.A, ws.Li, ws.r, ws.z, ws.out3)
            if not need_grad:
                return
            gctx.alpha_batched(ws.Li, ws.z, ws.alpha)
            gctx.lauum_batched(ws.Li, ws.Ki)
            gctx.grad_reduce_batched(Ud, wd, sd, grp, S, ws.alpha, ws.Ki, dU if need_U else 0, g_w, g_s, g_t, g_Ud,
                                     kind=kind, d_split=d_split)

        ok = _factor_batched(gctx, ws, Ud, wd, sd, td, grp, kind, d_split, after=rest)
        ctx.saved = (g_w, g_s, g_t, g_Ud, ws.alpha.clone() if need_grad else None, ok, (B, N, Ud.shape[-1], dU))
        ctx.in_dtypes = (U.dtype, w.dtype, sf2.dtype, tau.dtype, mean.dtype, y.dtype)
        ctx.shapes = (U.shape, sf2.shape, tau.shape, mean.shape, y.shape)
        mll = ws.out3[:, 2].clone()
        return torch.where(ok, mll, torch.full_like(mll, float("nan")))

    @staticmethod
    def backward(ctx, grad_out):
        g_w, g_s, g_t, g_Ud, alpha_all, ok, (B, N, Dfull, dU) = ctx.saved
        dev = alpha_all.device
        U_shape, sf2_shape, tau_shape, mean_shape, y_shape = ctx.shapes
        need_U = g_Ud is not None
        go = torch.where(ok, grad_out.to(torch.float64), torch.zeros_like(grad_out, dtype=torch.float64))  # failed: no gradient
        dt = ctx.in_dtypes
        alpha = torch.where(ok.unsqueeze(1), alpha_all, torch.zeros_like(alpha_all))
        g_U = None
        if ctx.needs_input_grad[0]:
            full = torch.zeros(B, N, Dfull, dtype=torch.float64, device=dev)
            if need_U:
                full[:, :, :dU] = torch.where(ok.view(B, 1, 1), g_Ud, torch.zeros_like(g_Ud)) * go.view(B, 1, 1)
            g_U = (full if len(U_shape) == 3 else full.sum(0)).to(dt[0])

        def red(t, shape):  # gradient of a broadcast argument: sum over the batch
            return t.reshape(shape) if len(shape) == t.dim() else t.sum(0).reshape(shape)

        g_mean = go.unsqueeze(1) * alpha
        nanfree = lambda t: torch.where(torch.isfinite(t), t, torch.zeros_like(t))
        return (g_U,
                nanfree(go.unsqueeze(1) * g_w).to(dt[1]) if ctx.needs_input_grad[1] else None,
                nanfree(go * g_s).reshape(sf2_shape).to(dt[2]) if ctx.needs_input_grad[2] else None,
                nanfree(go.unsqueeze(1) * g_t).reshape(tau_shape).to(dt[3]) if ctx.needs_input_grad[3] else None,
                red(g_mean, mean_shape).to(dt[4]) if ctx.needs_input_grad[4] else None,
                red(-g_mean, y_shape).to(dt[5]) if ctx.needs_input_grad[5] else None,
                None, None, None, None)


def batched_mll(U: torch.Tensor, w: torch.Tensor, sf2: torch.Tensor, tau: torch.Tensor, mean: torch.Tensor, y: torch.Tensor,
                grp: Optional[torch.Tensor] = None, kind: int = KIND_RBF, d_split: int = 0, n_grad_dims: int = 0) -> torch.Tensor:
    """(B,) log N(y | mean_b, sf2_b k(U_b, U_b; w_b) + diag(tau_b[grp])) for B parameter sets in one pass."""
    return BatchedMLLFunction.apply(U, w, sf2, tau, mean, y, grp, kind, d_split, int(n_grad_dims))
