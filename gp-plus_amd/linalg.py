"""Exact-GP linear algebra on the MI355X: one ``torch.autograd.Function`` whose forward and backward are sequences
of libgpp_hip calls (no ATen linear algebra, no CPU fallback).

Replaces, for the path ``optim/mll_torch.py:114-117``:
  forward  = gpytorch ``MultivariateNormal.log_prob`` -> ``inv_quad_logdet`` -> ``psd_safe_cholesky``
             (kernel build K1-K4, Cholesky K5, solve + logdet + quadratic form K6)
  backward = ATen ``cholesky_backward`` + the backward of every N^2 kernel op (K7): Ky^-1 by trtri + lauum, then ONE
             tiled reduction of W = (alpha alpha^T - Ky^-1)/2 against dKy/dtheta.
Both halves are ENQUEUED together in the Function's forward when a gradient is wanted (see ExactMLLFunction), so that
the single host sync of an evaluation — reading the factorisation status — comes after all of its device work.
Jitter policy restates gpytorch.utils.cholesky.psd_safe_cholesky [3P]: 1e-8 * 10^i, i = 0..2 (fp64), warn, then
``NotPSDError``; NaN inputs raise ``NanError``.
"""
from __future__ import annotations

import threading
import os
import warnings
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from .backend import (KIND_RBF, OP_MLL_EVAL, UPLO_FULL, UPLO_UPPER, GppContext, get_context, panel_timed_out, square_buffer)
from .errors import NanError, NotPSDError
from . import settings

__all__ = ["KernelSpec", "exact_mll", "ExactMLLFunction", "EvalWorkspace", "dense_kernel", "cross_kernel",
           "FactorCache", "factorize", "dense_log_prob"]


@dataclass
class KernelSpec:
    """What the fused tile kernel needs: K_ij = sf2 * k(sum_d w_d (u_id-u_jd)^2).  All tensors live on the GPU."""
    w: torch.Tensor            # (D,) weights, autograd-connected to the raw lengthscales
    sf2: torch.Tensor          # () outputscale, autograd-connected
    kind: int = KIND_RBF
    d_split: int = 0


class EvalWorkspace:
    """Device buffers of one N-point evaluation, reused across evaluations (3 N x N fp64 matrices: Ky -> U (upper),
    Linv (lower) + its mirror (upper), scratch / Kinv (lower)).  ``epoch`` increments on every forward so a stale backward can tell its factors were overwritten."""

    def __init__(self, ctx: GppContext, N: int):
        dev = ctx.device
        self.N = N
        self.A = square_buffer(N, dev)
        self.Li = square_buffer(N, dev)
        self.Ki = square_buffer(N, dev)
        self.z = torch.empty(N, dtype=torch.float64, device=dev)
        self.alpha = torch.empty(N, dtype=torch.float64, device=dev)
        self.r = torch.empty(N, dtype=torch.float64, device=dev)
        self.out3 = torch.empty(3, dtype=torch.float64, device=dev)
        self.info = torch.zeros(1, dtype=torch.int32, device=dev)
        self.info_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.info_event = torch.cuda.Event()
        self.epoch = 0


_workspaces: Dict[Tuple[int, int, int], EvalWorkspace] = {}
_workspaces_lock = threading.Lock()
_tls = threading.local()


@contextmanager
def eval_slot(slot: int):
    """Evaluations issued inside this block (by this host thread) use workspace ``slot``.  Independent evaluations
    that run concurrently — e.g. two restarts of a fit driven from two threads on two HIP streams — take distinct
    slots so their N x N buffers do not alias (each slot holds 3 x 8 N^2 bytes)."""
    old = getattr(_tls, "slot", 0)
    _tls.slot = slot
    try:
        yield
    finally:
        _tls.slot = old


def current_slot() -> int:
    return getattr(_tls, "slot", 0)

#: smallest N for which gpp_potrf_ws (with scratch) runs its look-ahead driver on the internal streams (gpp_api.hip)
LOOKAHEAD_MIN_N = 3840

#: optional stage timing (bench.py): when this is a list, every stage appends (name, start_event, end_event) recorded
#: on the stream the kernels are launched on (PyTorch's current stream).
STAGE_EVENTS = None


class _stage:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if STAGE_EVENTS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if STAGE_EVENTS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            STAGE_EVENTS.append((self.name, self.e0, e1))
        return False


def get_workspace(ctx: GppContext, N: int, slot: int = 0) -> EvalWorkspace:
    key = (ctx.index, N, slot)
    ws = _workspaces.get(key)
    if ws is None:
        with _workspaces_lock:
            # keep at most one size per (device, slot): drop others so 3 x 8 N^2 bytes are not held per historical N
            for k in [k for k in _workspaces if k[0] == ctx.index and k[2] == slot]:
                del _workspaces[k]
            ws = EvalWorkspace(ctx, N)
            _workspaces[key] = ws
    return ws


def _as_f64(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float64).contiguous()


_NO_PRESYNC = os.environ.get("GPP_NO_PRESYNC", "0") not in ("", "0")  # experiment knob (see _factor): measured +1 ms at C2 and C4
_SPIN_SYNC = os.environ.get("GPP_SPIN_SYNC", "0") not in ("", "0")


def _factor(ctx: GppContext, ws: EvalWorkspace, U, w, sf2, tau, grp, kind, d_split, after=None) -> float:
    """Build Ky (upper) and factor it, with gpytorch's jitter-retry policy.  Returns the jitter that was needed.
    ``after()`` enqueues whatever follows the factorisation BEFORE ``info`` is read back, so the GPU keeps working while
    the host waits (and afterwards runs the Python between forward and backward); a failed attempt just repeats it."""
    if torch.cuda.is_current_stream_capturing():
        # Inside a HIP-graph capture (gp-plus_amd/graphed.py) nothing may wait for the host: ONE attempt without jitter, the
        # status stays on the device in ``ws.info`` — the owner of the graph reads it with the result of every replay and
        # falls back to this eager path (jitter retries, exceptions) when it is not zero.
        if ws.N >= LOOKAHEAD_MIN_N:
            raise RuntimeError("graph capture of the evaluation is limited to the single-stream factorisation (N < 3840)")
        ctx.kernel_build(U, w, sf2, tau, grp, ws.A, jitter=0.0, kind=kind, d_split=d_split, uplo=UPLO_UPPER)
        ctx.potrf(ws.A, ws.Li, ws.info, ws.Ki)
        if after is not None:
            after()
        return 0.0
    jitters = [0.0] + [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
    attempts = list(jitters)
    while attempts:
        jit = attempts.pop(0)
        # The factorisation's launches (a DAG over the library's internal streams, ~1000 launches at N = 20000) run fastest
        # when they are enqueued while the device executes them, and measurably slower when they were parked in the
        # queues beforehand — N = 20000: potrf 55.7 ms when enqueued on an idle device, 58.0 when enqueued ~1 ms ahead
        # (behind the previous evaluation's gradient reduction), 58.7-59.3 when enqueued a whole evaluation ahead, and
        # the stages of the evaluation that is still running slow down as well (GPU_MAX_HW_QUEUES 4 / 8 / 16 alike; an
        # idle pause alone changes nothing).  So the host waits HERE for whatever is still running on this stream: the
        # Python between two evaluations has overlapped the previous one's inverse stages by now and nothing is exposed.
        # bench.py at N = 20000 on one box: 151-153 ms per evaluation without this wait, 143-144 with it; C3 (N = 10000)
        # 25.4-26.0 -> 24.3-24.8 ms.
        if ws.N >= LOOKAHEAD_MIN_N and not _NO_PRESYNC:  # (below, the factorisation is a single-stream chain and the host is the bottleneck)
            if _SPIN_SYNC:  # experiment knob: poll an event instead of the runtime's blocking wait (wake-up latency)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(ctx.index))
                while not ev.query():
                    pass
            else:
                torch.cuda.current_stream(ctx.index).synchronize()
        # (Measured and NOT adopted: building the first diagonal block's columns first and handing them to the panel stream
        #  while the rest of Ky is written — the build is an unmasked launch that floods every CU, the panel's 32 included,
        #  so the first leaf waits for it anyway: 53.6 ms against 0.64 + 52.4.)
        with _stage("kernel_build"):
            ctx.kernel_build(U, w, sf2, tau, grp, ws.A, jitter=jit, kind=kind, d_split=d_split, uplo=UPLO_UPPER)
        with _stage("potrf"):
            ctx.potrf(ws.A, ws.Li, ws.info, ws.Ki)
        # the second host wait of an evaluation (the reference syncs on loss.item() too) covers the factorisation only: the
        # status goes to pinned host memory behind an event, the rest of the evaluation is enqueued, THEN the host waits
        ws.info_host.copy_(ws.info, non_blocking=True)
        ws.info_event.record(torch.cuda.current_stream(ctx.index))
        if after is not None:
            after()
        ws.info_event.synchronize()
        info = int(ws.info_host[0])
        if panel_timed_out(ctx, info):
            attempts.insert(0, jit)  # not a statement about the matrix: the same attempt again, without the panel
            continue
        if info == 0:
            if jit > 0:
                warnings.warn(f"A not p.d., added jitter of {jit:.1e} to the diagonal", RuntimeWarning)
            return jit
        if jit == 0.0:
            bad = [n for n, t in (("inputs", U), ("weights", w), ("outputscale", sf2), ("noise", tau)) if not torch.isfinite(t).all()]
            if bad:
                raise NanError(f"cholesky: NaN/Inf in {', '.join(bad)} of the covariance")
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitters[-1]:.1e} "
                      f"(leading minor {info}).")


class ExactMLLFunction(torch.autograd.Function):
    """mll = log N(y | mean, sf2*k(U,U;w) + diag(tau[grp]))  with gradients for U[:, :dU], w, sf2, tau, mean, y.

    The whole evaluation — value AND the parameter gradients (K7: Ky^-1 by trtri + lauum, one fused reduction) — is
    enqueued in ``forward`` ahead of the single host sync on the factorisation status whenever an input needs a gradient:
    the device then works through the Python that lies between the reference's ``-mll(...)`` and ``loss.backward()``
    (optim/mll_torch.py:116-117), and ``backward`` only scales the stored gradients by the incoming one."""

    @staticmethod
    def forward(ctx, U, w, sf2, tau, mean, y, grp, kind, d_split, dU, slot):
        get_context(U.device)  # raises GppError for anything but a GPU: there is no CPU path
        with torch.cuda.device(U.device):  # streams, events and the library's launches all refer to the model's GPU
            return ExactMLLFunction._forward(ctx, U, w, sf2, tau, mean, y, grp, kind, d_split, dU, slot)

    @staticmethod
    def _forward(ctx, U, w, sf2, tau, mean, y, grp, kind, d_split, dU, slot):
        dev = U.device
        gctx = get_context(dev)
        N, D = U.shape
        Ud, wd, sd, td = _as_f64(U.detach(), dev), _as_f64(w.detach(), dev), _as_f64(sf2.detach().reshape(1), dev), \
            _as_f64(tau.detach().reshape(-1), dev)
        S = td.numel()
        if grp is not None and grp.dtype != torch.int32:
            grp = grp.to(torch.int32)
        ws = get_workspace(gctx, N, slot)
        ws.epoch += 1
        torch.sub(_as_f64(y.detach(), dev), _as_f64(mean.detach(), dev), out=ws.r)
        need = ctx.needs_input_grad
        need_grad = any(need[:6])
        need_U = need[0] and dU > 0
        g_w = g_s = g_t = g_Ud = None
        if need_grad:
            g_w = torch.empty(D, dtype=torch.float64, device=dev)
            g_s = torch.empty(1, dtype=torch.float64, device=dev)
            g_t = torch.empty(S, dtype=torch.float64, device=dev)
            g_Ud = torch.empty(N, dU, dtype=torch.float64, device=dev) if need_U else None

        def rest():
            with _stage("trtri"):
                gctx.trtri(ws.A, ws.Li, ws.Ki)
            with _stage("mll_reduce"):
                gctx.mll_reduce(ws.A, ws.Li, ws.r, ws.z, ws.out3)
            if not need_grad:
                return
            # (Measured and NOT adopted: z, the MLL scalars and alpha on a side stream beside the LAUUM launch — the extra
            #  stream perturbs the hardware-queue mapping of the NEXT evaluation's factorisation DAG: potrf 53.3 -> 58-60 ms,
            #  137 -> 142-144 ms per evaluation at N = 20000, tools/attic/side_ab.py.)
            with _stage("alpha"):
                gctx.alpha(ws.Li, ws.z, ws.alpha)
            with _stage("lauum"):
                gctx.lauum(ws.Li, ws.Ki)
            with _stage("grad_reduce"):
                gctx.grad_reduce(Ud, wd, sd, grp, S, ws.alpha, ws.Ki, dU if need_U else 0, g_w, g_s, g_t, g_Ud, kind=kind,
                                 d_split=d_split)

        _factor(gctx, ws, Ud, wd, sd, td, grp, kind, d_split, after=rest)
        ctx.saved = (g_w, g_s, g_t, g_Ud, ws.alpha.clone() if need_grad else None, (N, D, dU))
        ctx.in_dtypes = (U.dtype, w.dtype, sf2.dtype, tau.dtype, mean.dtype, y.dtype)
        ctx.shapes = (sf2.shape, tau.shape)
        return ws.out3[2].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g_w, g_s, g_t, g_Ud, alpha, (N, D, dU) = ctx.saved
        need = ctx.needs_input_grad
        go = grad_out.to(torch.float64)
        g_U = None
        if need[0]:
            g_U = torch.zeros(N, D, dtype=torch.float64, device=alpha.device)
            if g_Ud is not None:
                g_U[:, :dU] = g_Ud
        dt = ctx.in_dtypes
        sf2_shape, tau_shape = ctx.shapes
        return (None if g_U is None else (go * g_U).to(dt[0]),
                (go * g_w).to(dt[1]) if need[1] else None,
                (go * g_s).reshape(sf2_shape).to(dt[2]) if need[2] else None,
                (go * g_t).reshape(tau_shape).to(dt[3]) if need[3] else None,
                (go * alpha).to(dt[4]) if need[4] else None,
                (-go * alpha).to(dt[5]) if need[5] else None,
                None, None, None, None, None)


def exact_mll(U: torch.Tensor, spec: KernelSpec, tau: torch.Tensor, mean: torch.Tensor, y: torch.Tensor,
              grp: Optional[torch.Tensor] = None, n_grad_dims: Optional[int] = None, slot: Optional[int] = None) -> torch.Tensor:
    """log N(y | mean, Ky) on the GPU; differentiable w.r.t. U[:, :n_grad_dims], spec.w, spec.sf2, tau, mean, y."""
    if slot is None:
        slot = current_slot()
    if n_grad_dims is None:
        n_grad_dims = U.shape[1] if U.requires_grad else 0
    shard = settings.sharded_evaluation.value()
    if shard is not None:
        from .sharded import sharded_mll
        return sharded_mll(U, spec.w, spec.sf2, tau, mean, y, grp, spec.kind, spec.d_split, int(n_grad_dims),
                           group=shard.get("group"), nb=int(shard.get("nb", 1024)))
    return ExactMLLFunction.apply(U, spec.w, spec.sf2, tau, mean, y, grp, spec.kind, spec.d_split, int(n_grad_dims), slot)


# ---------------------------------------------------------------------------------------------------
# dense evaluations (no autograd): .evaluate(), cross covariances, prediction
# ---------------------------------------------------------------------------------------------------
@torch.no_grad()
def dense_kernel(U: torch.Tensor, spec: KernelSpec, tau: Optional[torch.Tensor] = None,
                 grp: Optional[torch.Tensor] = None, jitter: float = 0.0) -> torch.Tensor:
    """Dense N x N covariance (full symmetric) — what ``lazy.evaluate()`` returns (models/gp_plus.py:474)."""
    dev = U.device
    gctx = get_context(dev)
    N = U.shape[0]
    out = square_buffer(N, dev)
    gctx.kernel_build(_as_f64(U, dev), _as_f64(spec.w, dev), _as_f64(spec.sf2.reshape(1), dev),
                      None if tau is None else _as_f64(tau.reshape(-1), dev),
                      None if grp is None else grp.to(torch.int32), out, jitter=jitter, kind=spec.kind,
                      d_split=spec.d_split, uplo=UPLO_FULL)
    return out


@torch.no_grad()
def cross_kernel(Ua: torch.Tensor, Ub: torch.Tensor, spec: KernelSpec) -> torch.Tensor:
    dev = Ua.device
    gctx = get_context(dev)
    M, N = Ua.shape[0], Ub.shape[0]
    ld = max(16, (N + 15) // 16 * 16)
    out = torch.empty((M, ld), dtype=torch.float64, device=dev)[:, :N]
    gctx.cross_kernel(_as_f64(Ua, dev), _as_f64(Ub, dev), _as_f64(spec.w, dev), _as_f64(spec.sf2.reshape(1), dev), out,
                      kind=spec.kind, d_split=spec.d_split)
    return out


class FactorCache:
    """Cholesky factor, its inverse and alpha = Ky^-1 (y - m) of the training covariance: the analogue of gpytorch's
    prediction strategy caches (mean_cache / covar_cache) used by models/gpregression.py:122-149."""

    def __init__(self, gctx, L, Linv, alpha, U, spec, jitter, ws=None, z=None, refactor=None):
        self.gctx, self.L, self.Linv, self.alpha, self.U, self.spec, self.jitter = gctx, L, Linv, alpha, U, spec, jitter
        self.z = z  # Linv (y - m): the mean of a prediction that also wants the variance is V z
        # L and Linv live in the shared prediction workspace: another model's factorisation of the same size overwrites
        # them.  ``stale()`` tells the owner to factor again instead of predicting from someone else's matrices.
        self._ws, self._epoch = ws, (ws.epoch if ws is not None else 0)
        self._refactor = refactor  # (tau, grp, r = y - m): what ``refresh`` needs besides U and spec (O(N) copies)

    def stale(self) -> bool:
        return self._ws is not None and self._ws.epoch != self._epoch

    def refresh(self) -> None:
        """Factor again when another model of the same size has taken the shared workspace since (two GPs fitted on one X
        whose predictions are read alternately): the lazily evaluated variance of an EARLIER prediction stays valid, as it
        is in gpytorch, whose prediction strategy owns its caches.  Same inputs, same jitter schedule: the same factor."""
        if not self.stale():
            return
        if self._refactor is None:
            raise RuntimeError("the prediction workspace was reused by another model and this cache cannot be rebuilt")
        tau, grp, r = self._refactor
        with torch.cuda.device(self.U.device):
            fresh = _factorize(self.U, self.spec, tau, grp, torch.zeros_like(r), r)
        self.L, self.Linv, self.alpha, self.z, self.jitter = fresh.L, fresh.Linv, fresh.alpha, fresh.z, fresh.jitter
        self._ws, self._epoch = fresh._ws, fresh._epoch


@torch.no_grad()
def factorize(U, spec: KernelSpec, tau, grp, mean, y) -> FactorCache:
    get_context(U.device)
    with torch.cuda.device(U.device):
        return _factorize(U, spec, tau, grp, mean, y)


def _factorize(U, spec: KernelSpec, tau, grp, mean, y) -> FactorCache:
    dev = U.device
    gctx = get_context(dev)
    N = U.shape[0]
    Ud, wd = _as_f64(U, dev), _as_f64(spec.w, dev)
    sd, td = _as_f64(spec.sf2.reshape(1), dev), _as_f64(tau.reshape(-1), dev)
    if grp is not None:
        grp = grp.to(torch.int32)
    ws = get_workspace(gctx, N, slot=-1)
    ws.epoch += 1
    jit = _factor(gctx, ws, Ud, wd, sd, td, grp, spec.kind, spec.d_split)
    gctx.trtri(ws.A, ws.Li, ws.Ki)
    torch.sub(_as_f64(y, dev), _as_f64(mean, dev), out=ws.r)
    gctx.mll_reduce(ws.A, ws.Li, ws.r, ws.z, ws.out3)
    gctx.alpha(ws.Li, ws.z, ws.alpha)
    return FactorCache(gctx, ws.A, ws.Li, ws.alpha.clone(), Ud, KernelSpec(wd, sd.reshape(()), spec.kind, spec.d_split), jit, ws,
                       z=ws.z.clone(), refactor=(td.clone(), None if grp is None else grp.clone(), ws.r.clone()))


@torch.no_grad()
def predict_from_cache(cache: FactorCache, Us: torch.Tensor, need_var: bool = True, need_V: bool = False):
    """K8: mean contribution K_*N alpha and prior-minus-explained variance; optionally V = K_*N Linv^T.
    Mean only: the [test][train] cross block and one row reduction, O(M N).  With the variance: the TRANSPOSED cross block, so that
    V = Kns^T Linv^T is the row-contiguous TN product (gpp_predict_tn), and both outputs come from one pass over V."""
    dev = Us.device
    gctx = cache.gctx
    M, N = Us.shape[0], cache.U.shape[0]
    mean = torch.empty(M, dtype=torch.float64, device=dev)
    if not (need_var or need_V):
        Ksn = cross_kernel(Us, cache.U, cache.spec)
        gctx.predict(cache.Linv, cache.alpha, Ksn, None, None, mean, None)
        return mean, None, None
    Kns = cross_kernel(cache.U, Us, cache.spec)  # N x M
    ldv = max(16, (N + 15) // 16 * 16)
    V = torch.empty((M, ldv), dtype=torch.float64, device=dev)[:, :N]
    kss = cache.spec.sf2.reshape(1).expand(M).contiguous()
    var = torch.empty(M, dtype=torch.float64, device=dev)
    gctx.predict_tn(cache.Linv, cache.z, Kns, kss, V, mean, var)
    return mean, var, V


@torch.no_grad()
def dense_log_prob(cov: torch.Tensor, diff: torch.Tensor) -> torch.Tensor:
    """log N(diff | 0, cov) for a dense covariance (used by ``evaluation``'s joint NLPD, models/gp_plus.py:900-903)."""
    dev = cov.device
    gctx = get_context(dev)
    N = cov.shape[0]
    A = square_buffer(N, dev)
    A.copy_(cov)
    Li, T = square_buffer(N, dev), square_buffer(N, dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    jitters = [0.0] + [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
    prev, attempts, fresh = 0.0, list(jitters), True
    while attempts:
        jit = attempts.pop(0)
        if not fresh:
            A.copy_(cov)
            A.diagonal().add_(jit)
        fresh = False
        gctx.potrf(A, Li, info)
        status = int(info.item())
        if panel_timed_out(gctx, status):
            attempts.insert(0, jit)  # the same attempt again, without the cooperative panel
            continue
        if status == 0:
            break
        prev = jit
    else:
        raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {prev:.1e}.")
    gctx.trtri(A, Li, T)
    z = torch.empty(N, dtype=torch.float64, device=dev)
    out3 = torch.empty(3, dtype=torch.float64, device=dev)
    gctx.mll_reduce(A, Li, _as_f64(diff, dev), z, out3)
    return out3[2].clone()
