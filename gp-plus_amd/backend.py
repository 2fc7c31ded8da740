"""Torch-tensor front of the C ABI: one :class:`GppContext` per GPU owns the library handle, follows PyTorch's
current HIP stream and keeps the scratch workspace.  PyTorch is used for device memory and streams only; every
number is produced by the kernels in ``csrc/``.

Reference boundary this replaces: the gpytorch/ATen operators reached from ``optim/mll_torch.py:112-117`` and
``models/gpregression.py:122-149``.
"""
from __future__ import annotations

import atexit
import ctypes
import os
import threading
import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import GppError, check

KIND_RBF, KIND_MATERN32, KIND_MATERN52 = 0, 1, 2
UPLO_FULL, UPLO_LOWER, UPLO_UPPER = 0, 1, 2
OPT_COOP_PANEL, OPT_PANEL_FAULT, OPT_PANEL_TIMEOUT_MS, OPT_EXEC_SCHED, OPT_DAG_SCHED = 1, 2, 3, 4, 5  # gpp_set_option (include/gpp.h)
OP_MLL_EVAL, OP_PREDICT = 0, 1
#: the tile kernels stage at most this many feature columns (manifold + quantitative) per point in LDS (gpp_build.hip DMAX)
MAX_FEATURES = 64


def _check_features(D: int) -> None:
    if D < 1 or D > MAX_FEATURES:
        raise GppError(f"the covariance kernels take 1..{MAX_FEATURES} feature columns per point (got {D})")

#: one context (library handle + stream binding + scratch) per (device, host thread): concurrent evaluations driven
#: from different threads on different HIP streams never share a handle
_contexts: Dict[Tuple[int, int], "GppContext"] = {}
_contexts_lock = threading.Lock()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda:
        raise GppError(f"{name} must live on the GPU (got {t.device}); libgpp_hip has no CPU path")
    if t.dtype != dtype:
        raise GppError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous() and t.dim() == 1:
        raise GppError(f"{name} must be contiguous")


#: status word of a factorisation whose cooperative panel kernel gave up waiting for one of its own work-groups (gpp_leaf.hip:
#: a wait is abandoned after ~1 s instead of hanging the GPU).  NOT "matrix not positive definite": adding jitter cannot help.
INFO_PANEL_TIMEOUT = 1 << 30
#: the same for a wait of one of the DAG executor's work-groups (gpp_dag_f64 and its gate kernels): bit 29 on top
INFO_EXEC_TIMEOUT = (1 << 30) | (1 << 29)


def check_status(info: int) -> None:
    """Raise for the status words that are not LAPACK's "leading minor k is not positive definite"."""
    if info >= INFO_PANEL_TIMEOUT:
        what = "an executor launch" if info & (1 << 29) else "the cooperative panel kernel"
        raise GppError(f"gpp_potrf: {what} timed out waiting for one of its work-groups (another kernel holding the stream's CUs "
                       "for seconds, or a caller-supplied CU-masked stream with fewer CUs than the launch assumed); "
                       "GPP_COOP_PANEL=0 selects the leaf-step factorisation, GPP_DAG_SCHED=0 the launch-per-product one")


def panel_timed_out(ctx: "GppContext", info: int) -> bool:
    """True when ``info`` is a time-out status the context can answer by switching a cooperative path off; the caller then factors
    again.  The DAG EXECUTOR's time-out (bit 29: its work-groups wait for the panel stream's launches — under co-tenancy or serialised
    dispatch they give up first) switches only the executor off: the factorisation falls back to launch-per-product with the
    cooperative panel, which needs 32 resident work-groups.  The PANEL's own
    time-out switches the panel off (and with it everything built on it): leaf-step launches need no co-residency at all."""
    if info < INFO_PANEL_TIMEOUT:
        return False
    import warnings

    ms = info & 0xFFFFF
    if info & (1 << 29):
        if not ctx.dag_sched:
            check_status(info)  # cannot happen without the executor: report it
        warnings.warn(f"libgpp_hip: an executor launch timed out after {ms} ms (another tenant of this GPU held part of its CUs, or "
                      "dispatches are serialised); this context now factors with one launch per product", RuntimeWarning)
        ctx.set_option(OPT_DAG_SCHED, 0)
        return True
    if not ctx.coop_panel:
        check_status(info)  # cannot happen without a panel: report it
    warnings.warn(f"libgpp_hip: a cooperative panel launch timed out after {ms} ms (another tenant of this GPU held part "
                  "of its CUs); this context now factors with leaf-step launches", RuntimeWarning)
    ctx.set_option(OPT_COOP_PANEL, 0)
    return True


def square_buffer(n: int, device) -> torch.Tensor:
    """Uninitialised n x n fp64 matrix whose rows are padded to a multiple of 16 doubles (128-byte lines)."""
    ld = max(16, (n + 15) // 16 * 16)
    return torch.empty((n, ld), dtype=torch.float64, device=device)[:, :n]


def _ld(m: torch.Tensor) -> int:
    if m.dim() != 2 or m.stride(1) != 1:
        raise GppError("matrix must be 2-D with unit column stride")
    return m.stride(0) if m.shape[0] > 1 else max(m.shape[1], m.stride(0))


def _on_own_device(fn):
    """Run a GppContext operator with the context's GPU as the CURRENT device.  The library enqueues on the stream it is
    handed; PyTorch's default stream is the null handle, which HIP resolves against the calling thread's current device —
    so a model on cuda:1 driven while cuda:0 is current would launch on the wrong GPU with cuda:1 pointers."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if torch.cuda.current_device() == self.index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self.index):
            return fn(self, *args, **kwargs)

    return wrapper


class GppContext:
    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise GppError("GppContext needs a cuda (HIP) device; there is no CPU fallback")
        if not torch.cuda.is_available():
            raise GppError("no GPU visible to PyTorch: the HIP path cannot run")
        self.lib = _lib.load()
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.index):
            check(self.lib.gpp_create(ctypes.byref(h), self.index), "gpp_create")
        self.h = h
        self.coop_panel = os.environ.get("GPP_COOP_PANEL", "1") != "0"  # mirrors the handle's GPP_OPT_COOP_PANEL
        self.dag_sched = os.environ.get("GPP_DAG_SCHED", "1") != "0"
        # kernels that wait for each other across launches cannot run when dispatches are serialised: they would only time out
        if any(os.environ.get(v, "0") not in ("", "0") for v in ("HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "CUDA_LAUNCH_BLOCKING")):
            self.set_option(OPT_DAG_SCHED, 0)
        self._ws: Optional[torch.Tensor] = None

    # -- plumbing --------------------------------------------------------------------------------
    def _stream(self) -> None:
        s = torch.cuda.current_stream(self.index).cuda_stream
        check(self.lib.gpp_set_stream(self.h, ctypes.c_void_p(s)), "gpp_set_stream")

    def set_option(self, option: int, value: int) -> None:
        check(self.lib.gpp_set_option(self.h, int(option), int(value)), "gpp_set_option")
        if option == OPT_COOP_PANEL:
            self.coop_panel = bool(value)
        elif option in (OPT_DAG_SCHED, OPT_EXEC_SCHED):
            self.dag_sched = bool(value)

    @_on_own_device
    def internal_streams(self):
        """(latency stream, throughput stream, unmasked stream): the handle's internal streams as torch streams — 32 CUs,
        the other 224, and one without a CU mask."""
        if getattr(self, "_istreams", None) is None:
            out = []
            for which in (0, 1, 2):
                p = ctypes.c_void_p()
                check(self.lib.gpp_internal_stream(self.h, which, ctypes.byref(p)), "gpp_internal_stream")
                out.append(torch.cuda.ExternalStream(p.value, device=self.device))
            self._istreams = tuple(out)
        return self._istreams

    @_on_own_device
    def ensure_workspace(self, op: int, N: int, M: int, D: int, S: int) -> None:
        need = int(self.lib.gpp_workspace_bytes(self.h, op, N, M, D, S))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(self.lib.gpp_set_workspace(self.h, self._ws.data_ptr(), self._ws.numel()), "gpp_set_workspace")

    def _check_groups(self, grp: torch.Tensor, N: int, S: int) -> None:
        """The kernels index tau[grp[i]] / g_tau[grp[i]] unchecked: the length is verified on every call, the value range
        once per index tensor (it costs a device read, i.e. a host sync)."""
        _need(grp, torch.int32, "grp")
        if grp.numel() != N:
            raise GppError(f"noise-group index has {grp.numel()} entries for {N} points (stale fidel_indices?)")
        # keyed on the tensor OBJECT (weak reference) and its version counter, not on its address: the caching allocator
        # hands a new index tensor of the same size the address of a freed one
        seen = getattr(self, "_grp_ok", None)
        if seen is None or seen[0]() is not grp or seen[1:] != (grp._version, N, S):
            lo, hi = int(grp.min()), int(grp.max())
            if lo < 0 or hi >= max(S, 1):
                raise GppError(f"noise-group index out of range: values in [{lo}, {hi}] for {S} noise levels")
            self._grp_ok = (weakref.ref(grp), grp._version, N, S)

    # -- operators -------------------------------------------------------------------------------
    @_on_own_device
    def kernel_build(self, U, w, sf2, tau, grp, out, *, jitter=0.0, kind=KIND_RBF, d_split=0, uplo=UPLO_FULL,
                     row0=0, nrows=None):
        N, D = U.shape
        _check_features(D)
        for t, n in ((U, "U"), (w, "w"), (sf2, "sf2"), (out, "Ky")):
            _need(t, torch.float64, n)
        if tau is not None:
            _need(tau, torch.float64, "tau")
        S = 0 if tau is None else tau.numel()
        if grp is not None:
            self._check_groups(grp, N, S)
        if not U.is_contiguous():
            raise GppError("U must be contiguous")
        self._stream()
        check(self.lib.gpp_kernel_build(self.h, U.data_ptr(), N, D, w.data_ptr(), sf2.data_ptr(), _ptr(tau), _ptr(grp), S,
                                        float(jitter), kind, d_split, uplo, out.data_ptr(), _ld(out), row0,
                                        N - row0 if nrows is None else nrows), "gpp_kernel_build")
        return out

    @_on_own_device
    def cross_kernel(self, Ua, Ub, w, sf2, out, *, kind=KIND_RBF, d_split=0):
        _check_features(Ua.shape[1])
        for t, n in ((Ua, "Ua"), (Ub, "Ub"), (w, "w"), (sf2, "sf2"), (out, "Kab")):
            _need(t, torch.float64, n)
        if not (Ua.is_contiguous() and Ub.is_contiguous()):
            raise GppError("Ua/Ub must be contiguous")
        self._stream()
        check(self.lib.gpp_cross_kernel(self.h, Ua.data_ptr(), Ua.shape[0], Ub.data_ptr(), Ub.shape[0], Ua.shape[1],
                                        w.data_ptr(), sf2.data_ptr(), kind, d_split, out.data_ptr(), _ld(out)),
              "gpp_cross_kernel")
        return out

    @_on_own_device
    def potrf(self, A, Linv, info, T=None):
        _need(A, torch.float64, "A"); _need(Linv, torch.float64, "Linv"); _need(info, torch.int32, "info")
        self._stream()
        if T is None:
            check(self.lib.gpp_potrf(self.h, A.data_ptr(), A.shape[0], _ld(A), Linv.data_ptr(), _ld(Linv), info.data_ptr()),
                  "gpp_potrf")
        else:
            check(self.lib.gpp_potrf_ws(self.h, A.data_ptr(), A.shape[0], _ld(A), Linv.data_ptr(), _ld(Linv), T.data_ptr(),
                                        _ld(T), info.data_ptr()), "gpp_potrf_ws")

    @_on_own_device
    def trtri(self, U, Linv, T):
        self._stream()
        check(self.lib.gpp_trtri(self.h, U.data_ptr(), U.shape[0], _ld(U), Linv.data_ptr(), _ld(Linv), T.data_ptr(), _ld(T)),
              "gpp_trtri")

    @_on_own_device
    def lauum(self, Linv, Kinv):
        self._stream()
        check(self.lib.gpp_lauum(self.h, Linv.data_ptr(), Linv.shape[0], _ld(Linv), Kinv.data_ptr(), _ld(Kinv)), "gpp_lauum")

    @_on_own_device
    def syrk_rows(self, Urow, C, nb, first_block, rank, nranks):
        """C(upper) -= Urow^T Urow on the block rows (height nb) of C this rank owns (block-cyclic), one launch."""
        self._stream()
        check(self.lib.gpp_syrk_rows(self.h, Urow.data_ptr(), _ld(Urow), C.data_ptr(), _ld(C), C.shape[0], Urow.shape[0], nb,
                                     first_block, rank, nranks), "gpp_syrk_rows")

    @_on_own_device
    def shard_list_begin(self, N, nb, rank, nranks, A, Kc, Lc, D, W, info, workers=0) -> bool:
        """Enqueue this rank's ticket list of the sharded factorisation + forward sweep (gpp_shard_list_begin in gpp.h).  False:
        not applicable here, nothing was enqueued."""
        self._stream()
        used = ctypes.c_int(0)
        check(self.lib.gpp_shard_list_begin(self.h, N, nb, rank, nranks, A.data_ptr(), _ld(A), Kc.data_ptr(), Lc.data_ptr(), _ld(Kc),
                                            D.data_ptr(), W[0].data_ptr(), W[1].data_ptr(), W[2].data_ptr(), _ld(W[0]),
                                            info.data_ptr(), int(workers), ctypes.byref(used)), "gpp_shard_list_begin")
        return bool(used.value)

    def shard_messages(self, N: int, nb: int, k: int):
        """Column ranges [(c0, c1), ...] of block row k's messages in the order they travel (gpp.h): the head — diagonal block and
        the next block's columns —, then the tail in pieces of ``gpp_shard_piece_cols()`` columns.  The index in the list is the
        ``tail`` argument of ``shard_list_gate`` / ``shard_list_signal``."""
        o, o2 = k * nb, min((k + 2) * nb, N)
        W = int(self.lib.gpp_shard_piece_cols()) or N
        return [(o, o2)] + [(c, min(c + W, N)) for c in range(o2, N, W)]

    @_on_own_device
    def shard_list_gate(self, stream, tail: int, k: int) -> None:
        check(self.lib.gpp_shard_list_gate(self.h, ctypes.c_void_p(stream.cuda_stream), int(tail), k), "gpp_shard_list_gate")

    @_on_own_device
    def shard_list_signal(self, stream, tail: int, k: int) -> None:
        check(self.lib.gpp_shard_list_signal(self.h, ctypes.c_void_p(stream.cuda_stream), int(tail), k), "gpp_shard_list_signal")

    @_on_own_device
    def shard_list_end(self) -> None:
        self._stream()
        check(self.lib.gpp_shard_list_end(self.h), "gpp_shard_list_end")

    @_on_own_device
    def shard_back_list(self, N, nb, rank, nranks, A, Kc, Lc, D, info, workers=0) -> bool:
        """The sharded back-substitution of this rank as one ticket list on the current stream (gpp_shard_back_list in gpp.h).
        False: not applicable here, nothing was enqueued."""
        self._stream()
        used = ctypes.c_int(0)
        check(self.lib.gpp_shard_back_list(self.h, N, nb, rank, nranks, A.data_ptr(), _ld(A), Kc.data_ptr(), Lc.data_ptr(), _ld(Kc),
                                           D.data_ptr(), info.data_ptr(), int(workers), ctypes.byref(used)), "gpp_shard_back_list")
        return bool(used.value)

    @_on_own_device
    def gemm_lower_cols(self, A, B, C, alpha, beta, nb, first_block, rank, nranks, row0=0, row1=None, compact=False):
        """C(lower, owned column blocks of width nb) = beta C + alpha A^T B;  A, B: K x M row-contiguous, C: M x M; rows
        [row0, row1) of C only.  ``compact``: B (K rows) and C (M rows) hold only the owned column blocks, side by side."""
        M, K = A.shape[1], A.shape[0]
        if compact:
            if B.shape[0] != K or C.shape[0] != M:
                raise GppError("gemm_lower_cols: shapes do not match")
        elif A.shape != B.shape or C.shape[0] != C.shape[1] or C.shape[0] != M:
            raise GppError("gemm_lower_cols: shapes do not match")
        self._stream()
        check(self.lib.gpp_gemm_lower_cols(self.h, A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), C.data_ptr(), _ld(C), M,
                                           K, float(alpha), float(beta), nb, first_block, rank, nranks, row0,
                                           M if row1 is None else row1, 1 if compact else 0),
              "gpp_gemm_lower_cols")

    @_on_own_device
    def trmv_lower_cols(self, T, x, y, nb, rank, nranks, trans=False, compact=False):
        """y = (owned column blocks of lower T) x, or their transpose times x on the owned entries (0 elsewhere).
        ``compact``: T (N rows) holds only the owned column blocks, side by side."""
        for t, n in ((x, "x"), (y, "y")):
            _need(t, torch.float64, n)
        if trans:
            self.ensure_workspace(OP_MLL_EVAL, T.shape[0], 0, 1, 1)
        self._stream()
        check(self.lib.gpp_trmv_lower_cols(self.h, T.data_ptr(), _ld(T), T.shape[0], x.data_ptr(), y.data_ptr(), nb, rank, nranks,
                                           1 if trans else 0, 1 if compact else 0), "gpp_trmv_lower_cols")

    @_on_own_device
    def mll_scalars(self, U, z, out3):
        self._stream()
        check(self.lib.gpp_mll_scalars(self.h, U.data_ptr(), _ld(U), U.shape[0], z.data_ptr(), out3.data_ptr()), "gpp_mll_scalars")

    @_on_own_device
    def lauum_rows(self, Linv, Kinv, rank, nranks):
        """This rank's cyclic share (128-row tile rows) of Kinv = Linv^T Linv, one launch."""
        self._stream()
        check(self.lib.gpp_lauum_rows(self.h, Linv.data_ptr(), Linv.shape[0], _ld(Linv), Kinv.data_ptr(), _ld(Kinv), rank, nranks),
              "gpp_lauum_rows")

    @_on_own_device
    def lauum_rows_range(self, Linv, Kinv, rank, nranks, row0, row1):
        """The part of this rank's cyclic share of Kinv = Linv^T Linv inside rows [row0, row1) (needs the column blocks of
        Linv up to row1 only)."""
        self._stream()
        check(self.lib.gpp_lauum_rows_range(self.h, Linv.data_ptr(), Linv.shape[0], _ld(Linv), Kinv.data_ptr(), _ld(Kinv), rank,
                                            nranks, row0, row1), "gpp_lauum_rows_range")

    @_on_own_device
    def transpose(self, src, dst):
        """dst = src^T, out of place (2-D views with unit column stride)."""
        if src.shape[0] != dst.shape[1] or src.shape[1] != dst.shape[0]:
            raise GppError("transpose: shapes do not match")
        self._stream()
        check(self.lib.gpp_transpose(self.h, src.data_ptr(), src.stride(0), src.shape[0], src.shape[1], dst.data_ptr(), dst.stride(0)),
              "gpp_transpose")

    @_on_own_device
    def mll_reduce(self, L, Linv, r, z, out3):
        for t, n in ((r, "r"), (z, "z"), (out3, "out3")):
            _need(t, torch.float64, n)
        self._stream()
        check(self.lib.gpp_mll_reduce(self.h, L.data_ptr(), _ld(L), Linv.data_ptr(), _ld(Linv), L.shape[0], r.data_ptr(),
                                      z.data_ptr(), out3.data_ptr()), "gpp_mll_reduce")

    @_on_own_device
    def alpha(self, Linv, z, alpha):
        N = Linv.shape[0]
        self._stream()
        check(self.lib.gpp_alpha(self.h, Linv.data_ptr(), _ld(Linv), N, z.data_ptr(), alpha.data_ptr()), "gpp_alpha")

    @_on_own_device
    def grad_reduce(self, U, w, sf2, grp, S, alpha, Kinv, dU, g_w, g_sf2, g_tau, g_U, *, kind=KIND_RBF, d_split=0):
        N, D = U.shape
        if grp is not None:
            self._check_groups(grp, N, S)
        self.ensure_workspace(OP_MLL_EVAL, N, 0, D, S)
        self._stream()
        check(self.lib.gpp_grad_reduce(self.h, U.data_ptr(), N, D, w.data_ptr(), sf2.data_ptr(), _ptr(grp), S, kind,
                                       d_split, alpha.data_ptr(), Kinv.data_ptr(), _ld(Kinv), dU, g_w.data_ptr(),
                                       g_sf2.data_ptr(), g_tau.data_ptr(), _ptr(g_U)), "gpp_grad_reduce")

    @_on_own_device
    def grad_reduce_rows(self, U, w, sf2, grp, S, alpha, Kinv, dU, nb, rank, nranks, g_w, g_sf2, g_tau, g_U, *,
                         kind=KIND_RBF, d_split=0):
        """Partial sums over the block rows of Kinv owned by ``rank`` (block-cyclic, block height ``nb``)."""
        N, D = U.shape
        if grp is not None:
            self._check_groups(grp, N, S)
        self.ensure_workspace(OP_MLL_EVAL, N, 0, D, S)
        self._stream()
        check(self.lib.gpp_grad_reduce_rows(self.h, U.data_ptr(), N, D, w.data_ptr(), sf2.data_ptr(), _ptr(grp), S, kind,
                                            d_split, alpha.data_ptr(), Kinv.data_ptr(), _ld(Kinv), dU, nb, rank, nranks,
                                            g_w.data_ptr(), g_sf2.data_ptr(), g_tau.data_ptr(), _ptr(g_U)),
              "gpp_grad_reduce_rows")

    @_on_own_device
    def grad_reduce_cols(self, U, w, sf2, grp, S, alpha, Kinv, dU, nb, rank, nranks, g_w, g_sf2, g_tau, g_U, *,
                         kind=KIND_RBF, d_split=0, compact=False):
        """Partial sums over the COLUMN blocks of Kinv's lower triangle owned by ``rank`` (block-cyclic, width ``nb``).
        ``compact``: Kinv (N rows) holds only the owned column blocks, side by side."""
        N, D = U.shape
        if grp is not None:
            self._check_groups(grp, N, S)
        self.ensure_workspace(OP_MLL_EVAL, N, 0, D, S)
        self._stream()
        check(self.lib.gpp_grad_reduce_cols(self.h, U.data_ptr(), N, D, w.data_ptr(), sf2.data_ptr(), _ptr(grp), S, kind,
                                            d_split, alpha.data_ptr(), Kinv.data_ptr(), _ld(Kinv), dU, nb, rank, nranks,
                                            g_w.data_ptr(), g_sf2.data_ptr(), g_tau.data_ptr(), _ptr(g_U), 1 if compact else 0),
              "gpp_grad_reduce_cols")

    @_on_own_device
    def predict(self, Linv, alpha, Ksn, kss, V, mean_out, var_out):
        self._stream()
        check(self.lib.gpp_predict(self.h, Linv.data_ptr(), _ld(Linv), Linv.shape[0], alpha.data_ptr(), Ksn.data_ptr(),
                                   _ld(Ksn), Ksn.shape[0], _ptr(kss), _ptr(V), 0 if V is None else _ld(V),
                                   mean_out.data_ptr(), _ptr(var_out)), "gpp_predict")

    @_on_own_device
    def predict_tn(self, Linv, z, Kns, kss, V, mean_out, var_out):
        """Prediction from the transposed cross block Kns (N x M) and z = Linv r: mean, variance and V = Kns^T Linv^T."""
        self._stream()
        check(self.lib.gpp_predict_tn(self.h, Linv.data_ptr(), _ld(Linv), Linv.shape[0], z.data_ptr(), Kns.data_ptr(), _ld(Kns),
                                      Kns.shape[1], kss.data_ptr(), V.data_ptr(), _ld(V), mean_out.data_ptr(), var_out.data_ptr()),
              "gpp_predict_tn")

    @_on_own_device
    def gemm(self, transA, transB, M, N, K, alpha, A, B, beta, C, *, a_mask=0, b_mask=0, klo_mode=0, khi_mode=0,
             c_tri=0):
        self._stream()
        check(self.lib.gpp_gemm(self.h, transA, transB, M, N, K, float(alpha), A.data_ptr(), _ld(A), B.data_ptr(), _ld(B),
                                float(beta), C.data_ptr(), _ld(C), a_mask, b_mask, klo_mode, khi_mode, c_tri), "gpp_gemm")

    @_on_own_device
    def gemm_batched(self, transA, transB, M, N, K, alpha, A, sA, B, sB, beta, C, sC, batch, *, a_mask=0, b_mask=0,
                     klo_mode=0, khi_mode=0, c_tri=0):
        """``batch`` products of one shape; A/B/C are the first elements' views, sA/sB/sC element strides between them."""
        self._stream()
        check(self.lib.gpp_gemm_batched(self.h, transA, transB, M, N, K, float(alpha), A.data_ptr(), _ld(A), sA, B.data_ptr(),
                                        _ld(B), sB, float(beta), C.data_ptr(), _ld(C), sC, batch, a_mask, b_mask, klo_mode,
                                        khi_mode, c_tri), "gpp_gemm_batched")

    # -- batched evaluation: tensors carry a leading batch dimension -------------------------------------------
    # matrices: (B, N, ld) views of a (B, N, ld) allocation ([:, :, :N]); vectors: (B, N) views of a (B, sv) allocation
    # with sv even (``batched_vector``); parameters (B, D), (B,), (B, S)
    def batched_buffer(self, B: int, n: int) -> torch.Tensor:
        ld = max(16, (n + 15) // 16 * 16)
        return torch.empty((B, n, ld), dtype=torch.float64, device=self.device)[:, :, :n]

    def batched_vector(self, B: int, n: int) -> torch.Tensor:
        return torch.empty((B, n + (n & 1)), dtype=torch.float64, device=self.device)[:, :n]

    @_on_own_device
    def ensure_workspace_batched(self, B, N, D, S):
        need = B * int(self.lib.gpp_workspace_bytes(self.h, OP_MLL_EVAL, N, 0, D, S))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            check(self.lib.gpp_set_workspace(self.h, self._ws.data_ptr(), self._ws.numel()), "gpp_set_workspace")

    @_on_own_device
    def kernel_build_batched(self, U, w, sf2, tau, grp, out, *, jitter=0.0, kind=KIND_RBF, d_split=0, uplo=UPLO_FULL):
        """U: (N, D) shared or (B, N, D); w: (B, D); sf2: (B,); tau: (B, S) or None; out: (B, N, ld) view."""
        B, N = out.shape[0], out.shape[1]
        D = w.shape[1]
        sU = 0 if U.dim() == 2 else U.stride(0)
        S = 0 if tau is None else tau.shape[1]
        self._stream()
        check(self.lib.gpp_kernel_build_batched(self.h, U.data_ptr(), sU, N, D, w.data_ptr(), sf2.data_ptr(), _ptr(tau),
                                                _ptr(grp), S, float(jitter), kind, d_split, uplo, out.data_ptr(), out.stride(1),
                                                out.stride(0), B), "gpp_kernel_build_batched")

    @_on_own_device
    def potrf_batched(self, A, Linv, info):
        self._stream()
        check(self.lib.gpp_potrf_batched(self.h, A.data_ptr(), A.shape[1], A.stride(1), A.stride(0), Linv.data_ptr(),
                                         Linv.stride(1), Linv.stride(0), info.data_ptr(), A.shape[0]), "gpp_potrf_batched")

    @_on_own_device
    def trtri_batched(self, U, Linv, T):
        self._stream()
        check(self.lib.gpp_trtri_batched(self.h, U.data_ptr(), U.shape[1], U.stride(1), U.stride(0), Linv.data_ptr(),
                                         Linv.stride(1), Linv.stride(0), T.data_ptr(), T.stride(1), T.stride(0), U.shape[0]),
              "gpp_trtri_batched")

    @_on_own_device
    def lauum_batched(self, Linv, Kinv):
        self._stream()
        check(self.lib.gpp_lauum_batched(self.h, Linv.data_ptr(), Linv.shape[1], Linv.stride(1), Linv.stride(0), Kinv.data_ptr(),
                                         Kinv.stride(1), Kinv.stride(0), Linv.shape[0]), "gpp_lauum_batched")

    @_on_own_device
    def mll_reduce_batched(self, L, Linv, r, z, out3):
        self._stream()
        check(self.lib.gpp_mll_reduce_batched(self.h, L.data_ptr(), L.stride(1), L.stride(0), Linv.data_ptr(), Linv.stride(1),
                                              Linv.stride(0), L.shape[1], r.data_ptr(), z.data_ptr(), self._sv(r, z), out3.data_ptr(),
                                              L.shape[0]), "gpp_mll_reduce_batched")

    @staticmethod
    def _sv(*vecs):
        sv = vecs[0].stride(0)
        if any(v.stride(0) != sv or v.stride(1) != 1 for v in vecs) or (sv & 1):
            raise GppError("batched vectors must share one even row stride (use GppContext.batched_vector)")
        return sv

    @_on_own_device
    def alpha_batched(self, Linv, z, alpha):
        self._stream()
        check(self.lib.gpp_alpha_batched(self.h, Linv.data_ptr(), Linv.stride(1), Linv.stride(0), Linv.shape[1], z.data_ptr(),
                                         alpha.data_ptr(), self._sv(z, alpha), Linv.shape[0]), "gpp_alpha_batched")

    @_on_own_device
    def grad_reduce_batched(self, U, w, sf2, grp, S, alpha, Kinv, dU, g_w, g_sf2, g_tau, g_U, *, kind=KIND_RBF, d_split=0):
        B, N = Kinv.shape[0], Kinv.shape[1]
        D = w.shape[1]
        sU = 0 if U.dim() == 2 else U.stride(0)
        self.ensure_workspace_batched(B, N, D, S)
        self._stream()
        check(self.lib.gpp_grad_reduce_batched(self.h, U.data_ptr(), sU, N, D, w.data_ptr(), sf2.data_ptr(), _ptr(grp), S, kind,
                                               d_split, alpha.data_ptr(), self._sv(alpha), Kinv.data_ptr(), Kinv.stride(1),
                                               Kinv.stride(0), dU,
                                               g_w.data_ptr(), g_sf2.data_ptr(), g_tau.data_ptr(), _ptr(g_U), B),
              "gpp_grad_reduce_batched")


def get_context(device) -> GppContext:
    device = torch.device(device)
    if device.type != "cuda":
        raise GppError(
            f"the exact-GP hot path runs only on an MI355X through libgpp_hip (device={device}); no CPU fallback exists")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, threading.get_ident())
    ctx = _contexts.get(key)
    if ctx is None:
        with _contexts_lock:
            ctx = GppContext(torch.device("cuda", idx))
            _contexts[key] = ctx
    return ctx


@atexit.register
def _destroy_contexts() -> None:
    """Release the library handles (and their internal CU-masked streams / events) while the HIP runtime is still
    alive; leaving them to process teardown crashes under rocprofv3."""
    if not _contexts:
        return
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
    for key, ctx in list(_contexts.items()):
        try:
            ctx.lib.gpp_destroy(ctx.h)
        except Exception:
            pass
        _contexts.pop(key, None)
