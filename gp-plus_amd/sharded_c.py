"""The sharded evaluation through the C driver ``gpp_shard_eval`` (include/gpp.h; gp-plus_amd/csrc/gpp_shard.hip) instead of the Python
choreography of ``sharded.py``: ONE call per rank does build, ticket lists with the block rows' messages, z / alpha,
back-substitution and the gradient reduction.  The collectives are either callbacks into ``torch.distributed`` (any backend; what a
C / MPI caller would implement with its own library) or RCCL opened by the library itself (``rccl=True``: rank 0's unique id travels
over the process group once).  Same contract as ``sharded.sharded_mll``'s forward + backward for the caller who wants the numbers:
returns (mll, alpha, g_w, g_sf2, g_tau, g_U) — identical on every rank.  Reference counterpart: `mll(output, y)` +
`loss.backward()`, optim/mll_torch.py:114-117 (the reference has no multi-GPU evaluation)."""
from __future__ import annotations

import ctypes
from ctypes import CFUNCTYPE, Structure, c_int, c_int64, c_size_t, c_void_p
import torch
import torch.distributed as dist

from .backend import OP_MLL_EVAL, GppContext, check, check_status, get_context
from .errors import NotPSDError
from . import settings

__all__ = ["sharded_eval_c", "GPP_SHARD_UNSUPPORTED"]

GPP_SHARD_UNSUPPORTED = 2000
_BCAST = CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
_ALLREDUCE = CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)


class _Comm(Structure):
    _fields_ = [("user", c_void_p), ("bcast", _BCAST), ("allreduce", _ALLREDUCE)]


class _Buffers(Structure):
    _fields_ = [("A", c_void_p), ("ld", c_int64), ("Kc", c_void_p), ("Lc", c_void_p), ("ldc", c_int64), ("D", c_void_p),
                ("W0", c_void_p), ("W1", c_void_p), ("W2", c_void_p), ("ldw", c_int64), ("msg", c_void_p), ("z", c_void_p),
                ("alpha", c_void_p), ("r", c_void_p), ("flat", c_void_p), ("out3", c_void_p), ("info", c_void_p)]


def _device_view(ptr: int, nbytes: int, dev: torch.device) -> torch.Tensor:
    """A uint8 tensor over device memory the library handed to a callback."""

    class _Mem:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Mem(), device=dev)


def _stream_of(ptr, dev: torch.device):
    """The torch stream for a hipStream_t handed over by the library (NULL: the legacy default stream)."""
    return torch.cuda.ExternalStream(int(ptr), device=dev) if ptr else torch.cuda.default_stream(dev)


def _torch_callbacks(group, dev: torch.device):
    """bcast / allreduce over ``torch.distributed`` on the stream the library names (host-staged unless the backend moves device
    memory itself)."""
    direct = dist.get_backend(group) == "nccl"
    rank = dist.get_rank(group)
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))

    def bcast(_user, buf, nbytes, root, stream):
        try:
            with torch.cuda.stream(_stream_of(stream, dev)):
                t = _device_view(buf, nbytes, dev)
                if direct:
                    dist.broadcast(t, glob(root), group=group)
                else:
                    h = t.cpu() if rank == root else torch.empty(nbytes, dtype=torch.uint8)
                    dist.broadcast(h, glob(root), group=group)
                    if rank != root:
                        t.copy_(h)
            return 0
        except Exception:  # noqa: BLE001  (an exception must not unwind through the C frames)
            import traceback
            traceback.print_exc()
            return 1

    def allreduce(_user, buf, count, kind, stream):
        try:
            with torch.cuda.stream(_stream_of(stream, dev)):
                raw = _device_view(buf, count * (4 if kind == 1 else 8), dev)
                t = raw.view(torch.int32 if kind == 1 else torch.float64)
                op = dist.ReduceOp.MAX if kind == 1 else dist.ReduceOp.SUM
                if direct:
                    dist.all_reduce(t, op=op, group=group)
                else:
                    h = t.cpu()
                    dist.all_reduce(h, op=op, group=group)
                    t.copy_(h)
            return 0
        except Exception:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return 1

    return _BCAST(bcast), _ALLREDUCE(allreduce)


_state = {}


def sharded_eval_c(U, w, sf2, tau, mean, y, grp=None, kind: int = 0, d_split: int = 0, n_grad_dims: int = 0, group=None, nb: int = 1024,
                   need_grad: bool = True, rccl: bool = False):
    """(mll, alpha, g_w, g_sf2, g_tau, g_U) of log N(y | mean, sf2 k(U, U; w) + diag(tau[grp])), every rank of ``group`` calling with
    identical arguments; gpytorch's jitter schedule around the C call.  Raises ``NotImplementedError`` where the lists do not apply
    (the caller then uses ``sharded.sharded_mll``)."""
    dev = U.device
    ctx: GppContext = get_context(dev)
    lib = ctx.lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    N, D = U.shape
    f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64).contiguous()  # noqa: E731
    Ud, wd, sd, td = f64(U), f64(w), f64(sf2).reshape(1), f64(tau).reshape(-1)
    S = td.numel()
    if grp is not None and grp.dtype != torch.int32:
        grp = grp.to(torch.int32)
    # The handle has ONE communicator slot (gpp_set_comm / gpp_comm_init_rccl overwrite it): the key names the group by its member
    # ranks (``id(group)`` can be reused by a later group object) and ``_state["installed", device]`` says which key the handle
    # currently carries — another group or backend re-installs before the call instead of running with the previous one's collectives.
    members = tuple(range(world)) if group is None else tuple(dist.get_process_group_ranks(group))
    key = (ctx.index, members, bool(rccl))
    if _state.get(("installed", ctx.index)) != key:
        if rccl:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (ctypes.c_uint8 * 128)()
                check(lib.gpp_comm_unique_id(raw), "gpp_comm_unique_id")
                uid = torch.tensor(list(raw), dtype=torch.uint8)
            if world > 1:
                if dist.get_backend(group) == "nccl":
                    u = uid.to(dev); dist.broadcast(u, 0 if group is None else dist.get_global_rank(group, 0), group=group); uid = u.cpu()
                else:
                    dist.broadcast(uid, 0 if group is None else dist.get_global_rank(group, 0), group=group)
            raw = (ctypes.c_uint8 * 128)(*uid.tolist())
            check(lib.gpp_comm_init_rccl(ctx.h, raw, rank, world), "gpp_comm_init_rccl")
            _state[key] = ("rccl",)
        else:
            cb = _torch_callbacks(group, dev)
            comm = _Comm(None, cb[0], cb[1])
            check(lib.gpp_set_comm(ctx.h, ctypes.byref(comm), rank, world), "gpp_set_comm")
            _state[key] = (cb, comm)  # (kept alive: the library calls them)
        _state[("installed", ctx.index)] = key
    n = lambda which: int(lib.gpp_shard_buffer_doubles(N, nb, rank, world, which))  # noqa: E731
    ld = (N + 15) // 16 * 16
    nblk = -(-N // nb)
    wc = max(len(range(rank, nblk, world)), 1) * nb
    mk = lambda k: torch.empty(k, dtype=torch.float64, device=dev)  # noqa: E731
    bufs = _state.get(("bufs", key, N, nb))
    if bufs is None:
        for k in [k for k in _state if k and k[0] == "bufs"]:
            del _state[k]
        bufs = dict(A=mk(n(0)), Kc=mk(n(1)), Lc=mk(n(1)), D=mk(n(2)), W0=mk(n(3)), W1=mk(n(3)), W2=mk(n(3)), msg=mk(n(4)),
                    z=mk(N), alpha=mk(N), r=mk(N), out3=mk(3), info=torch.zeros(2, dtype=torch.int32, device=dev))
        _state[("bufs", key, N, nb)] = bufs
    dU = int(n_grad_dims) if need_grad else 0
    flat = torch.zeros(D + 1 + S + N * dU, dtype=torch.float64, device=dev)
    torch.sub(f64(y), f64(mean), out=bufs["r"])
    b = _Buffers(bufs["A"].data_ptr(), ld, bufs["Kc"].data_ptr(), bufs["Lc"].data_ptr(), wc, bufs["D"].data_ptr(), bufs["W0"].data_ptr(),
                 bufs["W1"].data_ptr(), bufs["W2"].data_ptr(), ld, bufs["msg"].data_ptr(), bufs["z"].data_ptr(), bufs["alpha"].data_ptr(),
                 bufs["r"].data_ptr(), flat.data_ptr(), bufs["out3"].data_ptr(), bufs["info"].data_ptr())
    ctx.ensure_workspace(OP_MLL_EVAL, N, 0, D, S)
    ctx._stream()
    info = c_int(0)
    jitters = [0.0] + [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
    for jit in jitters:
        rc = lib.gpp_shard_eval(ctx.h, N, nb, Ud.data_ptr(), D, wd.data_ptr(), sd.data_ptr(), td.data_ptr(),
                                None if grp is None else grp.data_ptr(), S, kind, d_split, float(jit), dU, 1 if need_grad else 0,
                                ctypes.byref(b), ctypes.byref(info))
        if rc == GPP_SHARD_UNSUPPORTED:
            raise NotImplementedError("gpp_shard_eval: the ticket lists do not apply to this size / block height")
        check(rc, "gpp_shard_eval")
        if info.value == 0:
            break
        check_status(info.value)  # (a time-out: raises)
    else:
        raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitters[-1]:.1e}.")
    mll = bufs["out3"][2].clone()
    if not need_grad:
        return mll, None, None, None, None, None
    g_U = flat[D + 1 + S:].view(N, dU).clone() if dU > 0 else None
    return mll, bufs["alpha"].clone(), flat[:D].clone(), flat[D:D + 1].clone(), flat[D + 1:D + 1 + S].clone(), g_U
