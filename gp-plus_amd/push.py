"""The sharded evaluation's messages by a direct one-to-all PUSH instead of a broadcast collective (round 6; SURVEY.md:204, :423-425;
protocol: ``csrc/gpp_push.hip``, C ABI: ``gpp_push_*`` in ``include/gpp.h``).  Opt-in: ``GPP_SHARD_PUSH=1``.

The owner of a block row copies it straight from the factor (strided, no packing) into a slot of every other rank over one stream
per peer — all links at once, one hop, no collective kernel — and raises a flag there; a receiver waits for its flag, copies the
parts into place and acknowledges.  The reference has no multi-GPU evaluation (its only parallelism: joblib multistart,
optim/mll_scipy.py:287-293); this replaces ``dist.broadcast`` in ``sharded.py`` for the messages of the factorisation only — the two
small all-reduces per evaluation stay collectives.

Unmeasurable on a one-GPU box: the tests run the ranks on one device (same-device IPC), which exercises the protocol, not the links.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char, c_int, c_int64, c_void_p
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from ._lib import check
from .backend import INFO_EXEC_TIMEOUT, GppContext

HANDLE_BYTES = 128
#: GPP_SHARD_PUSH=1 selects the push transport for the block rows' messages (every rank must set it alike)
ENABLED = os.environ.get("GPP_SHARD_PUSH", "0") not in ("", "0")
#: messages moved by push in this process (tests, bench.py's ``transport``)
MESSAGES = 0

Part = Tuple[torch.Tensor, int, int]  # (2-D view with unit column stride, offset in the slot, row pitch in the slot) — in elements


def _arrays(parts: Sequence[Part]):
    n = len(parts)
    ptr, vp, off, sp, wd, ht = (c_void_p * n)(), (c_int64 * n)(), (c_int64 * n)(), (c_int64 * n)(), (c_int64 * n)(), (c_int64 * n)()
    for i, (v, o, pitch) in enumerate(parts):
        if v.dim() != 2 or (v.shape[1] > 1 and v.stride(1) != 1) or v.dtype != torch.float64:
            raise ValueError("push: a part must be a 2-D float64 view with unit column stride")
        rows, cols = v.shape
        es = v.element_size()
        ptr[i] = v.data_ptr()
        vp[i] = max(v.stride(0) if rows > 1 else cols, cols) * es
        off[i], sp[i], wd[i], ht[i] = o * es, max(pitch, cols) * es, cols * es, rows
    return n, ptr, vp, off, sp, wd, ht


class PushChannel:
    """Two slots + a flag page on this rank, every peer's mapped (collective constructor: every rank of ``group`` calls it)."""

    def __init__(self, ctx: GppContext, rank: int, world: int, group, slot_bytes: int):
        self.lib, self.rank, self.world, self.group = ctx.lib, rank, world, group
        h = c_void_p()
        rec = (c_char * HANDLE_BYTES)()
        check(self.lib.gpp_push_create(ctx.index, rank, world, slot_bytes, ctypes.byref(h), rec), "gpp_push_create")
        self.h = h
        recs: List[Optional[bytes]] = [None] * world
        dist.all_gather_object(recs, bytes(rec), group=group)
        blob = b"".join(recs)  # type: ignore[arg-type]
        check(self.lib.gpp_push_connect(self.h, blob), "gpp_push_connect")
        sb, kind = c_int64(), c_int()
        check(self.lib.gpp_push_info(self.h, ctypes.byref(sb), ctypes.byref(kind)), "gpp_push_info")
        self.slot_bytes, self.flag_kind = int(sb.value), ("uncached", "fine-grained", "ordinary")[kind.value]
        self.seq = 0

    def advance(self) -> int:
        """The number of the next message — every rank calls this once per message, in the same order."""
        self.seq += 1
        return self.seq

    def send(self, stream: torch.cuda.Stream, seq: int, parts: Sequence[Part], status: torch.Tensor) -> None:
        global MESSAGES
        n, ptr, vp, off, sp, wd, ht = _arrays(parts)
        check(self.lib.gpp_push_send(self.h, stream.cuda_stream, seq, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht),
              "gpp_push_send")
        MESSAGES += 1

    def recv(self, stream: torch.cuda.Stream, seq: int, parts: Sequence[Part], status: torch.Tensor) -> None:
        global MESSAGES
        n, ptr, vp, off, sp, wd, ht = _arrays(parts)
        check(self.lib.gpp_push_recv(self.h, stream.cuda_stream, seq, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht),
              "gpp_push_recv")
        MESSAGES += 1

    def ack(self, stream: torch.cuda.Stream, seq: int) -> None:
        check(self.lib.gpp_push_ack(self.h, stream.cuda_stream, seq), "gpp_push_ack")

    def close(self) -> None:
        """Collective: unmap the peers' memory everywhere, then free the own."""
        if self.h is None:
            return
        check(self.lib.gpp_push_destroy(self.h, 0), "gpp_push_destroy")
        dist.barrier(group=self.group)
        check(self.lib.gpp_push_destroy(self.h, 1), "gpp_push_destroy")
        self.h = None


_channels = {}
_disabled = False


def disable() -> None:
    """After a time-out anywhere the message numbers may no longer agree between the ranks: the rest of this process broadcasts.
    (Called on a status every rank has seen — the MAX over the ranks — so every rank switches at the same evaluation.)"""
    global _disabled
    _disabled = True


def active(world: int) -> bool:
    return ENABLED and not _disabled and world > 1


def channel(ctx: GppContext, rank: int, world: int, group, slot_bytes: int) -> Optional[PushChannel]:
    """The channel of (device, group), created — or replaced by a larger one — collectively; None when the transport is off."""
    if not active(world):
        return None
    key = (ctx.index, rank, world, id(group) if group is not None else None)
    ch = _channels.get(key)
    if ch is not None and ch.slot_bytes < slot_bytes:
        ch.close()
        ch = None
    if ch is None:
        ch = PushChannel(ctx, rank, world, group, slot_bytes)
        _channels[key] = ch
    return ch
