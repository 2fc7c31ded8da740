"""Sharded single evaluation of the exact-GP marginal likelihood over the GPUs of one node (SURVEY.md §8(e) mode 2,
BASELINE config 5: N = 60 000 on 8 x MI355X).  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI); every rank calls :func:`sharded_mll` with identical arguments and receives the identical value and gradients.

The reference has no multi-GPU evaluation (its only parallelism is joblib over restarts, optim/mll_scipy.py:287-293);
this is the same computation as ``linalg.ExactMLLFunction`` (reference call sites optim/mll_torch.py:114-117) with its
O(N^3) stages split 1-D block-cyclically (block height / width ``nb``, owner of block k = k mod P):

  build    every rank builds the block ROWS of Ky it owns                                      (no communication)
  potrf    right-looking over block rows of the upper-stored matrix: the owner of block row k factors and inverts the
           diagonal block on the 32-CU panel stream, solves the block row with one GEMM on the throughput stream and
           BROADCASTS the finished row slab (packed: nb x (N - o) doubles) and the diagonal block's inverse; every rank
           then updates the block rows it owns.  Inside each rank's share the update is split like the single-GPU
           driver's (gpp_potrf_ws): the rows the next two steps depend on first, then — on the rank whose next diagonal
           block is being factored meanwhile — a trapezoid of early rows on the CU-masked stream beside that panel and the
           bulk on the stream WITHOUT a CU mask once the panel is done.  As each row slab becomes final its mirror
           (L = U^T) is written into the unused strict lower triangle of the same buffer.
  forward  column blocks of L^-1 are independent forward substitutions against the replicated factor: the owner of column
           block c sweeps  Y_j = -L_jj^-1 sum_{c<=k<j} L_jk Y_k  right-looking, all its column blocks in ONE batched launch
           per step and product.
  z, alpha z = L^-1 r and alpha = L^-T z from the owned column blocks (gpp_trmv_lower_cols) + two all-reduces of N doubles.
  backward each rank turns ITS column blocks of L^-1 into the same column blocks of Ky^-1 = L^-T L^-1 by BACK-substitution
           against the replicated factor ("each GPU solves for its own block-columns of L^-T L^-1 E_k —
           block-parallel, no further comm"): Z_j = L_jj^-T (Y_j - sum_{k>j} U_jk Z_k), right-looking from the last block
           row up, only the rows at and below each column block's diagonal (Ky^-1 is symmetric), one TN GEMM per step
           over the lower-triangular tiles of the owned column blocks (gpp_gemm_lower_cols; the factor's mirror makes the
           product row-contiguous).  Same flops as the LAUUM share it replaces (sum_c (N - c)^2 nb = N^3 / 3 over P),
           nothing on the wire.
  grad     ``gpp_grad_reduce_cols`` over the owned column blocks, then ONE all-reduce of D + 1 + S (+ N dU) doubles.

Communication per evaluation and GPU: the packed factor slabs (4 N^2 B received) + the diagonal-block inverses (8 N nb B)
+ three small all-reduces — 14.4 GB at C5 for every P; round 2 also moved the inverse's column blocks (another 4 N^2 B).
Memory per GPU (round 4): ONE N x N buffer — the replicated factor (upper) and its mirror (strict lower), which the sweeps read —
plus what the rank OWNS of the other two: its column blocks of L^-1 and of Ky^-1 stored side by side (N x N/P each), the diagonal
blocks' inverses (N x nb) and an nb x N row of scratch: 28.8 + 2 x 3.6 + 1 GB = 37 GB at C5 on 8 GPUs where three full matrices took
86 GB on every rank ("28.8 GB (3.6 GB/GPU sharded)") — what grows with N on a rank is the factor alone.

Round 5: where they apply (N >= 4096, block height nb = 1024 or another the cooperative panel takes, a last block of more than 256
rows) the factorisation + forward sweep and the back-substitution of a rank run as TICKET LISTS of the DAG executor
(``_factor_list``, ``gpp_shard_back_list``; csrc/gpp_dag.hip DAG_SHARD / DAG_BACK): the launches described above become tile tasks
in dependency order taken by persistent work-groups, the block rows of other ranks arrive as the same two messages and raise one
counter each, and the factor's mirror is written beside the list.  ``GPP_SHARD_LIST=0`` keeps the launch-per-product choreography,
which is also the fall-back (other sizes, a time-out).  One rank: 136 ms against 168 at C2, 3.4 s against 3.5 at C5.  The whole
evaluation is also available as ONE C call per rank (``gpp_shard_eval``; ``sharded_c.py``).

Round 6: the tail of a block row travels in pieces (``gpp_shard_piece_cols``), and ``GPP_SHARD_PUSH=1`` replaces the broadcasts of
the block rows by direct one-to-all pushes through hipIpc-mapped slots (push.py; SURVEY.md:204) — the same messages, bit for bit.

Without RCCL (tests: several processes sharing one GPU over "gloo") the collectives are staged through host memory.
``GPP_SHARDED_FORCE_COLLECTIVES=1`` issues every collective (and the packing around it) even in a group of one rank, so that
the RCCL branch runs on a single GPU (tests/test_gpu_sharded.py).
"""
from __future__ import annotations

import os
import warnings
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .backend import INFO_PANEL_TIMEOUT, KIND_RBF, UPLO_FULL, GppContext, get_context, panel_timed_out, square_buffer
from . import push as _push
from .errors import NanError, NotPSDError
from . import settings

__all__ = ["ShardedWorkspace", "sharded_mll", "ShardedMLLFunction"]


class _Comm:
    def __init__(self, group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("sharded evaluation needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.direct = dist.get_backend(group) == "nccl"  # RCCL moves device memory itself
        # data travels (and is packed / unpacked around the collectives) when there is more than one rank — or always, on
        # request: the RCCL calls, their stream ordering and the buffer reuse then execute in a group of ONE rank too
        self.travel = self.world > 1 or os.environ.get("GPP_SHARDED_FORCE_COLLECTIVES", "0") not in ("", "0")
        self.calls = 0  # collectives issued (tests)
        # per-collective timing (bench.py --mode sharded): (stage, bytes, start event, end event) on the stream the collective is
        # enqueued on; read out by ``comm_report`` after a synchronisation.  Off unless a log list is attached.
        self.log = None
        self.stage = "factor"

    def _timed(self, fn, nbytes: int) -> None:
        if self.log is None:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.log.append((self.stage, nbytes, e0, e1))

    def _global(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def bcast(self, t: torch.Tensor, src: int) -> None:
        if not self.travel:
            return
        self.calls += 1
        if self.direct:
            self._timed(lambda: dist.broadcast(t, self._global(src), group=self.group), t.numel() * t.element_size())
            return

        def staged():
            h = t.detach().cpu() if self.rank == src else torch.empty(t.shape, dtype=t.dtype)
            dist.broadcast(h, self._global(src), group=self.group)
            if self.rank != src:
                t.copy_(h)

        self._timed(staged, t.numel() * t.element_size())

    def allreduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM) -> None:
        if not self.travel:
            return
        self.calls += 1
        if self.direct:
            self._timed(lambda: dist.all_reduce(t, op=op, group=self.group), t.numel() * t.element_size())
            return

        def staged():
            h = t.detach().cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)

        self._timed(staged, t.numel() * t.element_size())


#: bench.py attaches a list here to have every collective of the following evaluations timed with events (see _Comm._timed)
COMM_LOG = None


def comm_report(log) -> dict:
    """Per stage: collectives issued, bytes moved and the sum of their durations on their stream in ms (after a device
    synchronisation).  At one rank with GPP_SHARDED_FORCE_COLLECTIVES=1 this is the cost of the calls themselves."""
    out = {}
    for stage, nbytes, e0, e1 in log:
        d = out.setdefault(stage, {"calls": 0, "bytes": 0, "comm_ms": 0.0})
        d["calls"] += 1
        d["bytes"] += int(nbytes)
        d["comm_ms"] += e0.elapsed_time(e1)
    return out


class ShardedWorkspace:
    """Per-rank buffers of an N-point sharded evaluation (reused across evaluations).  Only ``A`` is N x N; the inverse factor and
    Ky^-1 exist as the OWNED column blocks, side by side: owned block q (global block rank + q P) in columns [q nb, (q+1) nb)."""

    def __init__(self, ctx: GppContext, N: int, nb: int, rank: int, world: int):
        dev = ctx.device
        self.N, self.nb, self.rank, self.world = N, nb, rank, world
        self.offs: List[int] = list(range(0, N, nb)) + [N]
        nblk = len(self.offs) - 1
        self.nq = len(range(rank, nblk, world))  # column blocks this rank owns
        wc = max(self.nq, 1) * nb
        self.A = square_buffer(N, dev)      # upper: Ky block rows (owned) -> the whole factor U; strict lower blocks: its mirror L
        # owned column blocks, compact: the forward sweep's running sums S, then Ky^-1 (lower) / the owned column blocks of L^-1
        self.Lc = torch.empty((N, wc), dtype=torch.float64, device=dev)
        self.Kc = torch.empty((N, wc), dtype=torch.float64, device=dev)
        self.D = torch.empty((nblk, nb, nb), dtype=torch.float64, device=dev)  # every diagonal block's inverse L_kk^-1 (+ mirror)
        # scratch of the row solves, THREE of them used in turn: the owner's consumers of a solved block row (next diagonal block,
        # trailing updates, mirror, packing) read it HERE, and the copy into the factor's place happens off every critical path.
        # (Three: step k's solve may overwrite what step k-3's bulk update read, and block row k's own last bulk update IS step
        # k-3's — the chain already waits for it; with two, every panel waited for the bulk update two steps back: 56.6 -> 63.3 ms.)
        self.W2 = [torch.empty((nb, self.A.stride(0)), dtype=torch.float64, device=dev)[:, :N] for _ in range(3)]
        self.W = self.W2[0]
        self.Tk = torch.empty((nb, nb), dtype=torch.float64, device=dev)       # scratch of a diagonal block's factorisation
        self.ld = self.A.stride(0)
        self.pack = torch.empty(N * nb, dtype=torch.float64, device=dev)       # the packed tail of a row slab of the factor
        self.hbuf = torch.empty(3 * nb * nb, dtype=torch.float64, device=dev)  # its packed head: diagonal block, block k+1, inverse
        self.comm_stream = torch.cuda.Stream(device=dev)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.dscr = torch.zeros(nb, nb, dtype=torch.float64, device=dev)       # scratch of the back-substitution's diagonal blocks
        self.z = torch.empty(N, dtype=torch.float64, device=dev)
        self.alpha = torch.empty(N, dtype=torch.float64, device=dev)
        self.r = torch.empty(N, dtype=torch.float64, device=dev)
        self.out3 = torch.empty(3, dtype=torch.float64, device=dev)
        self.info = torch.zeros(len(self.offs), dtype=torch.int32, device=dev)
        self.epoch = 0

    def dblk(self, k: int) -> torch.Tensor:
        """The inverse of diagonal block k (lower) and its mirror (strict upper)."""
        n = self.offs[k + 1] - self.offs[k]
        return self.D[k, :n, :n]

    def col(self, c: int) -> slice:
        """Columns of the compact buffers that hold global column block c (owned by this rank)."""
        q = (c - self.rank) // self.world
        return slice(q * self.nb, q * self.nb + self.offs[c + 1] - self.offs[c])

    def nbytes(self) -> int:
        """Device bytes of the matrices (what grows with N)."""
        return sum(t.untyped_storage().nbytes() for t in (self.A, self.Lc, self.Kc, self.D, *self.W2, self.pack))


_workspaces = {}


def _workspace(ctx: GppContext, N: int, nb: int, rank: int = 0, world: int = 1) -> ShardedWorkspace:
    key = (ctx.index, N, nb, rank, world)
    ws = _workspaces.get(key)
    if ws is None:
        _workspaces.clear()
        ws = ShardedWorkspace(ctx, N, nb, rank, world)
        _workspaces[key] = ws
    return ws


def _push_channel(ctx: GppContext, comm: "_Comm", ws: ShardedWorkspace):
    """The push transport's channel for this workspace's messages (GPP_SHARD_PUSH=1, several ranks), else None: the messages are
    broadcasts.  A slot holds the widest message: a head (nb x 2 nb + the nb x nb inverse) or a piece of a tail."""
    if not comm.travel or not _push.active(comm.world):
        return None
    if getattr(ws, "push_slot_bytes", None) is None:
        widest = max([c1 - c0 for k in range(len(ws.offs) - 1) for c0, c1 in ctx.shard_messages(ws.N, ws.nb, k)[1:]] + [0])
        ws.push_slot_bytes = 8 * ws.nb * max(3 * ws.nb, widest)
    return _push.channel(ctx, comm.rank, comm.world, comm.group, ws.push_slot_bytes)


class _RowEvents:
    """Which event marks the latest update of each block row, and on which stream it was recorded: a launch on another
    stream waits for the events of the rows it touches (launches on the producing stream are ordered anyway)."""

    def __init__(self):
        self.last = {}

    def produced(self, rows, stream) -> None:
        rows = list(rows)
        if not rows:
            return
        ev = torch.cuda.Event()
        ev.record(stream)
        for i in rows:
            self.last[i] = (ev, stream)

    def needed(self, rows, stream) -> None:
        seen = set()
        for i in rows:
            hit = self.last.get(i)
            if hit is not None and hit[1] is not stream and id(hit[0]) not in seen:
                seen.add(id(hit[0]))
                stream.wait_event(hit[0])


#: one step of look-ahead in the forward / backward sweeps (the small per-step products on a second stream)
_SWEEP_LOOKAHEAD = os.environ.get("GPP_SHARD_SWEEP_LOOKAHEAD", "0") not in ("", "0")
#: entries of the trailing matrix the CU-masked stream takes beside a running panel (the single-GPU driver's GPP_SPLIT_ELEMS)
_EARLY_ELEMS = int(os.environ.get("GPP_SHARD_EARLY_ELEMS", "0"))


def _factor(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace, U, w, sf2, tau, grp, kind, d_split, jitter: float) -> int:
    """Distributed build + Cholesky.  Returns the LAPACK-style info (0 = ok) agreed on by all ranks.

    Step k, stream by stream (``side`` = the library's 32-CU panel stream, ``upd`` = its 224-CU throughput stream, ``full`` = its
    stream without a CU mask, ``cs`` = collectives; events in capitals):
      owner of k   side: wait DIAG(k); factor + invert the diagonal block; FACTORED; solve the HEAD of the row (the columns of
                         block k+1)
                   upd : wait FACTORED; solve the TAIL of the row (columns from block k+2 on)
      everybody    cs  : broadcast head (+ diagonal block and its inverse), then tail  [nothing when there is one rank]
      owner of k+1 side: on the head: A[k+1, k+1] -= U[k, k+1]^T U[k, k+1]; DIAG(k+1) — the next panel starts here, while the
                         tail is still being solved / travelling: the chain of diagonal blocks carries no wide launch and never
                         leaves the 32 panel CUs
      everybody    upd : on the tail: rest of block row k+1, block row k+2 (the next steps' heads and diagonal blocks depend on
                         them), a few early rows beside the running panel;  full: the bulk, once that panel is done.
    """
    N, offs, P, me, nb = ws.N, ws.offs, comm.world, comm.rank, ws.nb
    nblk = len(offs) - 1
    A = ws.A
    main = torch.cuda.current_stream(ctx.index)
    side, upd, full = ctx.internal_streams()
    cs = ws.comm_stream
    owned = lambda a, b: [i for i in range(max(a, 0), min(b, nblk)) if i % P == me]  # noqa: E731
    pushc, pstat = _push_channel(ctx, comm, ws), ws.info[nblk:nblk + 1]  # (GPP_SHARD_PUSH=1; a wait's time-out lands in pstat)
    ws.info.zero_()
    for k in owned(0, nblk):
        ctx.kernel_build(U, w, sf2, tau, grp, A, jitter=jitter, kind=kind, d_split=d_split, uplo=UPLO_FULL, row0=offs[k],
                         nrows=offs[k + 1] - offs[k])
    for s in (side, upd, full, cs, ws.copy_stream):
        s.wait_stream(main)
    rows = _RowEvents()   # latest update of each block row (the build is ordered before everything by the waits above)
    diag_ready = None     # DIAG(k): the diagonal block of the step about to start carries every update
    pending = None        # the bulk of the previous step's update, held back until this step's panel is enqueued

    def row_update(i, c0, src, k_n, stream):
        """A[i, c0:] -= U[k, i]^T U[k, c0:] for block row i (columns from c0 on); ``src``: where block row k of the factor is read
        (all N columns addressed as in A)."""
        oi, oi1 = offs[i], offs[i + 1]
        ctx.gemm(1, 0, oi1 - oi, N - c0, k_n, -1.0, src[:, oi:oi1], src[:, c0:N], 1.0, A[oi:oi1, c0:N])

    w_free = [None, None, None]  # per scratch row: the event behind its last reader (the step that used it three steps ago)

    def issue_bulk(job, after=None):
        o_, o1_, first, arrived_, src_, slot_ = job
        with torch.cuda.stream(full):
            full.wait_event(arrived_)
            if after is not None:
                full.wait_event(after)
            bulk = owned(first, nblk)
            if bulk:
                oc = offs[first]
                rows.needed(bulk, full)
                ctx.syrk_rows(src_[:, oc:N], A[oc:N, oc:N], nb, first, me, P)
                rows.produced(bulk, full)
            bulk_done = torch.cuda.Event()
            bulk_done.record(full)
        # off EVERY critical path, on a stream of their own (on ``full`` they sat between two steps' bulk updates: measured
        # 56.6 -> 61.9 ms for the one-rank factor at C2): the mirror L[o1:, k] = U[k, o1:]^T for the back-substitution and, on the
        # owner, the solved row into the factor's place, where the sweeps read it
        cpy = ws.copy_stream
        with torch.cuda.stream(cpy):
            cpy.wait_event(arrived_)
            if o1_ < N:
                ctx.transpose(src_[:, o1_:N], A[o1_:N, o_:o1_])
                if slot_ is not None:
                    A[o_:o1_, o1_:N].copy_(src_[:, o1_:N])
            if slot_ is not None:
                cpy.wait_event(bulk_done)
                w_free[slot_] = torch.cuda.Event()
                w_free[slot_].record(cpy)

    for k in range(nblk):
        o, o1 = offs[k], offs[k + 1]
        o2 = offs[k + 2] if k + 2 <= nblk else N  # end of the head's columns (block k+1)
        nbk, own = o1 - o, (k % P == me)
        Lkk = ws.dblk(k)
        slot = k % 3 if own else None
        W = ws.W2[k % 3][:nbk]  # scratch of the row solves (the product cannot run in place)
        # where this step's consumers read block row k of the factor: the owner straight from the solve's output (round 5: the
        # strided copies into A were ~0.2 ms each on the critical path of every step), the others from the unpacked slab in A
        src = W if own else A[o:o1]
        head_solved = tail_solved = None
        if own:
            with torch.cuda.stream(side):
                if w_free[slot] is not None:
                    side.wait_event(w_free[slot])
                if diag_ready is not None:
                    side.wait_event(diag_ready)
                else:
                    rows.needed([k], side)
                ctx.potrf(A[o:o1, o:o1], Lkk, ws.info[k:k + 1], ws.Tk[:nbk, :nbk])
                ctx.trtri(A[o:o1, o:o1], Lkk, ws.Tk[:nbk, :nbk])
                factored = torch.cuda.Event()
                factored.record(side)
                # U12 = W_kk^T A12 (W_kk = mirrored inverse of the diagonal block).  The HEAD (the columns of block k+1) is
                # solved here, on the panel's own CUs: the chain factor -> head -> next diagonal block -> factor never queues
                # behind a wide launch of the throughput streams
                if o2 > o1:
                    rows.needed([k], side)  # block row k carries every update of the steps before k
                    ctx.gemm(1, 0, nbk, o2 - o1, nbk, 1.0, Lkk, A[o:o1, o1:o2], 0.0, W[:, o1:o2], a_mask=1, khi_mode=1)
                head_solved = torch.cuda.Event()
                head_solved.record(side)
            if pending is not None:
                # the bulk of step k-1 takes every CU, so it starts only when this diagonal block is done (a leaf needs a CU
                # to itself and would otherwise wait for the whole update to drain)
                issue_bulk(pending, after=factored)
                pending = None
            with torch.cuda.stream(upd):
                upd.wait_event(factored)  # (which follows w_free[slot] on the panel stream)
                rows.needed([k], upd)
                if N > o2:  # the TAIL of the row (columns from block k+2 on): one wide launch on the throughput CUs
                    ctx.gemm(1, 0, nbk, N - o2, nbk, 1.0, Lkk, A[o:o1, o2:N], 0.0, W[:, o2:N], a_mask=1, khi_mode=1)
                upd.wait_event(head_solved)  # (TAIL below stands for the whole row)
                tail_solved = torch.cuda.Event()
                tail_solved.record(upd)
        elif pending is not None:  # (cannot happen: a held-back bulk belongs to the rank that factors this block)
            issue_bulk(pending)
            pending = None
        if comm.travel:
            with torch.cuda.stream(cs):
                # only the meaningful part of the row travels (columns o..N, packed): half the xGMI volume of full rows
                wh = o2 - o
                head = ws.hbuf[:nbk * wh].view(nbk, wh)
                dblk = ws.hbuf[nbk * wh:nbk * (wh + nbk)].view(nbk, nbk)
                if own:
                    cs.wait_event(head_solved)
                if pushc is not None:  # (the same parts at the same places of the message, straight from / to where they lie)
                    seq = pushc.advance()
                    if own:
                        pushc.send(cs, seq, [(A[o:o1, o:o1], 0, wh), (W[:, o1:o2], o1 - o, wh), (Lkk, nbk * wh, nbk)], pstat)
                    else:
                        pushc.recv(cs, seq, [(A[o:o1, o:o2], 0, wh), (Lkk, nbk * wh, nbk)], pstat)
                    pushc.ack(cs, seq)
                else:
                    if own:
                        head[:, :o1 - o].copy_(A[o:o1, o:o1])   # the factored diagonal block ...
                        head[:, o1 - o:].copy_(W[:, o1:o2])      # ... and the solved head, from where the solve left it
                        dblk.copy_(Lkk)
                    comm.bcast(ws.hbuf[:nbk * (wh + nbk)], k % P)
                    if not own:
                        A[o:o1, o:o2].copy_(head)
                        Lkk.copy_(dblk)
                head_arrived = torch.cuda.Event()
                head_arrived.record(cs)
                # (the tail in the SAME pieces as the ticket lists' messages — gpp_shard_piece_cols in gpp.h — so that a rank on this
                #  launch path and a rank on the list path can take part in one evaluation: which of the two a rank runs depends on
                #  its own handle's options, e.g. a panel switched off after a time-out)
                for c0, c1 in ctx.shard_messages(N, nb, k)[1:]:
                    tail = ws.pack[:nbk * (c1 - c0)].view(nbk, c1 - c0)
                    if own:
                        cs.wait_event(tail_solved)
                    if pushc is not None:
                        seq = pushc.advance()
                        if own:
                            pushc.send(cs, seq, [(W[:, c0:c1], 0, c1 - c0)], pstat)
                        else:
                            pushc.recv(cs, seq, [(A[o:o1, c0:c1], 0, c1 - c0)], pstat)
                        pushc.ack(cs, seq)
                        continue
                    if own:
                        tail.copy_(W[:, c0:c1])
                    comm.bcast(ws.pack[:nbk * (c1 - c0)], k % P)
                    if not own:
                        A[o:o1, c0:c1].copy_(tail)
                arrived = torch.cuda.Event()
                arrived.record(cs)
        else:
            head_arrived, arrived = head_solved, tail_solved
        if k + 1 >= nblk:
            break
        # ---- this rank's share of the trailing update A[i, i:] -= U[k, i]^T U[k, i:], i > k -------------------------------
        panel_here = (k + 1) % P == me  # the next diagonal block is factored on THIS GPU while the update runs
        rem = N - o1
        extra = max(0, -(-_EARLY_ELEMS // max(rem, 1)) // nb - 2) if panel_here else 0
        e_end = min(k + 3 + extra, nblk)  # block rows [k+3, e_end) go early; [e_end, nblk) are the bulk
        if nblk - e_end < 2:
            e_end = nblk
        diag_ready = None
        if panel_here:
            with torch.cuda.stream(side):  # the next diagonal block, as soon as the head is here (panel CUs again)
                side.wait_event(head_arrived)
                rows.needed([k + 1], side)
                ctx.gemm(1, 0, o2 - o1, o2 - o1, nbk, -1.0, src[:, o1:o2], src[:, o1:o2], 1.0, A[o1:o2, o1:o2], c_tri=2)
                diag_ready = torch.cuda.Event()
                diag_ready.record(side)
        with torch.cuda.stream(upd):
            upd.wait_event(arrived)
            if panel_here and N > o2:
                row_update(k + 1, o2, src, nbk, upd)  # the rest of block row k+1: the next step's head and tail come from it
                rows.produced([k + 1], upd)
            for i in owned(k + 2, e_end):  # block row k+2 (the diagonal block after next), then the early rows: one launch each
                rows.needed([i], upd)
                row_update(i, offs[i], src, nbk, upd)
                rows.produced([i], upd)
        job = (o, o1, e_end, arrived, src, slot)
        if panel_here:
            pending = job  # issued in the next iteration, behind that panel
        else:
            issue_bulk(job)
    if pending is not None:
        issue_bulk(pending)
    for s in (side, upd, full, cs, ws.copy_stream):
        main.wait_stream(s)
    info = ws.info.max().to(torch.int32).reshape(1)
    comm.allreduce(info, dist.ReduceOp.MAX)
    return int(info.item())


#: GPP_SHARD_LIST=0: the launch-per-product factorisation and forward sweep of rounds 2-4 (also the library's own knob)
_USE_LIST = os.environ.get("GPP_SHARD_LIST", "1") not in ("", "0")
#: the factor's mirror beside the list on the copy stream or behind it (GPP_SHARD_MIRROR_BESIDE=1 / 0; default: see _mirror_beside)
_MIRROR_ENV = os.environ.get("GPP_SHARD_MIRROR_BESIDE", "")


def _mirror_beside(world: int) -> bool:
    """Where the factor's mirror L = U^T is written.  ONE rank: beside the list on the copy stream (same-box A/B in round 5: -1 ms
    at C2, -10 ms at C5 — its list keeps every CU busy and the strided copies hide behind it).  SEVERAL ranks: behind the list, as
    the library's LDS-tiled transposition on every CU.  Measured with tools/replay_rank.py (round 6, profiles/r06_virtual_rank.txt):
    a rank of a P = 8 run is bound by the chain of panels and messages, which lives on the 32 panel CUs — the executor's
    work-groups fill the other 224 to the last register whether they compute or wait — and the copy stream's strided
    transpositions (up to 150 MB each) landed on those same 32 CUs between the packing copies, the gates and the panel."""
    if _MIRROR_ENV != "":
        return _MIRROR_ENV != "0"
    return world <= 1
#: evaluations whose factorisation + forward sweep ran as a ticket list (tests)
LIST_EVALS = 0
BACK_LIST_EVALS = 0
#: work-groups of the list's executor (0 = two per throughput CU); tests in which several ranks share one GPU pass fewer
_LIST_WORKERS = int(os.environ.get("GPP_SHARD_WORKERS", "0"))


def _factor_list(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace, U, w, sf2, tau, grp, kind, d_split, jitter: float) -> Optional[int]:
    """Build + Cholesky + the forward sweep (the owned column blocks of L^-1 into ``Kc``) as ONE ticket list per rank
    (gpp_shard_list_begin in gpp.h; gp-plus_amd/csrc/gpp_dag.hip): tile tasks in dependency order taken by persistent work-groups,
    the rank's diagonal blocks on the panel stream, and the messages of ``_factor`` — per block row a head (diagonal block, its
    inverse, the columns of the next block) and a tail, in the same formats — on the communication stream, each behind a gate on
    its owner (the list has copied those strips into place) and in front of a signal on its receivers (the list's tasks that read
    them may run).  Returns the agreed info, or None when the list does not apply here (the caller runs ``_factor`` + ``_forward``)."""
    N, offs, P, me, nb = ws.N, ws.offs, comm.world, comm.rank, ws.nb
    nblk = len(offs) - 1
    A = ws.A
    if not ctx.dag_sched or me >= nblk:
        return None
    main = torch.cuda.current_stream(ctx.index)
    cs = ws.comm_stream
    pushc, pstat = _push_channel(ctx, comm, ws), ws.info[nblk:nblk + 1]  # (GPP_SHARD_PUSH=1; a wait's time-out lands in pstat)
    ws.info.zero_()
    for k in range(me, nblk, P):
        ctx.kernel_build(U, w, sf2, tau, grp, A, jitter=jitter, kind=kind, d_split=d_split, uplo=UPLO_FULL, row0=offs[k],
                         nrows=offs[k + 1] - offs[k])
    # (ordered against the caller's stream BEFORE the list starts: with the legacy default stream as the caller's, an event recorded on
    #  it behind the executor's launch completes only with the list — every blocking stream's earlier work precedes such a marker)
    cpy = ws.copy_stream
    cs.wait_stream(main)
    cpy.wait_stream(main)
    if not ctx.shard_list_begin(N, nb, me, P, A, ws.Kc, ws.Lc, ws.D, ws.W2, ws.info[0:1], _LIST_WORKERS):
        return None
    arrived = {}  # per block row of another rank: the event behind its unpacked tail
    try:
        if comm.travel:
            with torch.cuda.stream(cs):
                for k in range(nblk):
                    o, o1 = offs[k], offs[k + 1]
                    o2 = offs[k + 2] if k + 2 <= nblk else N
                    nbk, own = o1 - o, (k % P == me)
                    Lkk = ws.dblk(k)
                    wh = o2 - o
                    head = ws.hbuf[:nbk * wh].view(nbk, wh)
                    dblk = ws.hbuf[nbk * wh:nbk * (wh + nbk)].view(nbk, nbk)
                    if own:
                        ctx.shard_list_gate(cs, 0, k)
                    if pushc is not None:  # GPP_SHARD_PUSH=1: the owner's copies go straight from the factor into every peer's slot
                        seq = pushc.advance()
                        hparts = [(A[o:o1, o:o2], 0, wh), (Lkk, nbk * wh, nbk)]
                        if own:
                            pushc.send(cs, seq, hparts, pstat)
                        else:
                            pushc.recv(cs, seq, hparts, pstat)
                            ctx.shard_list_signal(cs, 0, k)
                        pushc.ack(cs, seq)
                    else:
                        if own:
                            head.copy_(A[o:o1, o:o2])
                            dblk.copy_(Lkk)
                        comm.bcast(ws.hbuf[:nbk * (wh + nbk)], k % P)
                        if not own:
                            A[o:o1, o:o2].copy_(head)
                            Lkk.copy_(dblk)
                            ctx.shard_list_signal(cs, 0, k)
                    # the tail in pieces (round 6; gpp_shard_piece_cols in gpp.h): the list's tasks wait for the piece of the column
                    # tile they read, so the next owner's first piece is updated, solved and sent while the rest of this row travels
                    for g, (c0, c1) in enumerate(ctx.shard_messages(N, nb, k)[1:]):
                        tail = ws.pack[:nbk * (c1 - c0)].view(nbk, c1 - c0)
                        if own:
                            ctx.shard_list_gate(cs, 1 + g, k)
                        if pushc is not None:
                            seq = pushc.advance()
                            if own:
                                pushc.send(cs, seq, [(A[o:o1, c0:c1], 0, c1 - c0)], pstat)
                            else:
                                pushc.recv(cs, seq, [(A[o:o1, c0:c1], 0, c1 - c0)], pstat)
                                ctx.shard_list_signal(cs, 1 + g, k)
                            pushc.ack(cs, seq)
                            continue
                        if own:
                            tail.copy_(A[o:o1, c0:c1])
                        comm.bcast(ws.pack[:nbk * (c1 - c0)], k % P)
                        if not own:
                            A[o:o1, c0:c1].copy_(tail)
                            ctx.shard_list_signal(cs, 1 + g, k)
                    if not own:
                        arrived[k] = torch.cuda.Event()
                        arrived[k].record(cs)
        # Beside the list, on a stream of its own: the factor's mirror L = U^T into A's strict lower triangle, which the
        # back-substitution reads row-contiguously — block row k as soon as it is in place (an owned one: behind the gates the packing
        # uses; another rank's: behind its unpacking).  Nothing in the list reads or writes the strict lower triangle.
        with torch.cuda.stream(cpy):
            for k in range(nblk - 1 if _mirror_beside(P) else 0):
                o, o1 = offs[k], offs[k + 1]
                if k % P == me:
                    for m in range(len(ctx.shard_messages(N, nb, k))):
                        ctx.shard_list_gate(cpy, m, k)
                else:
                    cpy.wait_event(arrived[k])
                A[o1:N, o:o1].copy_(A[o:o1, o1:N].t())
    finally:
        ctx.shard_list_end()
    main.wait_stream(cs)
    main.wait_stream(cpy)
    if not _mirror_beside(P):  # (several ranks: the mirror behind the list, as the library's transposition launches)
        for k in range(nblk - 1):
            ctx.transpose(A[offs[k]:offs[k + 1], offs[k + 1]:N], A[offs[k + 1]:N, offs[k]:offs[k + 1]])
    # what the list leaves to the launches behind it: the owned diagonal blocks of L^-1 (lower triangles of D) into Kc
    for c in range(me, nblk, P):
        blk = ws.Kc[offs[c]:offs[c + 1], ws.col(c)]
        blk.copy_(ws.dblk(c))
        blk.tril_()
    info = ws.info.max().to(torch.int32).reshape(1)
    comm.allreduce(info, dist.ReduceOp.MAX)
    st = int(info.item())
    if st == 0:
        global LIST_EVALS
        LIST_EVALS += 1
    elif st >= INFO_PANEL_TIMEOUT and os.environ.get("GPP_SHARD_DEBUG"):
        import ctypes
        buf = (ctypes.c_int * 1024)()
        n = ctx.lib.gpp_debug_dag_counters(ctx.h, buf, 1024)
        v = list(buf[:max(n, 0)])
        Bk = v[2] if n > 3 else 0
        names = ["PD", "G1D", "CPH", "CPT", "ART"]
        rows = {names[q]: v[3 + q * Bk:3 + (q + 1) * Bk] for q in range(5)} if Bk else {}
        print(f"[sharded rank {me}] ticket list: status {st:#x} (mine {int(ws.info[0].item()):#x}) abort {v[:1]} tickets {v[1:2]} {rows}", flush=True)
    return st


def _first_owned(ws: ShardedWorkspace, comm: _Comm) -> Optional[int]:
    return comm.rank if comm.rank < len(ws.offs) - 1 else None


def _forward(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace) -> None:
    """The owned column blocks of L^-1 (rows at and below their diagonal block) into ``Kc`` (compact), from the replicated factor and
    diagonal-block inverses: forward substitution of all owned column blocks together, one block row j at a time.  The owned
    blocks up to j (c = me, me+P, ... <= j, all nb wide) sit at the regular column spacing P*nb, so each step is ONE batched
    launch per product:
        Y_j[c] = -X_jj S_j[c]                (X_jj^T = the mirror in the upper part of the diagonal block; finished rows go to ``Kc``;
                                              the own block of the step is Y_j[j] = X_jj itself)
        S_k[c] += L[k, j] Y_j[c],  k > j     (L[k, j] = U[j, k]^T, shared by the whole batch; the sums S live in ``Lc``)
    with one step of look-ahead: the update's first block row (k = j+1) is issued on its own, and the small product of step j+1
    then runs on a second stream beside the rest of step j's update instead of alone on the chip."""
    N, offs, P, me, nb = ws.N, ws.offs, comm.world, comm.rank, ws.nb
    nblk = len(offs) - 1
    A, Lc, Kc = ws.A, ws.Lc, ws.Kc
    if _first_owned(ws, comm) is None:
        return
    main = torch.cuda.current_stream(ctx.index)
    aux = ctx.internal_streams()[2] if _SWEEP_LOOKAHEAD else main
    aux.wait_stream(main)
    for c in range(me, nblk - 1, P):  # the sums of the owned column blocks start from zero
        Lc[offs[c + 1]:N, ws.col(c)].zero_()
    row_done = torch.cuda.Event()  # block row j of the sums is complete
    row_done.record(main)
    for j in range(me, nblk):
        oj, oj1 = offs[j], offs[j + 1]
        nbj = oj1 - oj
        nleft = len(range(me, j, P))
        own = j % P == me
        Xjj = ws.dblk(j)
        with torch.cuda.stream(aux):
            aux.wait_event(row_done)
            if nleft > 0:  # (the owned blocks left of j are the first nleft compact blocks, at the column spacing nb)
                ctx.gemm_batched(1, 0, nbj, nb, nbj, -1.0, Xjj, 0, Lc[oj:oj1, 0:nb], nb, 0.0, Kc[oj:oj1, 0:nb], nb, nleft,
                                 a_mask=1, khi_mode=1)
            if own:  # Y_j[j] = X_jj: lower triangle, zeros above (the diagonal block's slot also holds the mirror)
                Kc[oj:oj1, ws.col(j)].copy_(Xjj)
                Kc[oj:oj1, ws.col(j)].tril_()
            y_done = torch.cuda.Event()
            y_done.record(aux)
        nb_all = nleft + (1 if own else 0)
        if oj1 < N and nb_all > 0:
            main.wait_event(y_done)
            oj2 = offs[j + 2]
            for r0, r1 in (((oj1, oj2), (oj2, N)) if _SWEEP_LOOKAHEAD else ((oj1, N),)):  # block row j+1 first: the next step's small product waits for it only
                if r1 > r0:
                    ctx.gemm_batched(1, 0, r1 - r0, nb, nbj, 1.0, A[oj:oj1, r0:r1], 0, Kc[oj:oj1, 0:nb], nb, 1.0,
                                     Lc[r0:r1, 0:nb], nb, nb_all)
                if r1 == oj2 and _SWEEP_LOOKAHEAD:
                    row_done = torch.cuda.Event()
                    row_done.record(main)
        else:
            main.wait_event(y_done)
    main.wait_stream(aux)


def _vectors(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace, need_alpha: bool) -> None:
    """z = L^-1 r, the scalars of the MLL, and alpha = L^-T z, from the owned column blocks of L^-1 in ``Kc``: each rank forms its
    part (sums over its columns / its entries) and the parts are added by an all-reduce of N doubles."""
    P, me, nb = comm.world, comm.rank, ws.nb
    ctx.trmv_lower_cols(ws.Kc, ws.r, ws.z, nb, me, P, trans=False, compact=True)
    comm.allreduce(ws.z)
    ctx.mll_scalars(ws.A, ws.z, ws.out3)
    if need_alpha:
        ctx.trmv_lower_cols(ws.Kc, ws.z, ws.alpha, nb, me, P, trans=True, compact=True)
        comm.allreduce(ws.alpha)


def _backward(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace) -> int:
    """The owned column blocks of Ky^-1 = L^-T L^-1 (rows at and below their diagonal block) into ``Lc``, by back-substitution of
    the owned column blocks Y of L^-1 (in ``Kc``) against the replicated factor:  U[c:, c:] Z = Y, block row j from the last
    one up:   Z_j = X_jj^T Y_j   (X_jj = L_jj^-1: the lower part of ``D[j]``), then the right-looking update
    Y_i -= U[i, j] Z_j of every block row c <= i < j — ONE TN GEMM per step over the lower-triangular tiles of the owned
    column blocks, its row-contiguous left operand being the factor's mirror L[j, i] in the strict lower triangle of ``A``.
    One step of look-ahead as in ``_forward``: block row j-1 of the update first, the small products of step j-1 on a second
    stream beside the rest.  No communication (bullet 4).  Returns 0, or the status of a ticket list that timed out
    (agreed on by all ranks; Kc is then destroyed and the caller evaluates again)."""
    offs, P, me, nb = ws.offs, comm.world, comm.rank, ws.nb
    nblk = len(offs) - 1
    A, Lc, Kc = ws.A, ws.Lc, ws.Kc
    # The status of the back-substitution's list is agreed on by ALL ranks, whether a rank's own list applied or not (a rank without
    # blocks, a size the list does not take): exactly one MAX all-reduce per evaluation when the lists are enabled at all — the
    # decision depends on the environment and on options every rank switches together, never on what a rank owns.
    agree = _USE_LIST and comm.travel

    def agreed(local: int) -> int:
        if not agree:
            return local
        info = torch.full((1,), int(local), dtype=torch.int32, device=ws.info.device)
        comm.allreduce(info, dist.ReduceOp.MAX)
        return int(info.item())

    if _first_owned(ws, comm) is None:
        return agreed(0)
    if _USE_LIST and ctx.dag_sched:
        # the same sweep as ONE ticket list (gpp_shard_back_list in gpp.h).  The diagonal blocks of Ky^-1 are written as lower
        # triangles: clear what the forward sweep's sums left above them first
        for c in range(me, nblk, P):
            Lc[offs[c]:offs[c + 1], ws.col(c)].zero_()
        used = ctx.shard_back_list(ws.N, nb, me, P, A, Kc, Lc, ws.D, ws.info[0:1], _LIST_WORKERS and 2 * _LIST_WORKERS)
        if not used and os.environ.get("GPP_SHARD_DEBUG"):
            print(f"[sharded rank {me}] back-substitution list: not used", flush=True)
        if used:
            # (a wait inside the list that ran out of its budget must not pass as a result — and must be EVERY rank's status: the
            #  gradient's all-reduce follows, so a rank that raised alone would leave the others blocked in it.  MAX over the ranks,
            #  as for the factor list; the caller then repeats the evaluation on the launch path, all ranks together.)
            mine = int(ws.info[0].item())
            st = agreed(mine)
            if st and os.environ.get("GPP_SHARD_DEBUG"):
                print(f"[sharded rank {me}] back-substitution list: status {st:#x} (mine {mine:#x})", flush=True)
            if st == 0:
                global BACK_LIST_EVALS
                BACK_LIST_EVALS += 1
            return st
    oc0 = offs[me]
    main = torch.cuda.current_stream(ctx.index)
    aux = ctx.internal_streams()[2] if _SWEEP_LOOKAHEAD else main
    row_done = torch.cuda.Event()
    row_done.record(main)
    for j in range(nblk - 1, me - 1, -1):
        oj, oj1 = offs[j], offs[j + 1]
        nbj = oj1 - oj
        nleft = len(range(me, j, P))  # owned column blocks strictly left of j
        Xjj = ws.dblk(j)
        with torch.cuda.stream(aux):
            aux.wait_event(row_done)
            if nleft > 0:
                ctx.gemm_batched(1, 0, nbj, nb, nbj, 1.0, Xjj, 0, Kc[oj:oj1, 0:nb], nb, 0.0, Lc[oj:oj1, 0:nb], nb, nleft,
                                 a_mask=2, klo_mode=1)
            if j % P == me:
                # the diagonal block of Ky^-1: both operands lower triangular, only its lower part is wanted (and only the lower
                # part of Y_j[j] is kept up to date) — the LAUUM shape; through the scratch, the product cannot run in place
                d = ws.dscr[:nbj, :nbj]
                ctx.gemm(1, 0, nbj, nbj, nbj, 1.0, Xjj, Kc[oj:oj1, ws.col(j)], 0.0, d, a_mask=2, b_mask=2, klo_mode=3, c_tri=1)
                Lc[oj:oj1, ws.col(j)].copy_(d)
            z_done = torch.cuda.Event()
            z_done.record(aux)
        main.wait_event(z_done)
        if nleft > 0:
            M = oj - oc0
            lo = max(offs[j - 1] - oc0, 0)
            for r0, r1 in (((lo, M), (0, lo)) if _SWEEP_LOOKAHEAD else ((0, M),)):  # block row j-1 first
                if r1 > r0:
                    ctx.gemm_lower_cols(A[oj:oj1, oc0:oj], Lc[oj:oj1], Kc[oc0:oj], -1.0, 1.0, nb, me, me, P, r0, r1, compact=True)
                if r0 == lo and _SWEEP_LOOKAHEAD:
                    row_done = torch.cuda.Event()
                    row_done.record(main)
    main.wait_stream(aux)
    return agreed(0)


class ShardedMLLFunction(torch.autograd.Function):
    """Same contract as ``linalg.ExactMLLFunction`` (value and gradients identical on every rank)."""

    @staticmethod
    def forward(ctx, U, w, sf2, tau, mean, y, grp, kind, d_split, dU, group, nb):
        dev = U.device
        gctx = get_context(dev)
        comm = _Comm(group)
        N, D = U.shape
        f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64).contiguous()
        Ud, wd, sd, td = f64(U), f64(w), f64(sf2).reshape(1), f64(tau).reshape(-1)
        if grp is not None and grp.dtype != torch.int32:
            grp = grp.to(torch.int32)
        comm.log = COMM_LOG
        ws = _workspace(gctx, N, nb, comm.rank, comm.world)
        ws.epoch += 1
        jitters = [0.0] + [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
        from .linalg import _stage
        need_grad = any(ctx.needs_input_grad[:6])
        for redo in (False, True):
            swept, used = ShardedMLLFunction._factor_with_jitter(gctx, comm, ws, Ud, wd, sd, td, grp, kind, d_split, jitters)
            if not swept:
                with _stage("shard_inverse"):
                    _forward(gctx, comm, ws)
            torch.sub(f64(y), f64(mean), out=ws.r)
            comm.stage = "vectors"
            _vectors(gctx, comm, ws, need_alpha=need_grad)
            st = 0
            if need_grad:
                with _stage("shard_backsolve"):
                    st = _backward(gctx, comm, ws)
            if st == 0:
                break
            # The back-substitution's ticket list timed out on some rank (the status is the MAX over the ranks, so every rank is
            # here): its input is destroyed.  Every rank switches the executor off and the evaluation is repeated on the
            # launch-per-product path — once; a second failure is reported on all ranks alike.
            if redo or not panel_timed_out(gctx, st):
                from .backend import check_status
                check_status(st)
                raise RuntimeError(f"sharded evaluation: back-substitution list status {st:#x}")
            comm.stage = "factor"
        ws.comm_calls = comm.calls  # (tests: the collectives really ran)
        ctx.saved = (gctx, comm, ws, ws.epoch, Ud, wd, sd, grp, td.numel(), kind, d_split, dU)
        ctx.in_dtypes = (U.dtype, w.dtype, sf2.dtype, tau.dtype, mean.dtype, y.dtype)
        ctx.shapes = (sf2.shape, tau.shape)
        return ws.out3[2].clone()

    @staticmethod
    def _factor_with_jitter(gctx, comm, ws, Ud, wd, sd, td, grp, kind, d_split, jitters):
        """gpytorch's psd_safe_cholesky policy around the distributed factorisation (optim/mll_torch.py:116 reaches it through
        ``log_prob``): returns (the list also swept the owned column blocks of L^-1, the jitter that succeeded)."""
        from .linalg import _stage
        used = None
        swept = False
        attempts = list(jitters)
        timeouts = 0
        while attempts:
            jit = attempts.pop(0)
            with _stage("shard_factor"):
                info = _factor_list(gctx, comm, ws, Ud, wd, sd, td, grp, kind, d_split, jit) if _USE_LIST else None
                swept = info is not None  # (the list builds the owned column blocks of L^-1 beside the factorisation)
                if info is None:
                    info = _factor(gctx, comm, ws, Ud, wd, sd, td, grp, kind, d_split, jit)
            if info >= INFO_PANEL_TIMEOUT:
                _push.disable()  # (a message may be missing somewhere: the ranks' message numbers no longer agree; broadcasts from here)
                # (the status is the MAX over the ranks: every rank sees it and repeats the attempt; the rank whose panel gave up —
                #  or every rank, it costs 1-2 % — switches the panel off)
                timeouts += 1
                if timeouts > 2:  # every rank has switched its panel off by now: this is not a time-out any more
                    from .backend import check_status
                    check_status(info)
                if gctx.coop_panel:
                    panel_timed_out(gctx, info)
                attempts.insert(0, jit)
                continue
            if info == 0:
                used = jit
                break
            if jit == 0.0:
                bad = [n for n, t in (("inputs", Ud), ("weights", wd), ("outputscale", sd), ("noise", td))
                       if not torch.isfinite(t).all()]
                if bad:
                    raise NanError(f"cholesky: NaN/Inf in {', '.join(bad)} of the covariance")
        if used is None:
            raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitters[-1]:.1e}.")
        if used > 0:
            warnings.warn(f"A not p.d., added jitter of {used:.1e} to the diagonal", RuntimeWarning)
        return swept, used

    @staticmethod
    def backward(ctx, grad_out):
        gctx, comm, ws, epoch, Ud, wd, sd, grp, S, kind, d_split, dU = ctx.saved
        if ws.epoch != epoch:
            raise RuntimeError("sharded evaluation: backward after another forward reused the buffers; call backward "
                               "before the next evaluation")
        N, D = Ud.shape
        dev = Ud.device
        need_U = ctx.needs_input_grad[0] and dU > 0
        nU = N * dU if need_U else 0
        flat = torch.zeros(D + 1 + S + nU, dtype=torch.float64, device=dev)
        g_w, g_s, g_t = flat[:D], flat[D:D + 1], flat[D + 1:D + 1 + S]
        g_Ud = flat[D + 1 + S:].view(N, dU) if need_U else None
        gctx.grad_reduce_cols(Ud, wd, sd, grp, S, ws.alpha, ws.Lc, dU if need_U else 0, ws.nb, comm.rank, comm.world, g_w,
                              g_s, g_t, g_Ud, kind=kind, d_split=d_split, compact=True)
        comm.stage = "grad"
        comm.allreduce(flat)
        ws.comm_calls = comm.calls
        g_U = None
        if ctx.needs_input_grad[0]:
            g_U = torch.zeros(N, D, dtype=torch.float64, device=dev)
            if need_U:
                g_U[:, :dU] = g_Ud
        go = grad_out.to(torch.float64)
        dt = ctx.in_dtypes
        sf2_shape, tau_shape = ctx.shapes
        alpha = ws.alpha
        return (None if g_U is None else (go * g_U).to(dt[0]),
                (go * g_w).to(dt[1]) if ctx.needs_input_grad[1] else None,
                (go * g_s).reshape(sf2_shape).to(dt[2]) if ctx.needs_input_grad[2] else None,
                (go * g_t).reshape(tau_shape).to(dt[3]) if ctx.needs_input_grad[3] else None,
                (go * alpha).to(dt[4]) if ctx.needs_input_grad[4] else None,
                (-go * alpha).to(dt[5]) if ctx.needs_input_grad[5] else None,
                None, None, None, None, None, None)


def sharded_mll(U: torch.Tensor, w: torch.Tensor, sf2: torch.Tensor, tau: torch.Tensor, mean: torch.Tensor, y: torch.Tensor,
                grp: Optional[torch.Tensor] = None, kind: int = KIND_RBF, d_split: int = 0, n_grad_dims: int = 0,
                group=None, nb: int = 1024) -> torch.Tensor:
    """log N(y | mean, sf2 k(U, U; w) + diag(tau[grp])) evaluated cooperatively by all ranks of ``group``."""
    if nb < 128 or nb % 128 != 0:
        raise ValueError("block height nb must be a multiple of 128")
    return ShardedMLLFunction.apply(U, w, sf2, tau, mean, y, grp, kind, d_split, int(n_grad_dims), group, int(nb))
