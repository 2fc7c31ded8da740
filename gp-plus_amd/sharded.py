"""Sharded single evaluation of the exact-GP marginal likelihood over the GPUs of one node (SURVEY.md §8(e) mode 2,
BASELINE config 5: N = 60 000 on 8 x MI355X).  One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI); every rank calls :func:`sharded_mll` with identical arguments and receives the identical value and gradients.

The reference has no multi-GPU evaluation (its only parallelism is joblib over restarts, optim/mll_scipy.py:287-293);
this is the same computation as ``linalg.ExactMLLFunction`` (reference call sites optim/mll_torch.py:114-117) with its
O(N^3) stages split by 1-D block-cyclic BLOCK ROWS of the upper-stored matrices (block height ``nb``, owner = k mod P):

  build    every rank builds the block rows of Ky it owns                                      (no communication)
  potrf    right-looking: the owner of block row k factors the diagonal block (leaf kernels), inverts it, solves the
           block row with one GEMM and BROADCASTS the finished row slab (nb x ld doubles) together with the inverse of
           the diagonal block; every rank then updates the block rows it owns with one TN GEMM each.  One step of
           look-ahead: the owner of k+1 updates that row first and factors / broadcasts it on a second stream while
           the remaining updates of step k run.
  inverse  column blocks of L^-1 are independent forward substitutions against the (now replicated) factor: the owner of
           column block c sweeps Y_k = -L_kk^-1 sum_{j<k} L_kj Y_j right-looking (one wide TN GEMM per step); the column
           blocks are then broadcast so that every rank holds L^-1 (lower) and its mirror (upper).
  lauum    every rank forms its tile-cyclic share (128-row tile rows) of Ky^-1 = L^-T L^-1, block row c as soon as column
           block c of the inverse has arrived (Ky^-1[i, j <= i] needs columns i and j only): the product runs on the main
           stream while the side stream broadcasts the next column block
  grad     ``gpp_grad_reduce_rows`` over the same tile rows, then ONE all-reduce of D + 1 + S (+ N dU) doubles.

Communication per evaluation: the packed factor slabs (4 N^2 B), the column blocks of the inverse (4 N^2 B) and the
tiny all-reduce; at C5 that is ~29 GB per GPU against ~2.7e13 flop of GEMM work per GPU.  Memory per GPU: the same three
N x N buffers as the single-GPU path (86 GB at C5 of 288 GB) — nothing is scattered, so every stage after the
factorisation reads local memory only.

Without RCCL (tests: two processes sharing one GPU over "gloo") the broadcasts are staged through host memory.
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .backend import KIND_RBF, UPLO_FULL, GppContext, get_context, square_buffer
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

    def _global(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def bcast(self, t: torch.Tensor, src: int) -> None:
        if self.world == 1:
            return
        if self.direct:
            dist.broadcast(t, self._global(src), group=self.group)
            return
        h = t.detach().cpu() if self.rank == src else torch.empty(t.shape, dtype=t.dtype)
        dist.broadcast(h, self._global(src), group=self.group)
        if self.rank != src:
            t.copy_(h)

    def allreduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM) -> None:
        if self.world == 1:
            return
        if self.direct:
            dist.all_reduce(t, op=op, group=self.group)
            return
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=self.group)
        t.copy_(h)


class ShardedWorkspace:
    """Per-rank buffers of an N-point sharded evaluation (reused across evaluations)."""

    def __init__(self, ctx: GppContext, N: int, nb: int):
        dev = ctx.device
        self.N, self.nb = N, nb
        self.A = square_buffer(N, dev)      # Ky block rows (owned) -> the whole factor U after the broadcasts
        self.Li = square_buffer(N, dev)     # L^-1 (lower) + mirror (upper)
        self.Ki = square_buffer(N, dev)     # scratch, then the owned block rows of Ky^-1 (lower)
        self.ld = self.A.stride(0)
        self.pack = torch.empty(2 * N * nb, dtype=torch.float64, device=dev)  # two slabs (factor rows / inverse column blocks)
        self.dbuf = torch.empty(nb * nb, dtype=torch.float64, device=dev)
        self.z = torch.empty(N, dtype=torch.float64, device=dev)
        self.alpha = torch.empty(N, dtype=torch.float64, device=dev)
        self.r = torch.empty(N, dtype=torch.float64, device=dev)
        self.out3 = torch.empty(3, dtype=torch.float64, device=dev)
        self.offs: List[int] = list(range(0, N, nb)) + [N]
        self.info = torch.zeros(len(self.offs), dtype=torch.int32, device=dev)
        self.epoch = 0

    def rows(self, buf: torch.Tensor, o: int, n: int) -> torch.Tensor:
        """Contiguous slab of ``n`` full rows (including the row padding) of one of the square buffers."""
        base = buf._base if buf._base is not None else buf
        return base[o:o + n]


_workspaces = {}


def _workspace(ctx: GppContext, N: int, nb: int) -> ShardedWorkspace:
    key = (ctx.index, N, nb)
    ws = _workspaces.get(key)
    if ws is None:
        _workspaces.clear()
        ws = ShardedWorkspace(ctx, N, nb)
        _workspaces[key] = ws
    return ws


def _factor(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace, U, w, sf2, tau, grp, kind, d_split, jitter: float) -> int:
    """Distributed build + Cholesky.  Returns the LAPACK-style info (0 = ok) agreed on by all ranks."""
    N, offs, P, me = ws.N, ws.offs, comm.world, comm.rank
    nblk = len(offs) - 1
    A, Li, Ki = ws.A, ws.Li, ws.Ki
    main = torch.cuda.current_stream(ctx.index)
    # The library's two CU-masked streams: diagonal blocks (single-work-group leaves that need a CU to themselves) are
    # factored on 32 reserved CUs while the other CUs run the trailing updates — see gpp_api.hip, ensure_streams.
    side, upd = ctx.internal_streams()
    ws.info.zero_()
    for k in range(me, nblk, P):
        ctx.kernel_build(U, w, sf2, tau, grp, A, jitter=jitter, kind=kind, d_split=d_split, uplo=UPLO_FULL, row0=offs[k],
                         nrows=offs[k + 1] - offs[k])
    side.wait_stream(main)
    upd.wait_stream(main)
    row_ready = torch.cuda.Event()
    row_ready.record(main)
    for k in range(nblk):
        o, o1 = offs[k], offs[k + 1]
        nbk, rem, own = o1 - o, N - o1, (k % P == me)
        with torch.cuda.stream(side):
            dblk = ws.dbuf[:nbk * nbk].view(nbk, nbk)
            if own:
                side.wait_event(row_ready)  # block row k carries every update of the steps before k
                Akk, Lkk, Tkk = A[o:o1, o:o1], Li[o:o1, o:o1], Ki[o:o1, o:o1]
                ctx.potrf(Akk, Lkk, ws.info[k:k + 1], Tkk)
                ctx.trtri(Akk, Lkk, Tkk)
                if rem > 0:
                    # U12 = W_kk^T A12 (W = mirrored inverse of the diagonal block), via the scratch: not in place
                    ctx.gemm(1, 0, nbk, rem, nbk, 1.0, Lkk, A[o:o1, o1:N], 0.0, Ki[o:o1, o1:N], a_mask=1, khi_mode=1)
                    A[o:o1, o1:N].copy_(Ki[o:o1, o1:N])
                dblk.copy_(Lkk)
            # only the meaningful part of the row slab travels (columns o..N, packed): half the xGMI volume of full rows
            slab = ws.pack[:nbk * (N - o)].view(nbk, N - o)
            if own:
                slab.copy_(A[o:o1, o:N])
            comm.bcast(slab, k % P)
            comm.bcast(dblk, k % P)
            if not own:
                A[o:o1, o:N].copy_(slab)
                Li[o:o1, o:o1].copy_(dblk)
            arrived = torch.cuda.Event()
            arrived.record(side)
        with torch.cuda.stream(upd):
            upd.wait_event(arrived)
            if k + 1 < nblk:
                # block row k+1 first (its owner factors it next), then every other owned block row in ONE launch
                o2 = offs[k + 2] if k + 2 <= nblk else N
                if (k + 1) % P == me:
                    ctx.gemm(1, 0, o2 - o1, N - o1, nbk, -1.0, A[o:o1, o1:o2], A[o:o1, o1:N], 1.0, A[o1:o2, o1:N])
                    row_ready = torch.cuda.Event()
                    row_ready.record(upd)
                if o2 < N:
                    ctx.syrk_rows(A[o:o1, o2:N], A[o2:N, o2:N], ws.nb, k + 2, me, P)
    main.wait_stream(side)
    main.wait_stream(upd)
    info = ws.info.max().to(torch.int32).reshape(1)
    comm.allreduce(info, dist.ReduceOp.MAX)
    return int(info.item())


def _inverse(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace) -> None:
    """L^-1 (lower) and its mirror (upper) on every rank, from the replicated factor and diagonal-block inverses."""
    N, offs, P, me = ws.N, ws.offs, comm.world, comm.rank
    nblk = len(offs) - 1
    A, Li, Ki = ws.A, ws.Li, ws.Ki
    nb, ld = ws.nb, ws.ld
    Li_base = Li._base if Li._base is not None else Li
    Ki_base = Ki._base if Ki._base is not None else Ki
    for c in range(me, nblk, P):
        oc, oc1 = offs[c], offs[c + 1]
        if oc1 < N:
            Li[oc1:N, oc:oc1].zero_()
    # Forward substitution of all owned column blocks together, one block row j at a time.  The owned blocks left of j
    # (c = me, me+P, ... < j, all nb wide) sit at the regular column spacing P*nb, so each step is ONE batched launch
    # per product:  Y_j[c] = -X_jj S_j[c]  (X_jj^T = the mirror in the upper part of the diagonal block), then
    # S_k[c] += L[k, j] Y_j[c] for every block k below (L[k, j] = U[j, k]^T, shared by the whole batch).
    oc0 = offs[me] if me < nblk else 0
    for j in range(nblk):
        oj, oj1 = offs[j], offs[j + 1]
        nbj = oj1 - oj
        nleft = (j - me + P - 1) // P if j > me else 0
        if nleft > 0:
            ctx.gemm_batched(1, 0, nbj, nb, nbj, -1.0, Li[oj:oj1, oj:oj1], 0, Li[oj:oj1, oc0:oc0 + nb], P * nb, 0.0,
                             Ki[oj:oj1, oc0:oc0 + nb], P * nb, nleft, a_mask=1, khi_mode=1)
            shape, strides, start = (nbj, nleft, nb), (ld, P * nb, 1), oj * ld + oc0
            Li_base.as_strided(shape, strides, start).copy_(Ki_base.as_strided(shape, strides, start))
            if oj1 < N:
                ctx.gemm_batched(1, 0, N - oj1, nb, nbj, 1.0, A[oj:oj1, oj1:N], 0, Li[oj:oj1, oc0:oc0 + nb], P * nb, 1.0,
                                 Li[oj1:N, oc0:oc0 + nb], P * nb, nleft)
        if j % P == me and oj1 < N:
            # own column block starts here: Y_j = X_jj itself (lower triangular; its slot also holds the mirror above the
            # diagonal, masked out: keep k >= n)
            ctx.gemm(1, 0, N - oj1, nbj, nbj, 1.0, A[oj:oj1, oj1:N], Li[oj:oj1, oj:oj1], 1.0, Li[oj1:N, oj:oj1], b_mask=2,
                     klo_mode=2)


def _exchange_inverse(ctx: GppContext, comm: _Comm, ws: ShardedWorkspace, with_lauum: bool) -> None:
    """Broadcast the column blocks of L^-1 (packed: the rows below each diagonal block) so that every rank holds L^-1 (lower)
    and its mirror (upper) — and, when the gradient is wanted, PIPELINE this rank's share of Ky^-1 = L^-T L^-1 with the
    broadcasts: Ky^-1[i, j <= i] needs the column blocks i and j of L^-1 only, so the tile rows of block row c are formed
    (``gpp_lauum_rows_range``, on the calling stream) as soon as column block c has arrived, while the side stream moves
    column block c+1.  4 N^2 bytes per GPU travel here against N^3 / (3 P) flops of product per GPU: at C5 on 8 GPUs 14 GB
    beside 9e12 flop, i.e. comparable times — serialised they would add up."""
    N, offs, P, me = ws.N, ws.offs, comm.world, comm.rank
    nblk = len(offs) - 1
    Li = ws.Li
    main = torch.cuda.current_stream(ctx.index)
    side, _ = ctx.internal_streams()
    side.wait_stream(main)  # the sweeps that produced the owned column blocks
    # two packing buffers so that the unpack of block c and the broadcast of block c+1 never share memory
    half = ws.pack.numel() // 2
    bounds, b = [], 0  # exclusive ends of the groups of block rows: halves of what is left, at most 6 groups
    if P == 1:
        bounds, b = [nblk], nblk  # nothing travels: one launch, as on the single-GPU path
    while b < nblk:
        b = nblk if len(bounds) == 5 else b + max(1, (nblk - b + 1) // 2)
        bounds.append(b)
    group_start = 0
    for c in range(nblk):
        oc, oc1 = offs[c], offs[c + 1]
        if oc1 < N:
            wc = oc1 - oc
            with torch.cuda.stream(side):
                if P > 1:
                    base = (c & 1) * half
                    buf = ws.pack[base:base + (N - oc1) * wc].view(N - oc1, wc)
                    if c % P == me:
                        buf.copy_(Li[oc1:N, oc:oc1])
                    comm.bcast(buf, c % P)
                    if c % P != me:
                        Li[oc1:N, oc:oc1].copy_(buf)
                ctx.transpose(Li[oc1:N, oc:oc1], Li[oc:oc1, oc1:N])  # mirror (tiled through LDS: gpp_transpose)
        # (the last column block is its diagonal block, which every rank already has)
        if with_lauum and c + 1 == bounds[0]:
            # a GROUP of block rows per launch: one launch per block row leaves the early rows (a handful of tiles with the
            # longest K ranges, several ms each) alone on the chip — measured with one rank at N = 20000: 70 ms for 20
            # launches against 41 ms for the single launch.  Groups halve what is left ([0, 1/2), [1/2, 3/4), ...): every
            # launch has hundreds of tiles, and the heavy early broadcasts still overlap the previous group's product.
            arrived = torch.cuda.Event()
            arrived.record(side)
            main.wait_event(arrived)
            ctx.lauum_rows_range(ws.Li, ws.Ki, me, P, offs[group_start], oc1)
            group_start = bounds.pop(0)
    main.wait_stream(side)


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
        ws = _workspace(gctx, N, nb)
        ws.epoch += 1
        jitters = [0.0] + [settings.cholesky_jitter.value() * (10 ** i) for i in range(settings.cholesky_max_tries.value())]
        used = None
        from .linalg import _stage
        for jit in jitters:
            with _stage("shard_factor"):
                info = _factor(gctx, comm, ws, Ud, wd, sd, td, grp, kind, d_split, jit)
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
        need_grad = any(ctx.needs_input_grad[:6])
        with _stage("shard_inverse"):
            _inverse(gctx, comm, ws)
            # (with a gradient: this rank's share of Ky^-1 is formed here too, block row by block row behind the broadcasts)
            _exchange_inverse(gctx, comm, ws, with_lauum=need_grad)
        torch.sub(f64(y), f64(mean), out=ws.r)
        gctx.mll_reduce(ws.A, ws.Li, ws.r, ws.z, ws.out3)
        if need_grad:
            gctx.alpha(ws.Li, ws.z, ws.alpha)
        ctx.saved = (gctx, comm, ws, ws.epoch, Ud, wd, sd, grp, td.numel(), kind, d_split, dU)
        ctx.in_dtypes = (U.dtype, w.dtype, sf2.dtype, tau.dtype, mean.dtype, y.dtype)
        ctx.shapes = (sf2.shape, tau.shape)
        return ws.out3[2].clone()

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
        gctx.grad_reduce_rows(Ud, wd, sd, grp, S, ws.alpha, ws.Ki, dU if need_U else 0, 128, comm.rank, comm.world, g_w,
                              g_s, g_t, g_Ud, kind=kind, d_split=d_split)
        comm.allreduce(flat)
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
