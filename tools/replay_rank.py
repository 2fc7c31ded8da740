"""Virtual-rank replay: every rank of a P-rank sharded evaluation, measured one at a time on ONE GPU.

The test boxes have one GPU, so the multi-GPU leg of DESIGN.md section 7 was a model.  This tool turns it into a measurement of
everything a rank does itself — with the DEFAULT multi-rank configuration (filler launches ON, one work-group per panel CU, and a
kernel with a collective's footprint resident on the communication stream), which had never executed on a GPU:

  1. the problem is factored and inverted once by the single-GPU path (gpp_potrf_ws + gpp_trtri [+ gpp_lauum for --check]): the
     reference block rows U[k, :] and diagonal inverses that other ranks WOULD send;
  2. rank r's REAL lists run (gpp_shard_list_begin ... _end, then gpp_shard_back_list; P ranks, nb = 1024, GPP_SHARD_FILL default)
     exactly as gp-plus_amd/sharded.py::_factor_list drives them — same gates, packing copies, signals, mirror stream — except that a
     block row of another rank is PLAYED into the message buffer by `gpp_debug_replay_copy` (gpp_shard.hip): `wgs` work-groups x
     `threads` threads that start no earlier than the moment the owner could have sent it and move the bytes no faster than
     `rate` GB/s; the rank's own messages occupy the communication stream for bytes / rate in the same way;
  3. "the moment the owner could have sent it" is itself measured: behind each of the rank's OWN gates a stamp kernel records when
     the head / tail could start on its communication stream (ready, and the stream free of the previous message — which ends at
     the same time on every rank; 100 MHz device clock, relative to the start of the list).  Ranks are replayed in turn
     (r = 0 .. P-1, each using the latest ready times of the others: the dependencies are triangular in the block index, so a
     sweep settles at least P more block rows) until the ready times stop moving: a self-consistent P-GPU timeline.

What it cannot contain: contention on the xGMI fabric, RCCL's own protocol latency beyond the footprint and the rate, the skew
between GPUs of one node (the pool's boxes differ by up to 5 %), host-side jitter of P processes.  When a SCALE_rNN.json exists it is
read against THIS (profiles/r06_virtual_rank.txt).

usage: python tools/replay_rank.py --config C5 [--P 8] [--rates 400,150,70,50] [--sweeps 8] [--check] [--json out.json]
       python tools/replay_rank.py --n 13000 --d 8 --P 4 --ranks 1 --rates 70 --check
"""
import argparse
import copy
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpplus_amd import sharded  # noqa: E402
from gpplus_amd.backend import UPLO_FULL, UPLO_UPPER, get_context, square_buffer  # noqa: E402

TICK_US = 0.01  # one tick of the 100 MHz constant clock


class _NoComm:
    """What sharded._vectors / _backward need of a communicator when nothing travels (the all-reduces of N doubles are not replayed)."""

    def __init__(self, rank, world):
        self.rank, self.world, self.travel, self.calls, self.log, self.stage = rank, world, False, 0, None, "factor"

    def allreduce(self, t, op=None):
        pass

    def bcast(self, t, src):
        pass


def make_inputs(config, n, d, dev):
    """(U, w, sf2, tau, r) on the device: a BASELINE config at its evaluation point, or random points (--n / --d)."""
    if config:
        from gpplus_amd.test_functions.baseline_configs import make_config

        X, y, _, theta = make_config(config, n)
        raw = [v for k, v in theta.items() if k.endswith("raw_lengthscale")][0].reshape(-1)
        w = (10.0 ** raw.double())  # Rough_RBF: lengthscale = 2^-1/2 10^(-omega/2), w = 1 / (2 l^2) (models/gp_plus.py:249-253)
        sf2 = torch.nn.functional.softplus(theta["covar_module.raw_outputscale"].double()).reshape(1)
        tau = (torch.exp(theta["likelihood.noise_covar.raw_noise"].double()) + 1e-8).reshape(-1)[:1]
        U, r = X.double(), (y - y.mean()).double()
    else:
        g = torch.Generator().manual_seed(0)
        U = torch.rand(n, d, generator=g, dtype=torch.float64) * 4.0
        r = torch.sin(U[:, 0]) + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
        w = torch.full((d,), 0.1, dtype=torch.float64)
        sf2 = torch.tensor([0.85], dtype=torch.float64)
        tau = torch.tensor([2.5e-3], dtype=torch.float64)
    return tuple(t.to(dev).contiguous() for t in (U, w, sf2, tau, r))


class Replay:
    def __init__(self, dev, U, w, sf2, tau, r, nb, P, wgs=16, threads=512, keep_kinv=False):
        self.dev, self.U, self.w, self.sf2, self.tau, self.r = dev, U, w, sf2, tau, r
        self.N, self.D = U.shape
        self.nb, self.P, self.wgs, self.threads = nb, P, wgs, threads
        self.push = False  # --push: the owner's side of the push transport (no packing copy)
        self.ctx = get_context(dev)
        lib = self.ctx.lib
        lib.gpp_debug_replay_copy.restype = ctypes.c_int
        lib.gpp_debug_replay_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_longlong,
                                              ctypes.c_void_p]
        lib.gpp_debug_replay_stamp.restype = ctypes.c_int
        lib.gpp_debug_replay_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib = lib
        self._reference(keep_kinv)
        self.ws0 = sharded.ShardedWorkspace(self.ctx, self.N, nb, 0, P)  # rank 0 owns the most blocks: every rank's buffers fit in its
        self.offs = self.ws0.offs
        self.nblk = len(self.offs) - 1
        self.sink = torch.empty_like(self.ws0.pack)
        self.msgs = [self.ctx.shard_messages(self.N, nb, k) for k in range(self.nblk)]  # per block row: column ranges of its messages
        self.M = max(len(m) for m in self.msgs)
        self.stamps = torch.zeros((self.nblk, self.M, 4), dtype=torch.int64, device=dev)  # [block][message][own ready, start, end, unpacked]
        self.epoch = torch.zeros(2, dtype=torch.int64, device=dev)
        self.after_list = None

    # ---- the single-GPU result: what the other ranks would send, and what this rank's results are compared with ----------------------
    def _reference(self, keep_kinv):
        ctx, N = self.ctx, self.N
        self.Aref, self.Liref, T = (square_buffer(N, self.dev) for _ in range(3))
        info = torch.zeros(1, dtype=torch.int32, device=self.dev)
        ctx.kernel_build(self.U, self.w, self.sf2, self.tau, None, self.Aref, uplo=UPLO_UPPER)
        self.Liref.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.potrf(self.Aref, self.Liref, info, T)
        ctx.trtri(self.Aref, self.Liref, T)
        torch.cuda.synchronize()
        self.ref_factor_inverse_ms = 1e3 * (time.perf_counter() - t0)
        assert int(info.item()) == 0, hex(int(info.item()))
        self.Kiref = None
        if keep_kinv:
            ctx.lauum(self.Liref, T)
            torch.cuda.synchronize()
            self.Kiref = T
        else:
            del T
            torch.cuda.empty_cache()

    def _ws(self, r):
        ws = copy.copy(self.ws0)
        ws.rank = r
        ws.nq = len(range(r, self.nblk, self.P))
        wc = max(ws.nq, 1) * self.nb
        ws.Lc = self.ws0.Lc.reshape(-1)[:self.N * wc].view(self.N, wc)
        ws.Kc = self.ws0.Kc.reshape(-1)[:self.N * wc].view(self.N, wc)
        return ws

    def _rcopy(self, stream, dst, src, rate, not_before, slot):
        assert src.shape == dst.shape and src.shape[1] % 2 == 0, (src.shape, dst.shape)
        rc = self.lib.gpp_debug_replay_copy(ctypes.c_void_p(stream.cuda_stream), dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0),
                                            src.shape[0], src.shape[1], self.wgs, self.threads, float(rate), self.epoch.data_ptr(),
                                            int(not_before), slot)
        assert rc == 0, rc

    def _stamp(self, stream, slot):
        rc = self.lib.gpp_debug_replay_stamp(ctypes.c_void_p(stream.cuda_stream), slot, self.epoch.data_ptr())
        assert rc == 0, rc

    # ---- one rank ------------------------------------------------------------------------------------------------------------------
    def run(self, r, rate, ready, check=False):
        """Rank r's evaluation with the other ranks' block rows replayed at `rate` GB/s (<= 0: as fast as the copy goes), message m of
        block row k (0 the head, 1 + g piece g of the tail) no earlier than ready[(k, m)] microseconds after the start.  Returns
        times, the rank's own ready times and status."""
        ctx, ws, N, nb, P, offs, nblk = self.ctx, self._ws(r), self.N, self.nb, self.P, self.offs, self.nblk
        dev, A = self.dev, self.ws0.A
        main = torch.cuda.current_stream(ctx.index)
        cs, cpy = ws.comm_stream, ws.copy_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        self.stamps.zero_()
        ws.info.zero_()
        comm = _NoComm(r, P)
        sp = self.stamps.data_ptr()
        slot = lambda k, m, q: sp + 8 * ((k * self.M + int(m)) * 4 + q)  # noqa: E731
        ev[0].record(main)
        for k in range(r, nblk, P):
            ctx.kernel_build(self.U, self.w, self.sf2, self.tau, None, A, uplo=UPLO_FULL, row0=offs[k], nrows=offs[k + 1] - offs[k])
        self.epoch.zero_()
        rc = self.lib.gpp_debug_replay_stamp(ctypes.c_void_p(main.cuda_stream), self.epoch.data_ptr(), None)
        assert rc == 0
        ev[1].record(main)
        for s in (cs, cpy):
            s.wait_stream(main)
        if not ctx.shard_list_begin(N, nb, r, P, A, ws.Kc, ws.Lc, ws.D, ws.W2, ws.info[0:1], 0):
            raise RuntimeError("the ticket list does not apply to this size / block height")
        arrived = {}
        try:
            with torch.cuda.stream(cs):
                for k in range(nblk):
                    o, o1 = offs[k], offs[k + 1]
                    o2 = offs[k + 2] if k + 2 <= nblk else N
                    nbk, own = o1 - o, (k % P == r)
                    Lkk = ws.dblk(k)
                    wh = o2 - o
                    head = ws.hbuf[:nbk * wh].view(nbk, wh)
                    dblk = ws.hbuf[nbk * wh:nbk * (wh + nbk)].view(nbk, nbk)
                    if own:
                        ctx.shard_list_gate(cs, 0, k)
                        self._stamp(cs, slot(k, 0, 0))
                        if self.push:  # the push transport (GPP_SHARD_PUSH=1): no packing, the copies read the factor where it lies
                            self._rcopy(cs, self.sink[:nbk * wh].view(nbk, wh), A[o:o1, o:o2], rate, -1, slot(k, 0, 1))
                            self._rcopy(cs, self.sink[:nbk * nbk].view(nbk, nbk), Lkk, rate, -1, slot(k, 0, 1))
                        else:
                            head.copy_(A[o:o1, o:o2])
                            dblk.copy_(Lkk)
                            msg = ws.hbuf[:nbk * (wh + nbk)].view(nbk, wh + nbk)  # the broadcast: the stream is busy for bytes / rate
                            self._rcopy(cs, self.sink[:msg.numel()].view_as(msg), msg, rate, -1, slot(k, 0, 1))
                    else:
                        self._rcopy(cs, head, self.Aref[o:o1, o:o2], rate, ready[(k, 0)] / TICK_US, slot(k, 0, 1))
                        self._rcopy(cs, dblk, self.Liref[o:o1, o:o1], rate, -1, slot(k, 0, 1))
                        A[o:o1, o:o2].copy_(head)
                        Lkk.copy_(dblk)
                        self._stamp(cs, slot(k, 0, 3))
                        ctx.shard_list_signal(cs, 0, k)
                    for g, (c0, c1) in enumerate(ctx.shard_messages(N, nb, k)[1:]):  # the tail's pieces
                        tail = ws.pack[:nbk * (c1 - c0)].view(nbk, c1 - c0)
                        if own:
                            ctx.shard_list_gate(cs, 1 + g, k)
                            self._stamp(cs, slot(k, 1 + g, 0))
                            if self.push:
                                self._rcopy(cs, self.sink[:tail.numel()].view_as(tail), A[o:o1, c0:c1], rate, -1, slot(k, 1 + g, 1))
                            else:
                                tail.copy_(A[o:o1, c0:c1])
                                self._rcopy(cs, self.sink[:tail.numel()].view_as(tail), tail, rate, -1, slot(k, 1 + g, 1))
                        else:
                            self._rcopy(cs, tail, self.Aref[o:o1, c0:c1], rate, ready[(k, 1 + g)] / TICK_US, slot(k, 1 + g, 1))
                            A[o:o1, c0:c1].copy_(tail)
                            self._stamp(cs, slot(k, 1 + g, 3))
                            ctx.shard_list_signal(cs, 1 + g, k)
                    if not own:
                        arrived[k] = torch.cuda.Event()
                        arrived[k].record(cs)
            with torch.cuda.stream(cpy):  # the factor's mirror beside the list, as sharded.py writes it
                for k in range(nblk - 1 if sharded._mirror_beside(P) else 0):
                    o, o1 = offs[k], offs[k + 1]
                    if k % P == r:
                        for m in range(len(ctx.shard_messages(N, nb, k))):
                            ctx.shard_list_gate(cpy, m, k)
                    else:
                        cpy.wait_event(arrived[k])
                    A[o1:N, o:o1].copy_(A[o:o1, o1:N].t())
        finally:
            ctx.shard_list_end()
        if self.after_list is not None:  # (--trace-rank: the factor + forward list is the handle's most recent plan exactly here)
            self.after_list()
        for s in (cs, cpy):
            main.wait_stream(s)
        if not sharded._mirror_beside(P):  # (the mirror behind the list, as the library's transposition launches on every CU)
            for k in range(nblk - 1):
                ctx.transpose(A[offs[k]:offs[k + 1], offs[k + 1]:N], A[offs[k + 1]:N, offs[k]:offs[k + 1]])
        for c in range(r, nblk, P):
            blk = ws.Kc[offs[c]:offs[c + 1], ws.col(c)]
            blk.copy_(ws.dblk(c))
            blk.tril_()
        ev[2].record(main)
        ws.r.copy_(self.r)
        sharded._vectors(ctx, comm, ws, True)
        ev[3].record(main)
        st_ff = int(ws.info[0].item())
        out = {"rank": r, "status_ff": st_ff}
        if check and st_ff == 0:
            out.update(self._check_forward(ws, r))
        st_back = sharded._backward(ctx, comm, ws)
        ev[4].record(main)
        D = self.D
        flat = torch.zeros(D + 2, dtype=torch.float64, device=dev)
        ctx.grad_reduce_cols(self.U, self.w, self.sf2, None, 1, ws.alpha, ws.Lc, 0, nb, r, P, flat[:D], flat[D:D + 1], flat[D + 1:], None, compact=True)
        ev[5].record(main)
        torch.cuda.synchronize()
        if check and st_ff == 0 and st_back == 0:
            out.update(self._check_back(ws, r))
        st = self.stamps.cpu().numpy().astype(np.float64) * TICK_US  # microseconds since the epoch
        own = list(range(r, nblk, P))
        nm = [len(m) for m in self.msgs]
        out.update(status_back=int(st_back), build_ms=ev[0].elapsed_time(ev[1]), ff_ms=ev[1].elapsed_time(ev[2]),
                   vec_ms=ev[2].elapsed_time(ev[3]), back_ms=ev[3].elapsed_time(ev[4]), grad_ms=ev[4].elapsed_time(ev[5]),
                   ready={(k, m): st[k, m, 0] for k in own for m in range(nm[k])},   # own messages: the gate passed
                   start=st[:, :, 1], end=st[:, :, 2], unpacked=st[:, :, 3])         # every message: on this rank's stream
        # the chain as this rank sees it: from the unpacked head of block k - 1 (another rank's) to its own head of block k being
        # ready to send — diagonal update, gate, panel, head solve, copies, gate (DESIGN.md section 7 assumed ~1.2 ms)
        out["chain_us"] = {k: st[k, 0, 0] - st[k - 1, 0, 3] for k in own if k > 0 and (k - 1) % P != r and st[k - 1, 0, 3] > 0}
        # ... and the tail's chain: from the unpacked FIRST piece of block row k - 1's tail to the own first piece being ready
        out["tail_chain_us"] = {k: st[k, 1, 0] - st[k - 1, 1, 3] for k in own if k > 0 and (k - 1) % P != r and nm[k] > 1 and st[k - 1, 1, 3] > 0}
        return out

    def trace(self, r, rate, ready):
        """Per-task trace of rank r's factor + forward list (tools/dag_check.py's report) in the converged timeline."""
        from dag_check import trace_report

        def on():
            torch.cuda.synchronize()
            assert self.lib.gpp_debug_dag_trace(self.ctx.h, 1) == 0

        def report():
            torch.cuda.synchronize()
            trace_report(self.ctx, self.N)
            self.lib.gpp_debug_dag_trace(self.ctx.h, 0)

        self.after_list = on
        self.run(r, rate, ready)
        self.after_list = report
        o = self.run(r, rate, ready)
        self.after_list = None
        print(f"  rank {r}'s communication stream, ms since its list started (own rows *: 'ready' = the gate passed; others: the earliest start):")
        print("    block owner | head: ready   sent/received   unpacked | tail, first piece: ready  sent/recvd  unpacked | last piece: ready  sent/recvd  unpacked (pieces)")
        ms = lambda v: f"{v / 1e3:10.3f}"  # noqa: E731
        for k in range(self.nblk):
            own, nm = k % self.P == r, len(self.msgs[k])
            rdy = lambda m: o["ready"][(k, m)] if own else ready[(k, m)]  # noqa: E731
            line = f"    {k:5d} {k % self.P:5d}{'*' if own else ' '}|   {ms(rdy(0))}      {ms(o['end'][k, 0])} {ms(o['unpacked'][k, 0])} |"
            if nm > 1:
                line += f"        {ms(rdy(1))} {ms(o['end'][k, 1])} {ms(o['unpacked'][k, 1])} |"
                line += f"  {ms(rdy(nm - 1))} {ms(o['end'][k, nm - 1])} {ms(o['unpacked'][k, nm - 1])} ({nm - 1})"
            print(line)
            if nm > 2 and os.environ.get("REPLAY_PIECES"):
                print("                  pieces ready>end: " + "  ".join(f"{rdy(m) / 1e3:.2f}>{o['end'][k, m] / 1e3:.2f}" for m in range(1, nm)))
        return o

    # ---- comparisons with the single-GPU result (--check, tests/test_gpu_replay.py) ----------------------------------------------------
    def _rel(self, a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))

    def _check_forward(self, ws, r):
        """Owned block rows of the factor (upper part) and owned column blocks of L^-1 (at and below the diagonal) vs the reference."""
        offs, N = self.offs, self.N
        e_fac = e_inv = 0.0
        for k in range(r, self.nblk, self.P):
            o, o1 = offs[k], offs[k + 1]
            e_fac = max(e_fac, self._rel(torch.triu(self.ws0.A[o:o1, o:N]), torch.triu(self.Aref[o:o1, o:N])))
            e_inv = max(e_inv, self._rel(torch.tril(ws.Kc[o:N, ws.col(k)]), torch.tril(self.Liref[o:N, o:o1])))
        return {"err_factor": e_fac, "err_linv": e_inv}

    def _check_back(self, ws, r):
        offs, N = self.offs, self.N
        e = 0.0
        for k in range(r, self.nblk, self.P):
            o, o1 = offs[k], offs[k + 1]
            e = max(e, self._rel(torch.tril(ws.Lc[o:N, ws.col(k)]), torch.tril(self.Kiref[o:N, o:o1])))
        return {"err_kinv": e}

    # ---- the self-consistent timeline ------------------------------------------------------------------------------------------------
    def model_ready(self, rate, chain_us=1200.0, tail_us=600.0):
        """First guess of when each head / tail can be sent: the chain of DESIGN section 7 (diagonal update + panel + head solve + copy
        + gate per step, then the message)."""
        N, nb, offs, nblk = self.N, self.nb, self.offs, self.nblk
        per_us = (lambda b: 0.0) if rate <= 0 else (lambda b: b / (rate * 1e3))
        ready, t_head, t_tail = {}, 0.0, 0.0
        for k in range(nblk):
            nbk = offs[k + 1] - offs[k]
            ready[(k, 0)] = t_head + chain_us
            c0, c1 = self.msgs[k][0]
            t_head = ready[(k, 0)] + per_us(8.0 * nbk * (c1 - c0 + nbk))
            for m, (c0, c1) in enumerate(self.msgs[k][1:], 1):
                ready[(k, m)] = max(t_head, t_tail) + (tail_us if m == 1 else 0.0)
                t_tail = t_head = ready[(k, m)] + per_us(8.0 * nbk * (c1 - c0))
        return ready

    def zero_ready(self):
        return {(k, m): 0.0 for k in range(self.nblk) for m in range(len(self.msgs[k]))}

    def converge(self, rate, ranks, sweeps, check=False, ready=None, verbose=True, tol_us=300.0):
        ready = dict(ready) if ready is not None else self.model_ready(rate)
        res, hist = {}, []
        for sweep in range(sweeps):
            delta = 0.0
            for r in ranks:
                o = self.run(r, rate, ready)
                for key, v in o["ready"].items():
                    delta = max(delta, abs(v - ready[key]))
                    ready[key] = v
                res[r] = o
            tot = {r: sum(res[r][q] for q in ("build_ms", "ff_ms", "vec_ms", "back_ms", "grad_ms")) for r in res}
            hist.append((delta, max(tot.values())))
            if verbose:
                print(f"    sweep {sweep}: ready times moved by up to {delta / 1e3:8.3f} ms; slowest rank {max(tot.values()):9.2f} ms "
                      f"(ff {max(res[r]['ff_ms'] for r in res):8.2f}, back {max(res[r]['back_ms'] for r in res):8.2f})", flush=True)
            if len(ranks) < self.P or delta < max(tol_us, 0.004 * max(ready.values())):
                break
        if check:  # one more pass for the comparisons with the single-GPU result: its TIMES are discarded (the checks wait for the device
            for r in ranks:  # between the stages), only the errors and the statuses are kept
                o = self.run(r, rate, ready, check=True)
                for key in ("err_factor", "err_linv", "err_kinv"):
                    res[r][key] = o.get(key, float("nan"))
                res[r]["status_ff"] = max(res[r]["status_ff"], o["status_ff"])
                res[r]["status_back"] = max(res[r]["status_back"], o["status_back"])
        return res, ready, hist


def summarise(rp, rate, res, hist):
    P = rp.P
    rows = []
    for r in sorted(res):
        o = res[r]
        last = float(o["end"].max()) / 1e3
        rows.append((r, o["build_ms"], o["ff_ms"], o["vec_ms"], o["back_ms"], o["grad_ms"], last, o["status_ff"], o["status_back"]))
    print(f"  rate {('unthrottled' if rate <= 0 else f'{rate:g} GB/s'):>12s}   rank   build      ff  z/alpha    back    grad    total | last message at   status")
    for r, b, f, v, bk, g, last, s1, s2 in rows:
        print(f"  {'':17s}{r:6d} {b:7.2f} {f:7.2f} {v:8.2f} {bk:7.2f} {g:7.2f} {b + f + v + bk + g:8.2f} | {last:10.2f} ms    {s1:#x} {s2:#x}")
    chain = [v for o in res.values() for v in o.get("chain_us", {}).values()]
    lag = [v for o in res.values() for v in o.get("tail_chain_us", {}).values()]
    unp = [o["unpacked"][k, m] - o["end"][k, m] for o in res.values() for k in range(rp.nblk) for m in range(1, rp.M) if o["unpacked"][k, m] > 0]
    if chain and lag and unp:
        print(f"  {'':17s}   chain per step (head k-1 unpacked -> own head k ready to send): mean {np.mean(chain) / 1e3:.3f} ms, "
              f"p10 {np.percentile(chain, 10) / 1e3:.3f}, p90 {np.percentile(chain, 90) / 1e3:.3f}, max {np.max(chain) / 1e3:.3f}; "
              f"tail chain (first piece of row k-1 unpacked -> own first piece ready): mean {np.mean(lag) / 1e3:.3f} ms (max {np.max(lag) / 1e3:.3f}); "
              f"unpacking a piece {np.mean(unp) / 1e3:.3f} ms (max {np.max(unp) / 1e3:.3f})")
    ff, back = max(x[2] for x in rows), max(x[4] for x in rows)
    small = max(x[1] + x[3] + x[5] for x in rows)
    total = ff + back + small  # the ranks meet at the all-reduces of z / alpha and of the gradient
    n3 = rp.N ** 3
    print(f"  {'':17s}   max {'':7s} {ff:7.2f} {'':8s} {back:7.2f} {'':7s} {total:8.2f} ms per evaluation on {P} GPUs = {1e3 / total:7.3f} evals/s, "
          f"{n3 / total / 1e9 / P:5.1f} TFLOP/s per GPU; sweeps {len(hist)}, last move {hist[-1][0] / 1e3:.3f} ms", flush=True)
    return {"rate_gbs": rate, "ff_ms": ff, "back_ms": back, "small_ms": small, "total_ms": total, "evals_per_s": 1e3 / total,
            "sweeps": len(hist), "last_move_ms": hist[-1][0] / 1e3,
            "chain_ms": None if not chain else {"mean": float(np.mean(chain)) / 1e3, "p90": float(np.percentile(chain, 90)) / 1e3,
                                                "max": float(np.max(chain)) / 1e3},
            "ranks": [{"rank": r, "build_ms": b, "ff_ms": f, "vec_ms": v, "back_ms": bk, "grad_ms": g, "last_message_ms": last,
                       "status": [s1, s2]} for r, b, f, v, bk, g, last, s1, s2 in rows]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, help="C2 or C5 (BASELINE.json); or --n / --d for random points")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--P", type=int, default=8)
    ap.add_argument("--nb", type=int, default=1024)
    ap.add_argument("--ranks", default=None, help="comma list (default: all; a subset is replayed against the model's ready times)")
    ap.add_argument("--rates", default="0,400,150,70,50", help="GB/s, descending (0 = unthrottled, every message ready at once: the work bound)")
    ap.add_argument("--sweeps", type=int, default=8)
    ap.add_argument("--wgs", type=int, default=16, help="work-groups of the replayed message's kernel (32 at >= 300 GB/s)")
    ap.add_argument("--threads", type=int, default=512)
    ap.add_argument("--push", action="store_true", help="the push transport's owner side: no packing copy, the message's copies read the factor in place")
    ap.add_argument("--check", action="store_true", help="compare the rank's block rows / column blocks with the single-GPU result")
    ap.add_argument("--json", default=None)
    ap.add_argument("--trace-rank", type=int, default=None, help="per-task trace of this rank's factor + forward list at the last rate")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n = args.n or {"C2": 20000, "C5": 60000}[args.config]
    U, w, sf2, tau, r = make_inputs(args.config, n, args.d, dev)
    rp = Replay(dev, U, w, sf2, tau, r, args.nb, args.P, args.wgs, args.threads, keep_kinv=args.check)
    rp.push = args.push
    ranks = [int(x) for x in args.ranks.split(",")] if args.ranks else list(range(args.P))
    print(f"virtual-rank replay: N = {n}, d = {U.shape[1]}, P = {args.P}, nb = {args.nb}, {rp.nblk} block rows; library {rp.ctx.lib.gpp_version().decode()}")
    print(f"  single-GPU factor + inverse (the reference the other ranks' block rows are taken from): {rp.ref_factor_inverse_ms:.1f} ms; "
          f"tail pieces of {int(rp.ctx.lib.gpp_shard_piece_cols()) or n} columns (up to {rp.M - 1} per block row); "
          f"GPP_SHARD_FILL = {os.environ.get('GPP_SHARD_FILL', 'default (one filler work-group per panel CU)')}; the factor's mirror "
          f"{'beside' if sharded._mirror_beside(args.P) else 'behind'} the list"
          + ("; own messages as the push transport sends them (no packing copy)" if args.push else ""), flush=True)
    out = {"N": n, "d": int(U.shape[1]), "P": args.P, "nb": args.nb, "library": rp.ctx.lib.gpp_version().decode(), "push": bool(args.push), "rates": []}
    t0 = time.perf_counter()
    zero = rp.zero_ready()
    for r_ in ranks:  # (the host plans each rank's two lists on their first use — seconds at N = 60 000 — with the device idle)
        rp.run(r_, 0.0, zero)
    print(f"  warm-up pass over {len(ranks)} rank(s) (plans, allocations): {time.perf_counter() - t0:.1f} s", flush=True)
    ready = None
    rates = [float(x) for x in args.rates.split(",")]
    for rate in rates:
        rp.wgs = max(args.wgs, 32) if (rate <= 0 or rate >= 300) else args.wgs
        if rate <= 0:  # the work bound: every block row of another rank is there when the list starts asking for it
            res, _, hist = rp.converge(rate, ranks, 1, check=args.check, ready=zero)
        else:
            res, ready, hist = rp.converge(rate, ranks, args.sweeps, check=args.check, ready=ready)
        rec = summarise(rp, rate, res, hist)
        if args.check:
            errs = {q: max(res[r].get(q, 0.0) for r in res) for q in ("err_factor", "err_linv", "err_kinv")}
            print(f"  {'':17s}   vs the single-GPU result: factor {errs['err_factor']:.2e}, L^-1 {errs['err_linv']:.2e}, Ky^-1 {errs['err_kinv']:.2e}")
            rec["errors"] = errs
        out["rates"].append(rec)
        if args.trace_rank is not None and rate > 0 and rate == rates[-1]:
            print(f"  trace of rank {args.trace_rank}'s factor + forward list at {rate:g} GB/s:")
            rp.trace(args.trace_rank, rate, ready)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
