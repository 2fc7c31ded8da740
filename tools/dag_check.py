"""Correctness + timing of the DAG executor's factorisation (+ inverse) against torch.linalg.cholesky, and per-task traces (dev tool).
usage: python tools/dag_check.py N [N ...]      env: TRACE=1 per-kind task statistics of the last run, CHECK=0 timing only,
       REPS=n timed repetitions; the knobs of gpp_api.hip / gpp_dag.hip apply (GPP_DAG_SCHED, GPP_DAG_NB, GPP_DAG_CHAIN_TILE, ...)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer  # noqa: E402

KINDS = ["S", "U", "CP", "XB", "XA", "SH", "UD"]


def trace_report(ctx, N):
    lib = ctx.lib
    info = (ctypes.c_int64 * 10)()
    if lib.gpp_debug_dag_info(ctx.h, info) != 0:
        print("  (no DAG plan on this handle)")
        return
    nt = int(info[0])
    tasks = np.zeros(nt, dtype=np.dtype([("group", "i4"), ("tm", "i2"), ("tn", "i2"), ("w", "i4", 3), ("v", "i4", 3), ("inc", "i4", 2), ("kind", "i4"), ("incv", "i2", 2)]))
    assert tasks.dtype.itemsize == 48
    trace = np.zeros((nt, 4), dtype=np.uint64)
    rc = lib.gpp_debug_dag_fetch(ctx.h, tasks.ctypes.data_as(ctypes.c_void_p), trace.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    t0 = trace[:, 0].astype(np.int64)
    base = t0[t0 > 0].min()
    grab = (trace[:, 0].astype(np.int64) - base) / 100.0  # us
    ready = (trace[:, 1].astype(np.int64) - base) / 100.0
    done = (trace[:, 2].astype(np.int64) - base) / 100.0
    wg = (trace[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    span = done.max()
    print(f"  plan: {nt} tasks, {int(info[2])} blocks, simulated {info[5] / 1000:.2f} ms at {info[6] / 10:.0f} % busy; traced span {span / 1000:.2f} ms")
    nw = int(wg.max()) + 1
    run = done - ready
    wait = ready - grab
    print(f"  workers {nw}: mean busy {run.sum() / nw / 1000:.2f} ms, mean wait {wait.sum() / nw / 1000:.2f} ms, "
          f"mean idle-between {(span * nw - run.sum() - wait.sum()) / nw / 1000:.2f} ms")
    for k, name in enumerate(KINDS):
        m = tasks["kind"] == k
        if m.any():
            print(f"    {name:3s} n={int(m.sum()):6d}  run mean {run[m].mean():7.1f} us (p10 {np.percentile(run[m], 10):6.1f}, p90 {np.percentile(run[m], 90):6.1f})"
                  f"  wait mean {wait[m].mean():7.1f} us  total run {run[m].sum() / nw / 1000:6.2f} ms/worker")
    # utilisation over time: tasks running per 0.5 ms bucket
    nb = int(span // 500) + 1
    util = np.zeros(nb)
    for b in range(nb):
        lo, hi = b * 500.0, (b + 1) * 500.0
        util[b] = np.clip(np.minimum(done, hi) - np.maximum(ready, lo), 0, None).sum() / 500.0
    print("  running tasks per 0.5 ms:", " ".join(f"{u:.0f}" for u in util))
    if os.environ.get("TRACE_DUMP"):
        np.savez(os.environ["TRACE_DUMP"], tasks=tasks, grab=grab, ready=ready, done=done, wg=wg)


def main():
    ctx = get_context("cuda:0")
    reps = int(os.environ.get("REPS", "4"))
    for N in [int(a) for a in sys.argv[1:]] or [4096, 10000]:
        g = torch.Generator(device="cuda").manual_seed(0)
        U = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g)
        w = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
        sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda")
        tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
        A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        times = []
        for rep in range(reps + 1):
            ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
            Li.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.potrf(A, Li, info, T); ctx.trtri(A, Li, T); e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
            st = int(info.item())
            assert st == 0, hex(st)
        ms = min(times[1:])
        msg = f"N={N}: potrf + inverse {ms:.2f} ms = {2 * N**3 / 3 / ms / 1e9:.1f} TFLOP/s"
        if os.environ.get("CHECK", "1") != "0" and N <= 24000:
            K = square_buffer(N, "cuda")
            ctx.kernel_build(U, w, sf2, tau, None, K, uplo=0)
            Lref = torch.linalg.cholesky(K)
            err = float((torch.triu(A) - Lref.T).abs().max() / Lref.abs().max())
            v = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
            t = torch.mv(torch.tril(Li), v)
            err2 = float((torch.mv(Lref, t) - v).norm() / v.norm())
            mir = float((torch.triu(Li, 1) - torch.tril(Li, -1).T).abs().max())
            msg += f"  |U-Lref^T|/|L| = {err:.2e}  |L Linv v - v|/|v| = {err2:.2e}  mirror {mir:.1e}"
            del K, Lref
        print(msg, flush=True)
        if os.environ.get("TRACE"):
            assert ctx.lib.gpp_debug_dag_trace(ctx.h, 1) == 0
            ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
            torch.cuda.synchronize()
            ctx.potrf(A, Li, info, T)
            torch.cuda.synchronize()
            trace_report(ctx, N)
            ctx.lib.gpp_debug_dag_trace(ctx.h, 0)
        del A, Li, T
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
