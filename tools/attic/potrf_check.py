"""Correctness + timing of gpp_potrf_ws against torch.linalg.cholesky at sizes above the bordering range (dev tool).
usage: python tools/attic/potrf_check.py N [N ...]   (env knobs of gpp_api.hip apply: GPP_LOOKAHEAD_NB, GPP_SPLIT_UPD, ...)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

ctx = get_context("cuda:0")
for N in [int(a) for a in sys.argv[1:]] or [12288, 20000]:
    g = torch.Generator(device="cuda").manual_seed(0)
    U = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g)
    w = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
    sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda")
    tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
    A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    times = []
    for rep in range(4):
        ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.potrf(A, Li, info, T); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    assert int(info.item()) == 0
    ms = min(times[1:])
    msg = f"N={N}: potrf {ms:.2f} ms = {N**3/3/ms/1e9:.1f} TFLOP/s"
    if os.environ.get("CHECK", "1") != "0" and N <= 24000:
        K = square_buffer(N, "cuda")
        ctx.kernel_build(U, w, sf2, tau, None, K, uplo=0)
        Lref = torch.linalg.cholesky(K)
        err = float((torch.triu(A) - Lref.T).abs().max() / Lref.abs().max())
        ctx.trtri(A, Li, T)
        v = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
        t = torch.mv(torch.tril(Li), v)
        err2 = float((torch.mv(Lref, t) - v).norm() / v.norm())
        msg += f"  |U-Lref^T|/|L| = {err:.2e}  |L Linv v - v| = {err2:.2e}"
        del K, Lref
    print(msg, flush=True)
    del A, Li, T
    torch.cuda.empty_cache()
