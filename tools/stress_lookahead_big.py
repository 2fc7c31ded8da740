"""The look-ahead factorisation at the benchmark sizes, repeated: every run must give status 0 and reproduce the first run's factor
and inverse-factor blocks bit for bit (the cooperative panels run beside the masked and unmasked trailing updates here).
usage: python tools/stress_lookahead_big.py [N reps] ...   Dev tool."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

args = [int(a) for a in sys.argv[1:]] or [15000, 200, 20000, 200, 30000, 40]
ctx = get_context("cuda:0")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
# STRESS_BG=n: n copies of 1 GB on a side stream beside every factorisation (memory-saturating traffic: what exposed the missing
# wait behind buffer_wbl2 in round 3; the DAG executor's tasks hand data between work-groups the same way)
BG = int(os.environ.get("STRESS_BG", "0"))
if BG:
    bg_stream = torch.cuda.Stream()
    bg_a = torch.empty(1 << 27, dtype=torch.float64, device="cuda"); bg_b = torch.empty_like(bg_a)
for N, reps in zip(args[0::2], args[1::2]):
    g = torch.Generator(device="cuda").manual_seed(N)
    U = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g)
    w = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
    sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda"); tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
    A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
    ref = None
    t0 = time.time()
    for rep in range(reps):
        ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
        if BG:
            bg_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(bg_stream):
                for _ in range(BG):
                    bg_b.copy_(bg_a)
        ctx.potrf(A, Li, info, T)
        torch.cuda.synchronize()
        assert int(info.item()) == 0, (N, rep, int(info.item()))
        cur = (A.sum().item(), A.abs().max().item(), A[::7, ::5].clone(), Li.diagonal().clone())
        if ref is None:
            ref = cur
        else:
            assert cur[0] == ref[0] and cur[1] == ref[1] and torch.equal(cur[2], ref[2]) and torch.equal(cur[3], ref[3]), (N, rep)
    print(f"N={N}: {reps} factorisations ({(time.time()-t0)/reps*1e3:.1f} ms each incl. build + checks), all status 0 and bitwise equal", flush=True)
    del A, Li, T, U, ref, cur
    torch.cuda.empty_cache()
