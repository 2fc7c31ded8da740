"""Stress of the cooperative panel kernel: many sizes, back to back, beside traffic on other streams; every factor checked against
the residual ||U^T U - K|| and the inverse against ||Linv L - I||, the status word must stay 0, and a repeated size must reproduce
its factor bit for bit.  Also the look-ahead (panels on the CU-masked stream) at a few sizes, repeated.  Dev tool:
python tools/stress_panel.py [seconds]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = get_context("cuda:0")
rng = np.random.default_rng(0)
info = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
noise = torch.empty(32 << 20, dtype=torch.float64, device="cuda")
big = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
t_end = time.time() + budget
count, worst_u, worst_i, seen = 0, 0.0, 0.0, {}
while time.time() < t_end:
    n = int(rng.integers(257, 2049))
    if count % 5 == 0 and seen:
        n = list(seen)[int(rng.integers(0, len(seen)))]
    g = torch.Generator(device="cuda").manual_seed(n)
    X = torch.randn(n, 4, dtype=torch.float64, device="cuda", generator=g)
    K = torch.exp(-0.4 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(n, dtype=torch.float64, device="cuda")
    A, Li, T = square_buffer(n, "cuda"), square_buffer(n, "cuda"), square_buffer(n, "cuda")
    A.copy_(K); Li.zero_()
    mode = count % 3
    if mode == 1:
        with torch.cuda.stream(side):
            noise.fill_(1.0); noise.mul_(1.0001)
    elif mode == 2:
        with torch.cuda.stream(side):
            torch.mm(big, big)
    ctx.potrf(A, Li, info, T)
    ctx.trtri(A, Li, T)
    torch.cuda.synchronize()
    if int(info.item()) != 0:
        print(f"FAIL at iteration {count}: N={n} mode={mode} info={int(info.item())}", flush=True)
        for again in range(5):  # does it persist?
            A.copy_(K); Li.zero_(); ctx.potrf(A, Li, info, T); torch.cuda.synchronize()
            print("   again:", int(info.item()), flush=True)
        fails = globals().get("fails", 0) + 1
        globals()["fails"] = fails
        count += 1
        if fails > 5: break
        continue
    U = torch.triu(A)
    ru = float((U.T @ U - K).abs().max())
    L = torch.tril(Li)
    ri = float((L @ U.T - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max())
    worst_u, worst_i = max(worst_u, ru), max(worst_i, ri)
    assert ru < 1e-11 and ri < 1e-8, (n, ru, ri)
    key = (U.cpu().numpy().tobytes(), L.cpu().numpy().tobytes()) if n in seen or len(seen) < 12 else None
    if n in seen:
        assert key == seen[n], f"N={n}: not bitwise repeatable"
    elif key is not None:
        seen[n] = key
    count += 1
print(f"{count} factorisations of 257..2048 rows: all status 0, worst |U^T U - K| = {worst_u:.1e}, worst |Linv L - I| = {worst_i:.1e}, "
      f"{len(seen)} sizes re-run bitwise equal", flush=True)
la_budget = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # extra seconds of look-ahead repetitions per size
for N in (4097, 6700, 9000, 12288):
    g = torch.Generator(device="cuda").manual_seed(N)
    X = torch.randn(N, 5, dtype=torch.float64, device="cuda", generator=g)
    K = torch.exp(-0.3 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(N, dtype=torch.float64, device="cuda")
    A, Li, T = square_buffer(N, "cuda"), square_buffer(N, "cuda"), square_buffer(N, "cuda")
    ref = None
    for rep in range(6):
        A.copy_(K); Li.zero_()
        if rep % 2:
            with torch.cuda.stream(side):
                noise.fill_(2.0)
        ctx.potrf(A, Li, info, T); ctx.trtri(A, Li, T); torch.cuda.synchronize()
        assert int(info.item()) == 0
        cur = (torch.triu(A).clone(), torch.tril(Li).clone())
        if ref is None:
            ref = cur
            U = cur[0]
            r = float((U.T @ U - K).abs().max())
            assert r < 1e-10, (N, r)
        else:
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), f"N={N} rep {rep}: not bitwise repeatable"
    extra, t1 = 0, time.time() + la_budget
    while time.time() < t1:  # many more repetitions: every run must reproduce the first bit for bit
        A.copy_(K); Li.zero_()
        if extra % 3 == 1:
            with torch.cuda.stream(side):
                noise.fill_(2.0)
        ctx.potrf(A, Li, info, T); ctx.trtri(A, Li, T); torch.cuda.synchronize()
        assert int(info.item()) == 0, (N, extra, int(info.item()))
        assert torch.equal(torch.triu(A), ref[0]) and torch.equal(torch.tril(Li), ref[1]), f"N={N} extra rep {extra}: not bitwise repeatable"
        extra += 1
    print(f"look-ahead N={N}: {6 + extra} runs bitwise equal, residual {r:.1e}", flush=True)
    del A, Li, T, K, X, ref, cur
    torch.cuda.empty_cache()
