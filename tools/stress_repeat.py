"""Race screen: factor + invert + Ky^-1 of the same matrix many times; every repetition must reproduce the first one
bit for bit (tile products accumulate in a fixed order, so any difference is a missing dependency between the internal
streams of the look-ahead driver).  Dev tool: python tools/stress_repeat.py N reps"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6700
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = get_context("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
U = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g)
w = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda")
tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
A, Li, Ki = (square_buffer(N, "cuda") for _ in range(3))
info = torch.zeros(1, dtype=torch.int32, device="cuda")
ref = None
bad = 0
for r in range(reps):
    Li.zero_(); Ki.zero_()
    ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
    ctx.potrf(A, Li, info, Ki)
    ctx.trtri(A, Li, Ki)
    fac = torch.triu(A).clone(); inv = Li.clone()
    ctx.lauum(Li, Ki)
    cur = (fac, inv, torch.tril(Ki).clone())
    assert int(info.item()) == 0
    if ref is None:
        ref = cur
    else:
        same = all(torch.equal(a, b) for a, b in zip(ref, cur))
        if not same:
            bad += 1
            print("rep", r, "differs:", [float((a - b).abs().max()) for a, b in zip(ref, cur)])
print("N=%d: %d repetitions, %d differ from the first" % (N, reps, bad))
sys.exit(1 if bad else 0)
