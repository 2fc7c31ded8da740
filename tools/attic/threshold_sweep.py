"""Factor + inverse + Ky^-1 at sizes around every dispatch threshold of the drivers (leaf steps / look-ahead / bordering /
block heights / split update), checked by residuals against the input matrix.  Dev tool: python tools/attic/threshold_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
sizes = [127, 128, 129, 255, 257, 1000, 4095, 4096, 4097, 4224, 5000, 6143, 6144, 6145, 7167, 7168, 7169, 7300, 8191, 8193,
         10001, 11263, 11264, 11265, 11400, 12289, 13313, 14337, 16385, 17000]
if len(sys.argv) > 1:
    sizes = [int(a) for a in sys.argv[1:]]
bad = 0
for N in sizes:
    for use_ws in (True, False):
        g = torch.Generator(device="cuda").manual_seed(N)
        U = torch.randn(N, 6, dtype=torch.float64, device="cuda", generator=g)
        w = torch.full((6,), 0.15, dtype=torch.float64, device="cuda")
        sf2 = torch.tensor([0.9], dtype=torch.float64, device="cuda")
        tau = torch.tensor([3e-3], dtype=torch.float64, device="cuda")
        A, Li, Ki, K = (square_buffer(N, "cuda") for _ in range(4))
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ctx.kernel_build(U, w, sf2, tau, None, K, uplo=0)
        A.copy_(torch.triu(K))
        ctx.potrf(A, Li, info, Ki if use_ws else None)
        ctx.trtri(A, Li, Ki)
        assert int(info.item()) == 0
        Uf = torch.triu(A)
        v = torch.randn(N, 4, dtype=torch.float64, device="cuda", generator=g)
        r1 = float(((Uf.T @ (Uf @ v)) - K @ v).norm() / (K @ v).norm())          # U^T U = K
        Linv = torch.tril(Li)
        r2 = float((Linv @ (Uf.T @ v) - v).norm() / v.norm())                      # Linv L = I
        mirror = float((torch.triu(Li, 1) - torch.tril(Li, -1).T).abs().max())
        ctx.lauum(Li, Ki)
        Kinv = torch.tril(Ki) + torch.tril(Ki, -1).T
        r3 = float((K @ (Kinv @ v) - v).norm() / v.norm())                         # K Kinv = I
        ok = r1 < 1e-12 and r2 < 1e-9 and r3 < 1e-7 and mirror == 0.0
        bad += not ok
        print("N=%5d ws=%d  |U'U-K| %.1e  |Linv L - I| %.1e  |K Kinv - I| %.1e  mirror %.1e  %s" % (N, use_ws, r1, r2, r3, mirror, "ok" if ok else "BAD"), flush=True)
        del A, Li, Ki, K, Uf, Linv, Kinv
print("bad:", bad)
sys.exit(1 if bad else 0)
