"""One trailing-update launch C(upper) -= A^T A at the sizes of the factorisation's late steps: how the rate depends on the number
of tiles in the launch (isolated launches, synchronised in between).  Dev tool; env knobs of gpp_gemm.hip apply (GPP_STAGGER)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
def t(M, K, reps=6):
    A = torch.randn(K, M, dtype=torch.float64, device="cuda")
    C = square_buffer(M, "cuda"); C.zero_()
    ctx.gemm(1, 0, M, M, K, -1.0, A, A, 1.0, C, c_tri=2); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.gemm(1, 0, M, M, K, -1.0, A, A, 1.0, C, c_tri=2); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    tiles = (M // 128) * (M // 128 + 1) // 2
    return ms, M * M * K / ms / 1e9, tiles
print("GPP_STAGGER =", os.environ.get("GPP_STAGGER", "0"))
for K in (512, 1024):
    for M in (3072, 4096, 6144, 8192, 12288, 17920):
        ms, tf, tiles = t(M, K)
        print("K=%4d M=%5d tiles %5d (%.1f waves of 512): %7.3f ms %5.1f TF" % (K, M, tiles, tiles / 512, ms, tf))
