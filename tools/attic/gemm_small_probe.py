"""Time small TN products through gpp_gemm with a forced work-group tile (GPP_GEMM_TILE=tm,tn).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
def t(M, N, K, a_mask=0, khi=0, beta=0.0, reps=50, inplace=False):
    A = torch.randn(K, max(M, 16), dtype=torch.float64, device="cuda")[:, :M]
    B = torch.randn(K, max(N, 16), dtype=torch.float64, device="cuda")[:, :N]
    C = B if inplace else torch.zeros(M, max(N, 16), dtype=torch.float64, device="cuda")[:, :N]
    for _ in range(5): ctx.gemm(1, 0, M, N, K, 1.0, A, B, beta, C, a_mask=a_mask, khi_mode=khi)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ctx.gemm(1, 0, M, N, K, 1.0, A, B, beta, C, a_mask=a_mask, khi_mode=khi)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("tile", os.environ.get("GPP_GEMM_TILE"))
for (M, N, K) in [(128, 128, 128), (128, 512, 128), (128, 512, 64), (128, 512, 16), (128, 512, 256), (128, 2048, 128)]:
    print("M=%4d N=%5d K=%4d : plain %6.1f us | a_mask=1 %6.1f us | a_mask=1,in place %6.1f us | beta=1 %6.1f us" %
          (M, N, K, t(M, N, K), t(M, N, K, a_mask=1), t(M, N, K, a_mask=1, inplace=(M == K)), t(M, N, K, beta=1.0)))
