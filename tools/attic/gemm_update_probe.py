"""Rank-K update C(upper) = beta*C - A^T A at look-ahead sizes: cost of the beta = 1 epilogue and of K.  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
def t(M, K, beta, reps=3):
    A = torch.randn(K, M, dtype=torch.float64, device="cuda")
    C = square_buffer(M, "cuda"); C.zero_()
    for _ in range(2): ctx.gemm(1, 0, M, M, K, -1.0, A, A, beta, C, c_tri=2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): ctx.gemm(1, 0, M, M, K, -1.0, A, A, beta, C, c_tri=2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, M * M * K / ms / 1e9
for M in (17920, 9984):
    for K in (512, 1024, 2048, 4096):
        a, b = t(M, K, 0.0), t(M, K, 1.0)
        print("M=%5d K=%4d : beta=0 %7.3f ms %5.1f TF | beta=1 %7.3f ms %5.1f TF" % (M, K, a[0], a[1], b[0], b[1]))
