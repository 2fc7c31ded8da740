"""Same launch as small_update_probe.py with the operands laid out as inside the factorisation at N = 20000: the K x M panel and
the M x M corner are views of N x ld buffers (row stride 160 KB) instead of compact arrays.  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
N = 20000
big = square_buffer(N, "cuda"); big.normal_()
def t(M, K, strided_a, strided_c, reps=6):
    o = N - M
    A = big[o - K:o, o:N] if strided_a else torch.randn(K, M, dtype=torch.float64, device="cuda")
    Cb = big if strided_c else None
    C = big[o:N, o:N] if strided_c else square_buffer(M, "cuda")
    if not strided_c: C.zero_()
    ctx.gemm(1, 0, M, M, K, -1e-9, A, A, 1.0, C, c_tri=2); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.gemm(1, 0, M, M, K, -1e-9, A, A, 1.0, C, c_tri=2); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    return ms, M * M * K / ms / 1e9
for K in (512, 1024):
    for M in (4096, 8192, 12288):
        r = [t(M, K, sa, sc) for sa, sc in ((False, False), (True, False), (False, True), (True, True))]
        print("K=%4d M=%5d: compact %5.1f TF | strided A %5.1f | strided C %5.1f | both %5.1f TF (%.3f ms)" % (K, M, r[0][1], r[1][1], r[2][1], r[3][1], r[3][0]))
