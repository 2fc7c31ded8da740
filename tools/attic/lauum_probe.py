"""LAUUM launch alone at size N on random data (timing probe for kernel experiments).  Dev tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = get_context("cuda:0")
Li, Ki = square_buffer(N, "cuda"), square_buffer(N, "cuda")
Li.normal_()
for _ in range(2): ctx.lauum(Li, Ki)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(3): ctx.lauum(Li, Ki)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("lauum N=%d: %.3f ms  %.2f TFLOP/s" % (N, ms, N**3 / 3 / ms / 1e9))
