"""Streaming-write and copy rates of this GPU with plain PyTorch kernels (fill_, copy_): the yardstick the HBM-bound N^2
kernels (gpp_cov_tile writes 4 N^2 B, gpp_grad_tiles reads 4 N^2 B) are compared with.  Dev tool."""
import torch
n = 400_000_000  # 3.2 GB of doubles
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print(f"fill_  3.2 GB written          : {ms:.3f} ms = {3.2 / ms:.2f} TB/s")
ms = t(lambda: b.copy_(a));   print(f"copy_  3.2 GB read + 3.2 written: {ms:.3f} ms = {6.4 / ms:.2f} TB/s (both directions)")
ms = t(lambda: a.sum());      print(f"sum    3.2 GB read             : {ms:.3f} ms = {3.2 / ms:.2f} TB/s")
