"""One trailing update A[nb:, nb:](upper) -= T[:nb, nb:]^T T[:nb, nb:] (K = nb = 1024) run (a) by the static-schedule executor's
kernel ALONE — a wait-free task list on the 224 throughput CUs, 448 persistent work-groups (gpp_debug_exec_update) — and (b) as
ONE launch of gpp_gemm_f64 on all 256 CUs.  The executor's launch has no gate / panel / counter here, so it can run under
`rocprofv3 --pmc` (which serialises dispatches): MFMA busy and HBM-side bytes of the kernel that carries the scheduled steps.
usage: python tools/exec_update_probe.py [N] [reps]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nb = 1024
ctx = get_context("cuda:0")
lib = ctx.lib
lib.gpp_debug_exec_update.restype = ctypes.c_int
lib.gpp_debug_exec_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_int64, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
A, T = square_buffer(N, "cuda"), square_buffer(N, "cuda")
T.normal_(generator=g); T.mul_(1e-3)
M = N - nb
flops = M * M * nb  # upper triangle: 2 * M^2 / 2 * nb
def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def run_exec(k):
    ctx._stream()
    rc = lib.gpp_debug_exec_update(ctx.h, A.data_ptr(), A.stride(0), T.data_ptr(), T.stride(0), N, nb, k)
    assert rc == 0, rc


def run_gemm(k):
    for _ in range(k):
        ctx.gemm(1, 0, M, M, nb, -1.0, T[:nb, nb:], T[:nb, nb:], 1.0, A[nb:, nb:], c_tri=2)


for name, fn in (("executor, 448 work-groups on 224 CUs", run_exec), ("one launch of gpp_gemm_f64, 256 CUs", run_gemm)):
    A.zero_(); fn(1)
    chk = float(torch.triu(A[nb:, nb:]).sum())
    # (the executor's debug entry builds and uploads its task list inside the call: per-launch time from the difference of 1 + reps
    #  and 1 launches)
    per = []
    for _ in range(3):
        t1 = timed(lambda: fn(1)); tk = timed(lambda: fn(1 + reps))
        per.append((tk - t1) / reps)
    ms = sorted(per)[1]
    print(f"N={N} K={nb}: {name}: {ms:7.3f} ms per launch = {flops / ms / 1e9:5.1f} TFLOP/s   (checksum of one update {chk:.6e})", flush=True)
