"""Per-task timeline of the statically scheduled steps of gpp_potrf_ws (gpp_exec_f64; dev tool).
usage: python tools/exec_trace.py N [reps]      (the env knobs of gpp_api.hip / gpp_plan.hip apply)
Runs the factorisation with the executor's time stamps switched on (3 stamps of the 100 MHz clock per task: fetched, waits over,
done) and prints, per step and task kind, when the tasks ran, how long they took and how long their waits lasted."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = get_context("cuda:0")
lib = ctx.lib
g = torch.Generator(device="cuda").manual_seed(0)
U = torch.randn(N, 8, dtype=torch.float64, device="cuda", generator=g)
w = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda")
tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
info = torch.zeros(1, dtype=torch.int32, device="cuda")


def run():
    ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.potrf(A, Li, info, T); e1.record(); torch.cuda.synchronize()
    assert int(info.item()) == 0, int(info.item())
    return e0.elapsed_time(e1)


for _ in range(2):
    ms = run()
print(f"N={N}: potrf {ms:.2f} ms without stamps")
h = ctx.h
lib.gpp_debug_exec_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
lib.gpp_debug_exec_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gpp_debug_exec_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
i6 = (ctypes.c_int64 * 6)()
assert lib.gpp_debug_exec_info(h, i6) == 0, "no plan: the executor did not run at this size"
nt, noff, K, W, F, nc = list(i6)
assert lib.gpp_debug_exec_trace(h, 1) == 0
for _ in range(reps):
    ms = run()
print(f"potrf {ms:.2f} ms with stamps; plan: {nt} tasks, K={K} W={W} F={F}")
tdt = np.dtype([("group", "<i4"), ("tm", "<i2"), ("tn", "<i2"), ("w0", "<i4"), ("w1", "<i4"), ("v0", "<i4"), ("v1", "<i4"),
                ("i0", "<i4"), ("i1", "<i4")])
tasks = np.zeros(nt, dtype=tdt)
offs = np.zeros(noff, dtype=np.int32)
tr = np.zeros((nt, 3), dtype=np.uint64)
assert lib.gpp_debug_exec_fetch(h, tasks.ctypes.data, offs.ctypes.data, tr.ctypes.data) == 0
worker = np.zeros(nt, dtype=np.int32)
o = list(offs) + [nt]
for i in range(noff):
    worker[o[i]:o[i + 1]] = i
real = tasks["group"] >= 0
t0 = tr[real].min()
us = (tr.astype(np.int64) - int(t0)) / 100.0  # 100 MHz -> us
step, kind = tasks["group"] // 3, tasks["group"] % 3
is_fill = worker >= W
names = {0: "solve", 1: "update", 2: "copy"}
print(f"{'step':>4} {'kind':>7} {'where':>6} {'n':>6} {'first start':>11} {'last end':>9} {'median us':>9} {'p95 us':>8} {'wait sum ms':>11} {'wait max us':>11}")
for k in range(K):
    for kd in (0, 1, 2):
        for fl in (False, True):
            m = real & (step == k) & (kind == kd) & (is_fill == fl)
            if not m.any():
                continue
            d = us[m, 2] - us[m, 1]
            wt = us[m, 1] - us[m, 0]
            print(f"{k:4d} {names[kd]:>7} {'fill' if fl else 'main':>6} {m.sum():6d} {us[m, 1].min() / 1e3:11.3f} {us[m, 2].max() / 1e3:9.3f} "
                  f"{np.median(d):9.1f} {np.percentile(d, 95):8.1f} {wt.sum() / 1e3:11.2f} {wt.max():11.1f}")
mm = real & ~is_fill
busy = np.zeros(W); wait = np.zeros(W); end = np.zeros(W)
np.add.at(busy, worker[mm], us[mm, 2] - us[mm, 1])
np.add.at(wait, worker[mm], us[mm, 1] - us[mm, 0])
np.maximum.at(end, worker[mm], us[mm, 2])
print(f"main workers: busy {busy.mean() / 1e3:.2f} ms (min {busy.min() / 1e3:.2f}, max {busy.max() / 1e3:.2f}), waits {wait.mean() / 1e3:.2f} ms "
      f"(max {wait.max() / 1e3:.2f}), end {end.mean() / 1e3:.2f} ms (min {end.min() / 1e3:.2f}, max {end.max() / 1e3:.2f})")
# update tiles by class: full K tile times of the main workers in the first and the last planned step
for k in (0, K // 2, K - 1):
    m = mm & (step == k) & (kind == 1)
    d = us[m, 2] - us[m, 1]
    print(f"step {k}: update tile median {np.median(d):.1f} us, mean {d.mean():.1f}, min {d.min():.1f}, max {d.max():.1f}")
if os.environ.get("EXEC_TRACE_DUMP"):
    np.savez_compressed(os.environ["EXEC_TRACE_DUMP"], tasks=tasks, offs=offs, us=us, worker=worker)
