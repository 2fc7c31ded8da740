"""Do replays of the captured L-BFGS objective keep being served by the graph with eager evaluations, device synchronisations and\nother work in between?  (Found: a replayed hipMemsetAsync node wrote garbage after a device synchronisation; with a probe build\n(-DGPP_PANEL_STAMP) the panel flag blocks are dumped at the end.)  Dev tool."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from gpplus_amd import settings
from gpplus_amd.models import GP_Plus
from gpplus_amd.optim.mll_scipy import MLLObjective
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(n)
X = rng.uniform(size=(n, 8)); y = np.sin(X @ np.arange(1, 9) / 3.0) + 0.01 * rng.standard_normal(n); y = (y - y.mean()) / y.std()
m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device=torch.device("cuda:0"))
obj = MLLObjective(m.train(), True, [0, 0])
x0 = obj.pack_parameters()
obj.fun(x0)
g = obj._graph
with settings.graphed_objective(False):
    eager = MLLObjective(m, True, [0, 0])
def status():
    return g.out_host.numpy()[-1]
def run(label, between):
    bad = 0
    for i in range(40):
        between(i)
        r = g.evaluate(x0 + 1e-3 * (i % 7))
        bad += r is None
    print(f"{label}: {bad}/40 replays failed, last status {status()}", flush=True)
run("replay only", lambda i: None)
run("eager fun between", lambda i: eager.fun(x0 + 1e-3 * (i % 7)))
run("eager fun + device sync between", lambda i: (eager.fun(x0 + 1e-3 * (i % 7)), torch.cuda.synchronize()))
run("replay only again", lambda i: None)



from gpplus_amd.backend import get_context, square_buffer
ctx = get_context("cuda:0")
A, Li = square_buffer(n, "cuda"), square_buffer(n, "cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
K = torch.eye(n, dtype=torch.float64, device="cuda") * 2
def pot(i):
    A.copy_(K); ctx.potrf(A, Li, info)

run("replay only again", lambda i: None)
import ctypes
from gpplus_amd import _lib
lib = _lib.load()
if hasattr(lib, "gpp_debug_panel_flags"):
    per = 4608 // 4
    buf = (ctypes.c_int * (8 * per))(); nxt = ctypes.c_int()
    lib.gpp_debug_panel_flags.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    print("rc", lib.gpp_debug_panel_flags(ctx.h, buf, 8 * per, ctypes.byref(nxt)), "next slot", nxt.value)
    a = np.array(buf[:]).reshape(8, per)
    PF = 2 + 64 + 1024
    for sl in range(8):
        print("slot", sl, "abort", a[sl, 0], "leaf", a[sl, 2:7], "diag", a[sl, 34:39], "found at start: abort/leaf0/leafC-1", a[sl, PF:PF + 3],
              "launches since memset", a[sl, PF + 3], "gave up: wg/flag/target/value", a[sl, PF + 8:PF + 12])
