"""Where does a cooperative panel launch spend its time?  Needs a probe build of the library:
   make -C gp-plus_amd/csrc clean && make -C gp-plus_amd/csrc CXXFLAGS_EXTRA=-DGPP_PANEL_STAMP
Prints, per leaf j: the chain's wait / leaf / publish times and the phases of the strip that feeds leaf j+1.  Dev tool."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd import _lib
from gpplus_amd.backend import get_context, square_buffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = get_context("cuda:0")
rng = np.random.default_rng(N)
X = rng.standard_normal((N, 6))
K = np.exp(-0.3 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)) + 1e-3 * np.eye(N)
A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
info = torch.zeros(1, dtype=torch.int32, device="cuda")
Kd = torch.tensor(K, device="cuda")
for rep in range(4):
    A.copy_(Kd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.potrf(A, Li, info, T); e1.record(); torch.cuda.synchronize()
print("potrf", e0.elapsed_time(e1) * 1e3, "us")
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
lib.gpp_debug_panel_stamps.restype = ctypes.c_int
assert lib.gpp_debug_panel_stamps(buf) == 0
s = np.array(buf[:], dtype=np.float64)
C = N // 128
t0 = s[1]
total = s[4 * (C - 1) + 3] - t0
tick = e0.elapsed_time(e1) * 1e3 / total  # rough: us per tick, assuming the chain spans the launch
print(f"ticks {total:.0f}, ~{tick*1e3:.2f} ns per tick (assuming the chain spans the launch)")
f = lambda a, b: (s[b] - s[a]) * tick
for j in range(C):
    line = f"leaf {j}: wait {f(4*j, 4*j+1):6.1f}  leaf {f(4*j+1, 4*j+2):6.1f}  publish {f(4*j+2, 4*j+3):5.1f}"
    if j < C - 1:
        w = 128 + 8 * j
        line += (f"  | strip(c=j+1,q=0): sees leaf +{(s[w+1]-s[4*j+3])*tick:5.1f}  S mma {f(w+1, w+2):5.1f}  S store+publish {f(w+2, w+3):5.1f}"
                 f"  waits solved {f(w+3, w+4):5.1f}  U {f(w+4, w+5):5.1f}  publish {f(w+5, w+6):5.1f}  -> chain sees +{(s[4*(j+1)+1]-s[w+6])*tick:5.1f}")
    print(line)
