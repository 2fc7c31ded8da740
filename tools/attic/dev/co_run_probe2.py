"""While one process's sharded ticket list (rank 1 of 2: its executor spins, waiting for a message that never comes, until
GPP_SHARD_TIMEOUT_MS) holds the GPU, what can ANOTHER process run?  usage: python tools/attic/dev/co_run_probe2.py"""
import os, subprocess, sys, time, tempfile

SPINNER = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from gpplus_amd.backend import get_context
from gpplus_amd.sharded import ShardedWorkspace
d = sys.argv[1]
ctx = get_context("cuda:0")
N, nb = 9000, 1024
ws = ShardedWorkspace(ctx, N, nb, 1, 2)
ws.A.copy_(torch.eye(N, dtype=torch.float64, device="cuda")[:, :N] * 4 + 0.001) if False else None
ws.info.zero_()
torch.cuda.synchronize()
open(os.path.join(d, "ready0"), "w").close()
while not os.path.exists(os.path.join(d, "ready1")):
    time.sleep(0.001)
t0 = time.time()
used = ctx.shard_list_begin(N, nb, 1, 2, ws.A, ws.Kc, ws.Lc, ws.D, ws.W2, ws.info[0:1], int(os.environ.get("SPIN_WORKERS", "224")))
print(f"spinner: list enqueued (used={used}) at {time.time() % 100:.2f}", flush=True)
ctx.shard_list_end()
torch.cuda.synchronize()
print(f"spinner: list over after {time.time() - t0:.2f} s, info {int(ws.info[0].item()):#x}", flush=True)
'''

TIMER = r'''
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from gpplus_amd.backend import get_context, square_buffer
d = sys.argv[1]
ctx = get_context("cuda:0")
n = 9000
A = square_buffer(n, "cuda"); Li = square_buffer(n, "cuda"); T = square_buffer(n, "cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
m = 1024
B = square_buffer(m, "cuda"); Bi = square_buffer(m, "cuda"); Bt = square_buffer(m, "cuda")
def fill(X, k):
    X.zero_(); X.diagonal().fill_(4.0); X[:k, :k].add_(0.001)
fill(A, n); ctx.potrf(A, Li, info, T); fill(B, m); ctx.potrf(B, Bi, info, Bt); torch.cuda.synchronize()
open(os.path.join(d, "ready1"), "w").close()
while not os.path.exists(os.path.join(d, "ready0")):
    time.sleep(0.001)
time.sleep(1.0)
x = torch.zeros(1 << 24, device="cuda")
for name, fn in (("small torch kernel", lambda: x.add_(1.0)),
                 ("1024-row factorisation (cooperative panel)", lambda: (fill(B, m), ctx.potrf(B, Bi, info, Bt))),
                 ("9000-row factorisation (executor list + panels)", lambda: (fill(A, n), ctx.potrf(A, Li, info, T)))):
    t1 = time.time(); fn(); torch.cuda.synchronize()
    print(f"timer: {name}: {time.time() - t1:.3f} s (at {time.time() % 100:.2f}), info {int(info.item()):#x}", flush=True)
'''

d = tempfile.mkdtemp()
ps = [subprocess.Popen([sys.executable, "-c", SPINNER, d]), subprocess.Popen([sys.executable, "-c", TIMER, d])]
for p in ps:
    p.wait()
