"""Compare the intermediate matrices of the sharded evaluation (1 rank) with the single-GPU path.  Dev tool."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29599")
dist.init_process_group("gloo", rank=0, world_size=1)
from gpplus_amd.backend import get_context
from gpplus_amd import sharded, linalg
N, D, nb = int(sys.argv[1]), 5, int(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
U = torch.rand(N, D, generator=g, dtype=torch.float64).to(dev)
w = torch.full((D,), 2.5, dtype=torch.float64, device=dev); sf2 = torch.tensor([0.8], dtype=torch.float64, device=dev)
tau = torch.tensor([2e-3], dtype=torch.float64, device=dev)
ctx = get_context(dev)
comm = sharded._Comm(None)
ws = sharded._workspace(ctx, N, nb)
info = sharded._factor(ctx, comm, ws, U, w, sf2, tau, None, 0, 0, 0.0)
print("info", info)
ref = linalg.get_workspace(ctx, N, 0)
linalg._factor(ctx, ref, U, w, sf2, tau, None, 0, 0)
ctx.trtri(ref.A, ref.Li, ref.Ki)
torch.cuda.synchronize()
def cmp(name, a, b):
    print("%-28s max|diff| %.3e  (max|ref| %.3e)" % (name, (a - b).abs().max().item(), b.abs().max().item()))
cmp("U (upper)", torch.triu(ws.A), torch.triu(ref.A))
sharded._inverse(ctx, comm, ws)
torch.cuda.synchronize()
cmp("Linv lower", torch.tril(ws.Li), torch.tril(ref.Li))
cmp("Linv mirror", torch.triu(ws.Li, 1), torch.triu(ref.Li, 1))
offs = ws.offs
for c in range(len(offs) - 1):
    for k in range(c, len(offs) - 1):
        d = (ws.Li[offs[k]:offs[k+1], offs[c]:offs[c+1]] - ref.Li[offs[k]:offs[k+1], offs[c]:offs[c+1]])
        if k == c: d = torch.tril(d)
        e = d.abs().max().item()
        if e > 1e-9: print("  block (%d,%d) err %.2e" % (k, c, e))
sharded._lauum_rows(ctx, comm, ws)
ctx.lauum(ref.Li, ref.Ki)
torch.cuda.synchronize()
cmp("Kinv lower", torch.tril(ws.Ki), torch.tril(ref.Ki))
