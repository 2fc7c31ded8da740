"""Repeatability screen of the sharded evaluation's multi-stream choreography (gp-plus_amd/sharded.py): the same evaluation R
times in one process (a one-rank group; GPP_SHARDED_FORCE_COLLECTIVES=1 and backend nccl exercise the collectives too) — every
kernel is deterministic, so ANY difference between repetitions is a race between the panel / throughput / bulk / collective
streams.  usage: python tools/stress_sharded.py N nb reps [nccl]"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.linalg import KernelSpec, exact_mll
from gpplus_amd import settings

N, nb, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29711")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
g = torch.Generator().manual_seed(5)
D, S, dU = 7, 2, 2
U = torch.rand(N, D, generator=g, dtype=torch.float64)
y = torch.sin(3.0 * U[:, 0]) + U[:, 1] ** 2 + 0.05 * torch.randn(N, generator=g, dtype=torch.float64)
grp = (torch.arange(N) % S).to(torch.int32).to(dev)
first, bad = None, 0
for rep in range(reps):
    Ud = U.to(dev).requires_grad_(True)
    w = torch.full((D,), 2.5, dtype=torch.float64, device=dev).requires_grad_(True)
    sf2 = torch.tensor(0.8, dtype=torch.float64, device=dev).requires_grad_(True)
    tau = torch.tensor([2e-3, 4e-3], dtype=torch.float64, device=dev).requires_grad_(True)
    mean = torch.full((N,), 0.1, dtype=torch.float64, device=dev).requires_grad_(True)
    with settings.sharded_evaluation({"group": None, "nb": nb}):
        mll = exact_mll(Ud, KernelSpec(w=w, sf2=sf2, kind=0, d_split=0), tau, mean, y.to(dev), grp=grp, n_grad_dims=dU)
    mll.backward()
    flat = torch.cat([mll.detach().reshape(1), w.grad, sf2.grad.reshape(1), tau.grad, mean.grad, Ud.grad[:, :dU].reshape(-1)])
    if first is None:
        first = flat.clone()
    elif not torch.equal(flat, first):
        bad += 1
        print(f"rep {rep}: differs from rep 0, max |d| = {(flat - first).abs().max().item():.3e}", flush=True)
from gpplus_amd import sharded as _sh
print(f"ticket lists ran in {_sh.LIST_EVALS} of {reps} evaluations")
print(f"N={N} nb={nb} backend={backend} force={os.environ.get('GPP_SHARDED_FORCE_COLLECTIVES', '0')}: {reps} repetitions, "
      f"{bad} differ from the first (mll = {first[0].item():.9f})")
dist.destroy_process_group()
sys.exit(1 if bad else 0)
