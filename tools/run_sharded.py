"""Time the sharded single evaluation (gp-plus_amd/sharded.py).  Launch with torch.distributed.run, one rank per GPU
(backend nccl = RCCL); with a single process it runs as a 1-rank group over gloo (algorithm overhead vs the single-GPU
path).   usage: [torchrun ...] tools/run_sharded.py N D [nb] [reps]"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context
from gpplus_amd import sharded

def main():
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29598")
    one_each = torch.cuda.device_count() >= world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if one_each else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if (one_each and world > 1) else "gloo", rank=rank, world_size=world)
    N, D = int(sys.argv[1]), int(sys.argv[2])
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    g = torch.Generator().manual_seed(0)
    U = torch.rand(N, D, generator=g, dtype=torch.float64).to(dev) * 4.0
    y = (torch.sin(U[:, 0]) + 0.1 * torch.randn(N, generator=g, dtype=torch.float64).to(dev))
    w = torch.full((D,), 0.1, dtype=torch.float64, device=dev); sf2 = torch.tensor([0.85], dtype=torch.float64, device=dev)
    tau = torch.tensor([2.5e-3], dtype=torch.float64, device=dev); mean = torch.zeros(N, dtype=torch.float64, device=dev)
    ctx = get_context(dev); comm = sharded._Comm(None); ws = sharded._workspace(ctx, N, nb, comm.rank, comm.world)
    if rank == 0:
        print('matrices per rank: %.2f GB (a full N x N fp64 matrix: %.2f GB)' % (ws.nbytes() / 1e9, 8e-9 * N * N), flush=True)
    def sync():
        torch.cuda.synchronize(); dist.barrier()
    for rep in range(reps + 1):
        sync(); t0 = time.perf_counter()
        # (round 5: factorisation + forward sweep as ONE ticket list per rank where it applies: "forward" is then 0)
        info = sharded._factor_list(ctx, comm, ws, U, w, sf2, tau, None, 0, 0, 0.0) if sharded._USE_LIST else None
        swept = info is not None
        if info is None:
            info = sharded._factor(ctx, comm, ws, U, w, sf2, tau, None, 0, 0, 0.0)
        sync(); t1 = time.perf_counter()
        if not swept:
            sharded._forward(ctx, comm, ws)
        sync(); t2 = time.perf_counter()
        torch.sub(y, mean, out=ws.r); sharded._vectors(ctx, comm, ws, True); sync(); t3 = time.perf_counter()
        sharded._backward(ctx, comm, ws)  # this rank's column blocks of Ky^-1 by back-substitution: nothing travels
        sync(); t4 = time.perf_counter()
        flat = torch.zeros(D + 2, dtype=torch.float64, device=dev)
        ctx.grad_reduce_cols(U, w, sf2, None, 1, ws.alpha, ws.Lc, 0, nb, comm.rank, comm.world, flat[:D], flat[D:D + 1], flat[D + 1:], None, compact=True)
        comm.allreduce(flat); sync(); t5 = time.perf_counter()
        if rank == 0 and rep > 0:
            tot = t5 - t0
            print("N=%d D=%d nb=%d ranks=%d backend=%s info=%d: factor %.1f ms, forward sweeps %.1f ms, z/alpha %.1f ms, back-substitution %.1f ms, grad %.1f ms; total %.1f ms -> %.3f evals/s, %.1f TFLOP/s (N^3) per GPU, mll=%.6f"
                  % (N, D, nb, world, dist.get_backend(), info, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * tot, 1 / tot,
                     N ** 3 / tot / 1e12 / world, ws.out3[2].item()), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
