"""Per-stage timing of one MLL evaluation at size N (HIP events on torch's current stream). Dev tool."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = get_context("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    U = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
    w = torch.full((D,), float(__import__("os").environ.get("STAGES_W", "0.1")), dtype=torch.float64, device="cuda")
    sf2 = torch.tensor([0.85], dtype=torch.float64, device="cuda")
    tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda")
    r = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    A, Li, T, Ki = (square_buffer(N, "cuda") for _ in range(4))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    z = torch.empty(N, dtype=torch.float64, device="cuda"); al = torch.empty_like(z)
    out3 = torch.empty(3, dtype=torch.float64, device="cuda")
    gw = torch.empty(D, dtype=torch.float64, device="cuda"); gs = torch.empty(1, dtype=torch.float64, device="cuda")
    gt = torch.empty(1, dtype=torch.float64, device="cuda")
    stages = {
        "build": lambda: ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2),
        "potrf": lambda: ctx.potrf(A, Li, info, Ki if len(sys.argv) <= 4 else None),
        "trtri": lambda: ctx.trtri(A, Li, Ki),
        "mll_reduce": lambda: ctx.mll_reduce(A, Li, r, z, out3),
        "alpha": lambda: ctx.alpha(Li, z, al),
        "lauum": lambda: ctx.lauum(Li, Ki),
        "grad": lambda: ctx.grad_reduce(U, w, sf2, None, 1, al, Ki, 0, gw, gs, gt, None),
    }
    only = __import__("os").environ.get("STAGES_ONLY")  # e.g. "build,potrf": profile passes whose kernels belong to one stage
    if only:
        stages = {k: v for k, v in stages.items() if k in only.split(",")}
    flops = {"potrf": N**3 / 3, "trtri": N**3 / 3, "lauum": N**3 / 3}
    tot = {k: [] for k in stages}
    nosync = __import__("os").environ.get("STAGES_NOSYNC", "0") != "0"  # enqueue the whole evaluation, then wait (as the product does)
    for rep in range(reps + 1):
        evs = []
        for k, fn in stages.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            if not nosync: torch.cuda.synchronize()
            evs.append((k, e0, e1))
        torch.cuda.synchronize()
        pause = float(__import__("os").environ.get("STAGES_PAUSE_MS", "0"))
        if pause > 0: time.sleep(pause * 1e-3)  # idle device between evaluations (clock recovery experiment)
        for k, e0, e1 in evs:
            if rep > 0: tot[k].append(e0.elapsed_time(e1))
        assert int(info.item()) == 0, info
    total = 0.0
    for k, v in tot.items():
        ms = float(np.median(v)); total += ms
        extra = f"  {flops[k] / ms / 1e9:8.2f} TFLOP/s" if k in flops else ""
        print(f"{k:12s} {ms:10.3f} ms{extra}")
    if only:
        return
    print("grad checksum", float(gw.sum() + gs.sum() + gt.sum()))
    print(f"{'total':12s} {total:10.3f} ms  -> {1000.0 / total:.3f} evals/s   N^3 rate {N**3 / total / 1e9:.2f} TFLOP/s   mll={out3[2].item():.6f}")

if __name__ == "__main__":
    main()
