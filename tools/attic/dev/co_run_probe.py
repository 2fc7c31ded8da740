"""Do kernels of two PROCESSES run concurrently on one GPU here?  Each of two processes spins one wave for ~T seconds (torch.cuda._sleep)
behind a file barrier; concurrent: both finish after ~T, serialised: the second after ~2T.  Then the same with a spin of many
work-groups (a large elementwise kernel repeated) to see time-slicing.  usage: python tools/attic/dev/co_run_probe.py"""
import os, subprocess, sys, time

CHILD = r'''
import os, sys, time, torch
rank = int(sys.argv[1]); d = sys.argv[2]
torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
if os.environ.get("PROBE_COOP"):
    sys.path.insert(0, os.getcwd())
    from gpplus_amd.backend import get_context, square_buffer
    ctx = get_context("cuda:0")
    n = 8192
    A = square_buffer(n, "cuda"); Li = square_buffer(n, "cuda"); T = square_buffer(n, "cuda")
    A.copy_(torch.eye(n, dtype=torch.float64, device="cuda") * 4 + 0.001)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.potrf(A, Li, info, T); torch.cuda.synchronize()
    print(f"rank {rank}: warm-up factorisation with the cooperative panel done, info {int(info.item())}", flush=True)
open(os.path.join(d, f"ready{rank}"), "w").close()
while not all(os.path.exists(os.path.join(d, f"ready{r}")) for r in (0, 1)):
    time.sleep(0.001)
t0 = time.time()
if os.environ.get("PROBE_COOP") == "2":
    # rank 0: a cooperative panel launch QUEUED behind a spinning kernel on its stream; rank 1: a cooperative panel of its own, timed
    m = 1024
    B = square_buffer(m, "cuda"); Bi = square_buffer(m, "cuda"); Bt = square_buffer(m, "cuda")
    B.copy_(torch.eye(m, dtype=torch.float64, device="cuda") * 4 + 0.001)
    if rank == 0:
        torch.cuda._sleep(int(4e9))
        ctx.potrf(B, Bi, info, Bt)
    else:
        time.sleep(0.3)
        t1 = time.time()
        ctx.potrf(B, Bi, info, Bt); torch.cuda.synchronize()
        print(f"rank 1: its own cooperative panel took {time.time() - t1:.3f} s while rank 0 has one pending behind a spin", flush=True)
else:
    torch.cuda._sleep(int(4e9))
torch.cuda.synchronize()
print(f"rank {rank}: one-wave spin took {time.time() - t0:.2f} s (started at {t0 % 100:.2f})", flush=True)
'''

def main():
    import tempfile
    d = tempfile.mkdtemp()
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(r), d]) for r in (0, 1)]
    for p in ps:
        p.wait()
    print(f"both done after {time.time() - t0:.2f} s")
    # one process alone, for the scale
    d2 = tempfile.mkdtemp()
    open(os.path.join(d2, "ready1"), "w").close()
    subprocess.run([sys.executable, "-c", CHILD, "0", d2])

main()
