"""Per-task trace of the sharded ticket lists at one rank (dev): python tools/attic/dev/shard_trace.py N [which]   which = back (default) | fwd"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context
from gpplus_amd import sharded
import dag_check

N = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "back"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29599")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=0, world_size=1)
g = torch.Generator().manual_seed(0)
D = 8
U = torch.rand(N, D, generator=g, dtype=torch.float64).to(dev) * 4.0
w = torch.full((D,), 0.1, dtype=torch.float64, device=dev); sf2 = torch.tensor([0.85], dtype=torch.float64, device=dev)
tau = torch.tensor([2.5e-3], dtype=torch.float64, device=dev)
ctx = get_context(dev); comm = sharded._Comm(None); ws = sharded._workspace(ctx, N, 1024, 0, 1)
ws.r.zero_()
def run(trace_fwd=False, trace_back=False):
    info = sharded._factor_list(ctx, comm, ws, U, w, sf2, tau, None, 0, 0, 0.0)
    assert info == 0, info
    torch.cuda.synchronize()
    if trace_fwd:
        dag_check.trace_report(ctx, N)
    sharded._vectors(ctx, comm, ws, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sharded._backward(ctx, comm, ws); e1.record(); torch.cuda.synchronize()
    if trace_back:
        print(f"back-substitution list: {e0.elapsed_time(e1):.2f} ms")
        dag_check.trace_report(ctx, N)
run(); run()
if which == "fwd":
    info = sharded._factor_list(ctx, comm, ws, U, w, sf2, tau, None, 0, 0, 0.0); torch.cuda.synchronize()
    assert ctx.lib.gpp_debug_dag_trace(ctx.h, 1) == 0
    run(trace_fwd=True)
else:
    assert ctx.lib.gpp_debug_dag_trace(ctx.h, 1) == 0   # (the most recent plan is the back-substitution's)
    run(trace_back=True)
dist.destroy_process_group()
