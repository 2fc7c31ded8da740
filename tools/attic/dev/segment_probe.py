"""Does a graph capture crash while the autograd node of ANOTHER graph's replay is alive?  usage: segment_probe.py VARIANT"""
import faulthandler, os, sys
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.graphed import GraphedSegment
var = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.nn.Parameter(torch.randn(8, dtype=torch.float64, device=dev))
b = torch.nn.Parameter(torch.randn(3, dtype=torch.float64, device=dev))
if os.environ.get("COPY_NODE"):
    idx = torch.tensor([0, 2, 4], device=dev)
    def f1():
        w = torch.zeros(8, dtype=torch.float64, device=dev).index_add(0, idx, b * b)
        return (a.exp().clone() * 2.0 + w, b.sin().contiguous().clone())
elif os.environ.get("VIEW_OUT"):
    f1 = lambda: ((a.exp() * 2.0), b[:1].expand(4096))
else:
    f1 = lambda: ((a.exp() * 2.0), b.sin())
f2 = lambda: ((a * a).sum().reshape(1) + b.sum(),)
if os.environ.get("DISJOINT"):
    f1 = lambda: (a.exp(),)
    f2 = lambda: (b.exp(),)
if os.environ.get("OVERLAP"):
    f1 = lambda: (a.exp(),)
    f2 = lambda: (b.exp() * a.sum(),)
if os.environ.get("SCALAR"):
    a = torch.nn.Parameter(torch.zeros((), dtype=torch.float64, device=dev))
    b = torch.nn.Parameter(torch.zeros(1, dtype=torch.float64, device=dev))
    f1 = lambda: (a.exp().reshape(1),)
    f2 = lambda: (b.exp().reshape(-1),)
s1 = GraphedSegment(f1, [a, b], dev)
print("seg1 built", flush=True)
if var == "live_node":
    o = s1()
elif var == "live_nograd":
    with torch.no_grad():
        o = s1()
elif var == "replay_only":
    s1.fwd.replay(); o = None
elif var == "backward_done":
    o = s1(); (o[0].sum() + o[1].sum()).backward(); o = None
elif var == "none":
    o = None
elif var == "plain_live_node":   # a live node of an ordinary custom Function (no graph inside)
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x): return x * 2
        @staticmethod
        def backward(ctx, g): return g * 2
    o = F.apply(a)
elif var == "plain_builtin_node":
    o = a.exp()
print("state ready:", var, flush=True)
s2 = GraphedSegment(f2, [a, b], dev)
print("seg2 built", flush=True)
r = s2()[0]
r.backward()
print("ok", float(r), a.grad.norm().item(), flush=True)
