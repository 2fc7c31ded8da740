"""Two (or more) processes on ONE GPU, each factoring in a loop with the cooperative panel kernel on its handle's 32-CU stream:
how often do two panel launches hold part of the same CUs (status 2^30 after the ~1 s time-out), and what does a factorisation cost
beside a neighbour?  usage: python tools/tenant_probe.py [processes] [seconds] [N] [model]   ('model': through GP_Plus, where the host recovers by itself;
dev tool; the child modes are internal)"""
import os, subprocess, sys, time

if len(sys.argv) > 1 and sys.argv[1] == "--child-model":
    # the same through the model API: the host must recover from a time-out by itself (backend.panel_timed_out)
    import warnings
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gpplus_amd.backend import get_context
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.models import GP_Plus
    secs, N = float(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, 6)); y = np.sin(X.sum(1))
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device=torch.device("cuda:0")).train()
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    def ev():
        m.zero_grad(); loss = -mll(m(*m.train_inputs), m.train_targets); loss.backward(); return loss.item()
    ref = ev()
    n = bad = 0; caught = []
    t0 = time.time(); t_end = t0 + secs
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        while time.time() < t_end:
            v = ev(); n += 1
            if abs(v - ref) > 1e-9 * abs(ref): bad += 1
        caught = [str(x.message)[:60] for x in w if "panel" in str(x.message)]
    print(f"pid {os.getpid()}: {n} evaluations of N={N} in {time.time()-t0:.1f} s ({(time.time()-t0)/n*1e3:.2f} ms each), {bad} off, "
          f"panel still on: {get_context('cuda:0').coop_panel}, warnings: {caught}", flush=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gpplus_amd.backend import INFO_PANEL_TIMEOUT, get_context, square_buffer
    secs, N = float(sys.argv[2]), int(sys.argv[3])
    ctx = get_context("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(N)
    X = torch.randn(N, 5, dtype=torch.float64, device="cuda", generator=g)
    K = torch.exp(-0.3 * torch.cdist(X, X) ** 2) + 1e-3 * torch.eye(N, dtype=torch.float64, device="cuda")
    A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    A.copy_(K); ctx.potrf(A, Li, info, T); torch.cuda.synchronize()
    ref = torch.triu(A).clone()
    n = timeouts = wrong = 0
    t_end = time.time() + secs
    t0 = time.time()
    while time.time() < t_end:
        A.copy_(K); ctx.potrf(A, Li, info, T); torch.cuda.synchronize()
        s = int(info.item())
        if s >= INFO_PANEL_TIMEOUT: timeouts += 1
        elif s != 0 or not torch.equal(torch.triu(A), ref): wrong += 1
        n += 1
    print(f"pid {os.getpid()}: {n} factorisations of N={N} in {time.time()-t0:.1f} s ({(time.time()-t0)/n*1e3:.2f} ms each), "
          f"{timeouts} panel time-outs, {wrong} wrong", flush=True)
    sys.exit(0)

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
secs = sys.argv[2] if len(sys.argv) > 2 else "20"
N = sys.argv[3] if len(sys.argv) > 3 else "4096"
mode = "--child-model" if len(sys.argv) > 4 and sys.argv[4] == "model" else "--child"
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), mode, secs, N]) for _ in range(P)]
rc = [p.wait() for p in procs]
print("exit codes", rc)
