"""Cooperative panel (gpp_panel_potrf_inv): correctness against numpy and time of potrf + trtri at the sizes one launch covers.
Run once with GPP_COOP_PANEL=0 and once without to compare with the chain of leaf-step launches.  Dev tool."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context, square_buffer

ctx = get_context("cuda:0")
print("GPP_COOP_PANEL =", os.environ.get("GPP_COOP_PANEL", "(default: on)"))
for N in [int(a) for a in sys.argv[1:]] or [256, 512, 640, 1024]:
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, 6))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    K = np.exp(-0.3 * d2) + 1e-3 * np.eye(N)
    A, Li, T = (square_buffer(N, "cuda") for _ in range(3))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    Kd = torch.tensor(np.triu(K) + np.tril(np.full((N, N), np.nan), -1), device="cuda")
    ts = []
    for rep in range(6):
        A.copy_(Kd); Li.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.potrf(A, Li, info, T); ctx.trtri(A, Li, T); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    assert int(info.item()) == 0, int(info.item())
    Lref = np.linalg.cholesky(K)
    Ah = A.cpu().numpy()
    eU = np.abs(np.triu(Ah) - Lref.T).max()
    untouched = bool(np.isnan(Ah[np.tril_indices(N, -1)]).all())
    full = Li.cpu().numpy()
    eI = np.abs(np.tril(full) @ Lref - np.eye(N)).max()
    mirror = bool(np.array_equal(np.triu(full, 1), np.tril(full, -1).T))
    print(f"N={N}: potrf+trtri {min(ts[2:])*1e3:.0f} us   |U-ref|={eU:.1e} |Linv L - I|={eI:.1e} mirror={mirror} lower untouched={untouched}",
          flush=True)
