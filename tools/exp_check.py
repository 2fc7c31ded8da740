"""Accuracy of the covariance kernels' exp for non-positive arguments (gpp_exp_nonpos, csrc/gpp_internal.h) through the C ABI:
one row of gpp_cross_kernel with D = 1, w = 1, sf2 = 1 is exp(-(u_j)^2); compared with numpy on the identical fp64 argument.
usage: python tools/attic/exp_check.py [n]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpplus_amd.backend import get_context


def max_rel_err(n=2_000_000, hi=745.0, seed=0):
    ctx = get_context("cuda:0")
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(0, hi, n), rng.uniform(0, 2.0, n), np.linspace(0, hi, 4097), [0.0]])
    u = np.sqrt(x)
    arg = u * u  # what the kernel forms: fma(df, df, 0) with df = 0 - u_j
    ref = np.exp(-arg)
    Ua = torch.zeros(1, 1, dtype=torch.float64, device="cuda")
    Ub = torch.as_tensor(u, device="cuda").reshape(-1, 1).contiguous()
    w = torch.ones(1, dtype=torch.float64, device="cuda")
    sf2 = torch.ones(1, dtype=torch.float64, device="cuda")
    m = Ub.shape[0]
    out = torch.empty(1, (m + 15) // 16 * 16, dtype=torch.float64, device="cuda")[:, :m]
    ctx.cross_kernel(Ua, Ub, w, sf2, out)
    got = out.cpu().numpy()[0]
    normal = ref > 2.3e-308
    rel = np.abs(got[normal] - ref[normal]) / ref[normal]
    ulp = np.abs(got[normal] - ref[normal]) / np.spacing(ref[normal])
    den = np.abs(got[~normal] - ref[~normal]).max() if (~normal).any() else 0.0
    return float(rel.max()), float(ulp.max()), float(den)


if __name__ == "__main__":
    rel, ulp, den = max_rel_err(int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000)
    print(f"gpp_exp_nonpos vs numpy.exp on [-745, 0]: max relative error {rel:.3e} = {ulp:.2f} ulp (normal results); "
          f"max absolute error among denormal results {den:.3e}")
