"""Values of the reference's OWN ``utils/transforms.py`` (softplus, inv_softplus: utils/transforms.py:19-22) on a grid, produced by
importing that module from /root/reference through a temporary ``gpplus`` alias (it needs only torch).  Run HERE, not on the GPU
box:   python tests/golden/make_ref_transforms.py   ->  tests/golden/ref_transforms.npz  (inputs and outputs only)."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
tmp = tempfile.mkdtemp(prefix="refalias_")
os.symlink("/root/reference", os.path.join(tmp, "gpplus"))
sys.path.insert(0, tmp)
from gpplus.utils.transforms import inv_softplus, softplus  # noqa: E402

x = torch.cat([torch.linspace(-30, 30, 121, dtype=torch.float64), torch.tensor([1e-8, 1e-3, 19.9, 20.0, 20.1, 50.0], dtype=torch.float64)])
pos = torch.cat([torch.logspace(-8, 2, 61, dtype=torch.float64), torch.tensor([0.6931471805599453, 1.0, 20.5], dtype=torch.float64)])
np.savez_compressed(os.path.join(HERE, "ref_transforms.npz"), x=x.numpy(), softplus_x=softplus(x).numpy(), pos=pos.numpy(),
                    inv_softplus_pos=inv_softplus(pos).numpy(), softplus_x32=softplus(x.float()).numpy(),
                    inv_softplus_pos32=inv_softplus(pos.float()).numpy())
print("written", len(x), len(pos))
