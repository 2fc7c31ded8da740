"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Plain-PyTorch fp64 restatement of GP+'s exact-GP hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module, and
only as the checker / reported baseline: nothing under ``gp-plus_amd/`` imports it and the product has no CPU path.

PARITY UNPINNED.  The reference (Bostanabad-Research-Group/GP-Plus @ 2024_08_07) ships no tests, fixtures or stored
notebook outputs for this path, and the arithmetic lives in third-party gpytorch (version not pinned by the
reference; API use implies ~1.6-1.8), which is not installed in this image and cannot be.  The gpytorch pieces are
therefore restated from their published algorithm and marked [3P]; what pins this file instead is listed in
DESIGN.md ("oracle pinning") and exercised by tests/test_oracle.py: closed forms, an independent numpy/scipy
evaluation, the analytic-gradient identity, central finite differences, torch.distributions for the priors, and
reference-generated INPUT fixtures (tests/golden/make_golden.py imports the reference's own data pipeline).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import itertools
import math
import warnings
from typing import Dict, List, Optional, Sequence

import torch

DT = torch.float64


# ---------------------------------------------------------------------------------------------------
# parameter transforms
# ---------------------------------------------------------------------------------------------------
def softplus(x: torch.Tensor) -> torch.Tensor:
    """utils/transforms.py:19 (torch.nn.Softplus, beta=1, threshold=20)."""
    return torch.nn.functional.softplus(x)


def inv_softplus(x: torch.Tensor) -> torch.Tensor:
    """utils/transforms.py:21-22."""
    return x + torch.log(-torch.expm1(-x))


def noise_transform(raw: torch.Tensor, lb: float) -> torch.Tensor:
    """models/gpregression.py:59  GreaterThan(lb, transform=exp): [3P] transform(raw) + lower_bound."""
    return torch.exp(raw) + lb


def rough_lengthscale(raw: torch.Tensor) -> torch.Tensor:
    """models/gp_plus.py:248-253  Positive(transform = 2^(-1/2) * 10^(-x/2))."""
    return 2.0 ** (-0.5) * torch.pow(torch.tensor(10.0, dtype=raw.dtype), -raw / 2)


def exp_lengthscale(raw: torch.Tensor) -> torch.Tensor:
    """models/gp_plus.py:243-247 / models/gpregression.py:93-95  Positive(transform=exp)."""
    return torch.exp(raw)


# ---------------------------------------------------------------------------------------------------
# priors
# ---------------------------------------------------------------------------------------------------
def log_half_horseshoe_log_prob(raw: torch.Tensor, scale: float, lb: float) -> torch.Tensor:
    """priors/horseshoe.py:60-66: log(log(1 + 3 (scale / (lb + exp(X)))^2)) + X."""
    return torch.log(torch.log(1 + 3 * (scale / (lb + torch.exp(raw))) ** 2)) + raw


def mollified_uniform_log_prob(x: torch.Tensor, a: float, b: float, tail_sigma: float = 0.1) -> torch.Tensor:
    """priors/mollified_uniform.py:67-82."""
    mean, half = (a + b) / 2, (b - a) / 2
    tail = ((x - mean).abs() - half).clamp(min=0)
    log_norm = -math.log(1 + (b - a) / (math.sqrt(2 * math.pi) * tail_sigma))
    return -0.5 * (tail / tail_sigma) ** 2 - math.log(tail_sigma) - 0.5 * math.log(2 * math.pi) + log_norm


def normal_log_prob(x: torch.Tensor, loc: float, scale: float) -> torch.Tensor:
    """[3P] gpytorch NormalPrior == torch.distributions.Normal.log_prob."""
    return -0.5 * ((x - loc) / scale) ** 2 - math.log(scale) - 0.5 * math.log(2 * math.pi)


def lognormal_log_prob(x: torch.Tensor, loc: float, scale: float) -> torch.Tensor:
    """[3P] gpytorch LogNormalPrior == torch.distributions.LogNormal.log_prob (models/gpregression.py:113-115)."""
    lx = torch.log(x)
    return -0.5 * ((lx - loc) / scale) ** 2 - math.log(scale) - 0.5 * math.log(2 * math.pi) - lx


# ---------------------------------------------------------------------------------------------------
# [3P] gpytorch kernels
# ---------------------------------------------------------------------------------------------------
def sq_dist_gpytorch(x1: torch.Tensor, x2: torch.Tensor, x1_eq_x2: bool) -> torch.Tensor:
    """[3P] gpytorch.kernels.kernel.Distance._sq_dist (reached from kernels/Rough_RBF.py:30-32 and RBFKernel):
    centre on x1's mean, one skinny GEMM [-2x, |x|^2, 1] [x, 1, |x|^2]^T, zero the diagonal only when no grad is
    required, clamp at 0."""
    adjustment = x1.mean(-2, keepdim=True)
    x1 = x1 - adjustment
    x2 = x2 - adjustment
    x1_norm = x1.pow(2).sum(dim=-1, keepdim=True)
    x1_pad = torch.ones_like(x1_norm)
    no_grad = not x1.requires_grad and not x2.requires_grad
    if x1_eq_x2 and no_grad:
        x2_norm, x2_pad = x1_norm, x1_pad
    else:
        x2_norm = x2.pow(2).sum(dim=-1, keepdim=True)
        x2_pad = torch.ones_like(x2_norm)
    x1_ = torch.cat([-2.0 * x1, x1_norm, x1_pad], dim=-1)
    x2_ = torch.cat([x2, x2_pad, x2_norm], dim=-1)
    res = x1_.matmul(x2_.transpose(-2, -1))
    if x1_eq_x2 and no_grad:
        res.diagonal(dim1=-2, dim2=-1).fill_(0)
    return res.clamp_min(0)


def rbf_gpytorch(x1: torch.Tensor, x2: torch.Tensor, lengthscale: torch.Tensor) -> torch.Tensor:
    """[3P] gpytorch RBFKernel.forward: exp(-0.5 * || (x1-x2)/l ||^2) via postprocess_rbf (div(-2).exp())."""
    eq = x1.shape == x2.shape and torch.equal(x1, x2)
    return sq_dist_gpytorch(x1 / lengthscale, x2 / lengthscale, eq).div(-2).exp()


def matern_gpytorch(x1: torch.Tensor, x2: torch.Tensor, lengthscale: torch.Tensor, nu: float) -> torch.Tensor:
    """[3P] gpytorch MaternKernel.forward (kernels/matern.py:4-8 fix nu = 1.5 / 2.5): centre on x1's mean, scale by the
    lengthscale, Euclidean distance = sqrt(clamp(sq_dist, 1e-30)), (1 + sqrt(3) d) e^{-sqrt(3) d}  or
    (1 + sqrt(5) d + 5/3 d^2) e^{-sqrt(5) d}."""
    mean = x1.reshape(-1, x1.size(-1)).mean(0)
    x1_ = (x1 - mean) / lengthscale
    x2_ = (x2 - mean) / lengthscale
    eq = x1.shape == x2.shape and torch.equal(x1, x2)
    dist = sq_dist_gpytorch(x1_, x2_, eq).clamp_min(1e-30).sqrt()
    exp_component = torch.exp(-math.sqrt(nu * 2) * dist)
    if nu == 1.5:
        constant_component = (math.sqrt(3) * dist).add(1)
    elif nu == 2.5:
        constant_component = (math.sqrt(5) * dist).add(1).add(5.0 / 3.0 * dist ** 2)
    else:
        raise ValueError(nu)
    return constant_component * exp_component


def rough_rbf_standalone(x1: torch.Tensor, x2: torch.Tensor, lengthscale: torch.Tensor, ard_num_dims: Optional[int] = None,
                         diag: bool = False) -> torch.Tensor:
    """kernels/Rough_RBF.py:18-40, both branches.
    Branch 1 (:19-32; taken when an input requires grad, ard_num_dims > 1, or diag): inputs scaled by sqrt(l), squared
    distance, and the file's OWN postprocess_rbf (:6-7) = div_(-1).exp_()  =>  exp(-sum_d l_d (x1_d-x2_d)^2).
    Branch 2 (:33-40): [3P] gpytorch RBFCovariance.apply(x1, x2, l, sq_dist)  =>  exp(-0.5 ||(x1-x2)/l||^2) (the
    ``postprocess=False`` distance, then RBFCovariance's own div_(-2).exp_())."""
    eq = x1.shape == x2.shape and torch.equal(x1, x2)
    if x1.requires_grad or x2.requires_grad or (ard_num_dims is not None and ard_num_dims > 1) or diag:
        s = lengthscale.sqrt()
        K = sq_dist_gpytorch(x1 * s, x2 * s, eq).div(-1).exp()
        return torch.diagonal(K) if diag else K
    return sq_dist_gpytorch(x1 / lengthscale, x2 / lengthscale, eq).div(-2).exp()


# ---------------------------------------------------------------------------------------------------
# categorical encoding (models/gp_plus.py:1027-1095)
# ---------------------------------------------------------------------------------------------------
def zeta_matrix(num_levels: Sequence[int]):
    """models/gp_plus.py:1027-1073: all level combinations (itertools.product order), the str(list)->row dictionary and
    the concatenated one-hot matrix zeta (P x sum(levels))."""
    perm = torch.tensor(list(itertools.product(*[range(l) for l in num_levels])), dtype=torch.int64)
    perm_dict = {}
    for i, row in enumerate(perm):
        perm_dict.setdefault(str(row.tolist()), i)
    one_hot = [torch.nn.functional.one_hot(perm[:, c]) for c in range(perm.shape[1])]
    return torch.cat(one_hot, dim=1), perm, perm_dict


def transform_categorical(xcat: torch.Tensor, perm_dict: Dict[str, int], zeta: torch.Tensor) -> torch.Tensor:
    """models/gp_plus.py:1077-1095 (train mode): per-row str(list) lookup -> rows of zeta."""
    if xcat.dim() == 1:
        xcat = xcat.reshape(-1, 1)
    index = [perm_dict[str(row.tolist())] for row in xcat]
    return zeta[index, :]


# ---------------------------------------------------------------------------------------------------
# [3P] Cholesky with jitter retries (gpytorch.utils.cholesky.psd_safe_cholesky)
# ---------------------------------------------------------------------------------------------------
class NotPSDError(RuntimeError):
    pass


class NanError(RuntimeError):
    pass


def psd_safe_cholesky(A: torch.Tensor, jitter: float = 1e-8, max_tries: int = 3):
    """[3P] try cholesky; on failure add jitter * 10^i (i = 0..max_tries-1) to the diagonal, warn, else NotPSDError.
    Returns (L, jitter_used)."""
    L, info = torch.linalg.cholesky_ex(A)
    if not torch.any(info):
        return L, 0.0
    if torch.isnan(A).any():
        raise NanError(f"cholesky_cpu: {int(torch.isnan(A).sum())} of {A.numel()} elements are NaN.")
    Aprime = A.clone()
    jitter_prev = 0.0
    for i in range(max_tries):
        jitter_new = jitter * (10 ** i)
        Aprime.diagonal(dim1=-2, dim2=-1).add_(jitter_new - jitter_prev)
        jitter_prev = jitter_new
        L, info = torch.linalg.cholesky_ex(Aprime)
        if not torch.any(info):
            warnings.warn(f"A not p.d., added jitter of {jitter_new:.1e} to the diagonal", RuntimeWarning)
            return L, jitter_new
    raise NotPSDError(f"Matrix not positive definite after repeatedly adding jitter up to {jitter_new:.1e}.")


# ---------------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------------
class OracleGP:
    """Restates GP_Plus.__init__ (models/gp_plus.py:79-382) + GPR.__init__ (models/gpregression.py:39-115) for the
    deterministic path: one manifold for all categorical columns, Rough_RBF / RBFKernel quantitative kernel, single /
    multiple constant or zero means, single or per-source noise.  ``params`` holds the raw parameters under the
    reference's state_dict names; all fp64."""

    def __init__(self, train_x, train_y, qual_dict: Optional[dict] = None, multiple_noise: bool = False,
                 lb_noise: float = 1e-8, fix_noise: bool = False, fix_noise_val: float = 1e-5,
                 quant_correlation_class: str = "Rough_RBF", embedding_dim: int = 2, m_gp: str = "single_constant",
                 m_gp_ref: str = "zero", seed: int = 0, ard_num_dims="all", fixed_weights=None):
        qual_dict = dict(qual_dict or {})
        self.train_x = torch.as_tensor(train_x, dtype=DT).clone()
        train_y = torch.as_tensor(train_y, dtype=DT).reshape(-1)
        self.N = self.train_x.shape[0]
        self.lb_noise, self.fix_noise, self.kclass = lb_noise, fix_noise, quant_correlation_class
        # index bookkeeping: models/gp_plus.py:190-217
        self.qual_cols: List[int] = list(qual_dict.keys())
        self.quant_index = sorted(set(range(self.train_x.shape[-1])) - set(self.qual_cols))
        self.levels = list(qual_dict.values())
        if len(self.qual_cols) == 1 and self.levels[0] < 2:
            self.quant_index = self.quant_index + [self.qual_cols[0]]
            self.qual_cols = []
        self.dz = embedding_dim if self.qual_cols else 0
        self.noise_indices = list(range(self.levels[-1])) if multiple_noise else []  # gp_plus.py:202-205
        # y scaling: models/gpregression.py:67-69
        self.y_min = train_y.min()
        self.y_std = train_y.max() - train_y.min()
        self.y_sc = (train_y - self.y_min) / self.y_std
        self.m_gp, self.m_gp_ref = m_gp, m_gp_ref
        self.num_sources = int(torch.max(self.train_x[:, -1]))  # gp_plus.py:371
        # parameters ([3P] gpytorch raw parameters initialise to 0)
        P: Dict[str, torch.Tensor] = {}
        S = max(1, len(self.noise_indices))
        P["likelihood.noise_covar.raw_noise"] = torch.zeros(S, dtype=DT)
        if fix_noise:  # gpregression.py:85-87: noise := fix_noise_val, no grad
            P["likelihood.noise_covar.raw_noise"] = torch.log(torch.full((S,), fix_noise_val - lb_noise, dtype=DT))
        P["covar_module.raw_outputscale"] = torch.zeros((), dtype=DT)
        dq = len(self.quant_index)
        self.ard_num_dims = self.train_x.shape[1] if ard_num_dims == "all" else ard_num_dims
        self.fixed_weights = torch.zeros(1, dq, dtype=DT) if fixed_weights is None else \
            torch.as_tensor(fixed_weights, dtype=DT).reshape(1, -1)
        if quant_correlation_class.startswith("GPR:") and self.ard_num_dims in (None, 1):
            dq = 1  # gpytorch: raw_lengthscale has shape (1, 1) without ARD
        if dq > 0:
            key = "covar_module.base_kernel.kernels.1.raw_lengthscale" if self.qual_cols else \
                "covar_module.base_kernel.raw_lengthscale"
            P[key] = torch.zeros(1, dq, dtype=DT)
            self.ls_key = key
        else:
            self.ls_key = None
        if self.qual_cols:
            self.zeta, self.perm, self.perm_dict = zeta_matrix(self.levels)
            g = torch.Generator().manual_seed(seed)
            # FFNN with no hidden layers = Linear_MAP (gp_plus.py:1245-1247, 1456-1461); nn.Linear init replaced by N(0,1)
            self.latent_key = "latent" + str(self.qual_cols)
            P[self.latent_key] = torch.randn(self.dz, sum(self.levels), generator=g, dtype=DT)
            self.cat_index = self._cat_index(self.train_x)
        if m_gp == "single_constant":
            P["mean_module.constant"] = torch.zeros(1, dtype=DT)
        elif m_gp == "multiple_constant":
            for s in range(1, self.num_sources + 1):  # source 0 uses m_gp_ref ('zero'), gp_plus.py:499-507
                P[f"mean_module_{s}.constant"] = torch.zeros(1, dtype=DT)
            if m_gp_ref != "zero":
                P["mean_module_0.constant"] = torch.zeros(1, dtype=DT)
        elif m_gp != "single_zero":
            raise ValueError(m_gp)
        self.params = P
        self.trainable = [k for k in P if not (fix_noise and k.endswith("raw_noise"))]

    # ---- pieces of forward -----------------------------------------------------------------------
    def _cat_index(self, x):
        """transform_categorical's row lookup (gp_plus.py:1085), as integer indices into zeta."""
        xc = x[:, self.qual_cols].to(torch.int64)
        return torch.tensor([self.perm_dict[str(r.tolist())] for r in xc], dtype=torch.int64)

    def features(self, x, p=None):
        """gp_plus.py:408-437: x_new = cat([A(zeta_rows), x[:, quant_index]])."""
        p = self.params if p is None else p
        xq = x[:, self.quant_index]
        if not self.qual_cols:
            return xq if len(self.quant_index) == x.shape[1] else xq
        zrows = transform_categorical(x[:, self.qual_cols].to(torch.int64), self.perm_dict, self.zeta).to(DT)
        z = torch.nn.functional.linear(zrows, p[self.latent_key])  # Linear_MAP.forward, gp_plus.py:1460-1461
        return torch.cat([z, xq], dim=-1)

    def mean(self, x_raw, p=None):
        """gp_plus.py:465-470, 509-544."""
        p = self.params if p is None else p
        n = x_raw.shape[0]
        if self.m_gp == "single_zero":
            return torch.zeros(n, dtype=DT)
        if self.m_gp == "single_constant":
            return p["mean_module.constant"].expand(n)
        src = x_raw[:, -1].to(torch.int64)
        m = torch.zeros(n, dtype=DT)
        for s in range(self.num_sources + 1):
            key = f"mean_module_{s}.constant"
            if key in p:
                m = torch.where(src == s, p[key].expand(n), m)
        return m

    def lengthscale(self, p=None):
        p = self.params if p is None else p
        raw = p[self.ls_key]
        if self.kclass in ("GPR:Rough_RBF", "GPR:wighted_RBF"):
            return exp_lengthscale(raw)  # models/gpregression.py:93-95: Positive(transform=exp) for a kernel given by name
        # gp_plus.py:243-272: only 'RBFKernel' uses exp; Rough_RBF and the Matern classes share the 10^(-x/2) transform
        return exp_lengthscale(raw) if self.kclass == "RBFKernel" else rough_lengthscale(raw)

    def prior_cov(self, U1, U2, p=None):
        """gp_plus.py:219-303 + gpregression.py:108-111: ScaleKernel(RBF(z; l=1) * RBF(x_quant; l(omega)))."""
        p = self.params if p is None else p
        K = None
        if self.qual_cols:
            K = rbf_gpytorch(U1[:, : self.dz], U2[:, : self.dz], torch.ones(1, self.dz, dtype=DT))  # gp_plus.py:223-226
        if self.kclass == "GPR:Rough_RBF":
            # models/gpregression.py:89-102: kernels.Rough_RBF(ard_num_dims = all input columns), no active_dims
            K = rough_rbf_standalone(U1, U2, self.lengthscale(p), ard_num_dims=self.ard_num_dims)
            return softplus(p["covar_module.raw_outputscale"]) * K
        if self.kclass == "GPR:wighted_RBF":
            # the product's kernels.wighted_RBF (documented deviation from the reference's unfinished stub,
            # kernels/wighted_RBF.py:31-41): exp(-sum_d (fixed_d + l_d) dx_d^2), ARD branch of Rough_RBF with shifted weights
            K = rough_rbf_standalone(U1, U2, self.lengthscale(p) + self.fixed_weights, ard_num_dims=max(2, self.ard_num_dims))
            return softplus(p["covar_module.raw_outputscale"]) * K
        if self.ls_key is not None:
            if self.kclass in ("Matern32Kernel", "Matern52Kernel"):
                Kq = matern_gpytorch(U1[:, self.dz:], U2[:, self.dz:], self.lengthscale(p),
                                     1.5 if self.kclass == "Matern32Kernel" else 2.5)
            else:
                Kq = rbf_gpytorch(U1[:, self.dz:], U2[:, self.dz:], self.lengthscale(p))
            K = Kq if K is None else K * Kq
        return softplus(p["covar_module.raw_outputscale"]) * K

    def noise_vector(self, x_raw, p=None):
        """GaussianLikelihood, or likelihoods_noise/multifidelity.py:78-136: sum_k 1[fidel==noise_indices[k]] * noise_k."""
        p = self.params if p is None else p
        tau = noise_transform(p["likelihood.noise_covar.raw_noise"], self.lb_noise)
        n = x_raw.shape[0]
        if not self.noise_indices:
            return tau.expand(n)
        fid = x_raw[:, -1]
        out = torch.zeros(n, dtype=DT)
        for k, lvl in enumerate(self.noise_indices):
            out = out + (fid == lvl).to(DT) * tau[k]
        return out

    def forward(self, x, p=None, literal_moment_matching: bool = False):
        """GP_Plus.forward (gp_plus.py:386-484), deterministic single pass."""
        U = self.features(x, p)
        m = self.mean(x, p)
        K = self.prior_cov(U, U, p)
        if literal_moment_matching:  # gp_plus.py:474-481 with k = 1: (K + m m^T)/1 - m m^T
            K = (K + torch.outer(m, m)) / 1 - torch.outer(m, m)
        return m, K

    # ---- objective ---------------------------------------------------------------------------------
    def log_priors(self, p=None):
        """[3P] ExactMarginalLogLikelihood: sum of prior.log_prob over named_priors()."""
        p = self.params if p is None else p
        tot = log_half_horseshoe_log_prob(p["likelihood.noise_covar.raw_noise"], 0.01, self.lb_noise).sum()  # gpregression.py:84
        tot = tot + lognormal_log_prob(softplus(p["covar_module.raw_outputscale"]), 1e-6, 1.0)  # gpregression.py:113-115
        if self.ls_key is not None:
            if self.kclass in ("RBFKernel", "GPR:Rough_RBF", "GPR:wighted_RBF"):  # gpregression.py:96-98 for the GPR:* classes
                tot = tot + mollified_uniform_log_prob(p[self.ls_key], math.log(0.1), math.log(10)).sum()  # gp_plus.py:274-277
            else:
                tot = tot + normal_log_prob(p[self.ls_key], -3.0, 3.0).sum()  # gp_plus.py:279-295 (Rough_RBF, Matern*)
        for k, v in p.items():
            if k.startswith("mean_module") and k.endswith(".constant"):
                tot = tot + normal_log_prob(v, 0.0, 1.0).sum()  # gp_plus.py:495
        if self.qual_cols:
            tot = tot + normal_log_prob(p[self.latent_key], 0.0, 1.0).sum()  # gp_plus.py:1247
        return tot

    def mll(self, p=None, literal_moment_matching=False):
        """[3P] MultivariateNormal.log_prob through Cholesky: -0.5 (r^T Ky^-1 r + logdet + N log 2pi)."""
        m, K = self.forward(self.train_x, p, literal_moment_matching)
        Ky = K + torch.diag(self.noise_vector(self.train_x, p))
        L, _ = psd_safe_cholesky(Ky)
        r = (self.y_sc - m).unsqueeze(-1)
        z = torch.linalg.solve_triangular(L, r, upper=False)
        quad = (z * z).sum()
        logdet = 2 * torch.log(torch.diagonal(L)).sum()
        return -0.5 * (quad + logdet + self.N * math.log(2 * math.pi))

    def loss(self, p=None, normalize: bool = True, literal_moment_matching=False):
        """optim/mll_torch.py:116 loss = -mll(output, y) ([3P] (log_prob + priors) / N); optim/mll_scipy.py:39-43 when
        ``normalize`` is False."""
        val = self.mll(p, literal_moment_matching) + self.log_priors(p)
        return -(val / self.N if normalize else val)

    def loss_and_grad(self, normalize: bool = True):
        """optim/mll_torch.py:114-117: forward, loss, loss.backward()."""
        p = {k: v.clone().requires_grad_(k in self.trainable) for k, v in self.params.items()}
        loss = self.loss(p, normalize)
        names = [k for k in self.trainable]
        grads = torch.autograd.grad(loss, [p[k] for k in names])
        return loss.detach(), {k: g for k, g in zip(names, grads)}

    def fit_adam(self, num_iter: int = 100, lr: float = 0.01, break_steps: int = 50):
        """One run of optim/mll_torch.py:99-137 (no restarts): torch.optim.Adam(lr) over the trainable raw parameters,
        ``loss = -mll(output, y); loss.backward(); optimizer.step()``, ``loss.item()`` appended per iteration, the early
        stop of :126-128 on the float32 mean of the previous window.  Leaves the final parameters in ``self.params``
        and returns the loss history."""
        p = {k: v.clone().requires_grad_(k in self.trainable) for k, v in self.params.items()}
        opt = torch.optim.Adam([p[k] for k in self.trainable], lr=lr)
        hist = []
        for j in range(num_iter):
            opt.zero_grad()
            loss = self.loss(p)
            loss.backward()
            opt.step()
            hist.append(loss.item())
            if j > break_steps and j % break_steps == 0:
                if (torch.mean(torch.Tensor(hist)[j - break_steps:j]) - hist[j]) <= 0:
                    break
        self.params = {k: v.detach().clone() for k, v in p.items()}
        return hist

    # ---- restart start points ------------------------------------------------------------------------
    def reset_parameters(self, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """models/gpregression.py:168-174: every trainable parameter with a prior is re-drawn,
        ``setting_closure(module, prior.expand(closure(module).shape).sample().to(**tkwargs))``, in the order of [3P] gpytorch
        ``Module.named_priors()`` — a module's own priors in registration order, then its children in registration order:
          GP_Plus itself      latent map          NormalPrior(0, 1)              models/gp_plus.py:1245-1247
          .likelihood         raw_noise           LogHalfHorseshoePrior(0.01, lb) models/gpregression.py:84
          .covar_module       outputscale         LogNormalPrior(1e-6, 1)        models/gpregression.py:113-115
            .base_kernel[...] raw_lengthscale     NormalPrior(-3, 3) / MollifiedUniformPrior(log .1, log 10)
                                                                                  models/gp_plus.py:274-295, gpregression.py:96-98
          .mean_module[_s]    constant            NormalPrior(0, 1)              models/gp_plus.py:495
        (likelihood and covar_module are set in GPR.__init__, the latent map and the means after it: gp_plus.py:305-382).
        The reference builds its priors from Python numbers, i.e. in torch's DEFAULT dtype (float32), and never casts them
        on the CPU: every draw is made in float32 from the global generator and converted afterwards.  Draws, written out:
          NormalPrior     [3P] torch.distributions.Normal.sample:  torch.normal(loc.expand(shape), scale.expand(shape))
          LogNormalPrior  [3P] TransformedDistribution.sample:     exp(torch.normal(loc, scale))
          MollifiedUniformPrior.rsample  priors/mollified_uniform.py:84-85:  Uniform(a, b).rsample() = a + rand * (b - a)
          LogHalfHorseshoePrior          priors/horseshoe.py:68-79, see ``_horseshoe_draw``.
        A parameter that does not require grad is skipped WITHOUT consuming random numbers (gpregression.py:172-173)."""
        f32 = torch.get_default_dtype()
        g = generator

        def normal(loc: float, scale: float, shape) -> torch.Tensor:
            return torch.normal(torch.full(tuple(shape), loc, dtype=f32), torch.full(tuple(shape), scale, dtype=f32), generator=g)

        P = self.params
        if self.qual_cols:
            P[self.latent_key] = normal(0.0, 1.0, P[self.latent_key].shape).to(DT)
        nk = "likelihood.noise_covar.raw_noise"
        if not self.fix_noise:
            P[nk] = self._horseshoe_draw(0.01, P[nk].shape, g).to(DT)
        # 'outputscale' names the CONSTRAINED value: the setter stores raw = inv_softplus(value) ([3P] ScaleKernel._set_outputscale),
        # evaluated after the conversion to the model's dtype
        v = torch.exp(normal(1e-6, 1.0, ()))
        P["covar_module.raw_outputscale"] = inv_softplus(v.to(DT))
        if self.ls_key is not None:
            shape = P[self.ls_key].shape
            if self.kclass in ("RBFKernel", "GPR:Rough_RBF", "GPR:wighted_RBF"):
                a, b = torch.tensor(math.log(0.1), dtype=f32), torch.tensor(math.log(10), dtype=f32)
                P[self.ls_key] = (a + torch.rand(tuple(shape), dtype=f32, generator=g) * (b - a)).to(DT)
            else:
                P[self.ls_key] = normal(-3.0, 3.0, shape).to(DT)
        for k in [k for k in P if k.startswith("mean_module") and k.endswith(".constant")]:  # insertion order = registration order
            P[k] = normal(0.0, 1.0, P[k].shape).to(DT)
        return P

    @staticmethod
    def _horseshoe_draw(scale: float, shape, generator=None) -> torch.Tensor:
        """priors/horseshoe.py:68-79.  ``expand`` (:77-79) rebuilds the prior from the expanded ``scale`` ALONE, so the draw
        is clamped at the constructor's default lb = 1e-6, not at the model's ``lb_noise`` (SURVEY B-7); ``lb`` then has the
        shape of ``scale`` and the clamp reads ``lb[0]`` when there are several noise levels (:71-74).
          local_shrinkage = HalfCauchy(1).rsample(scale.shape)        = | Cauchy(0, 1) draw |    ([3P] loc.new(shape).cauchy_())
          param_sample    = HalfNormal(local_shrinkage * scale).rsample() = | N(0, 1) draw * (local_shrinkage * scale) |
          param_sample[param_sample < lb] = lb ;  return log(param_sample)         (the prior is on log noise = raw_noise)."""
        f32 = torch.get_default_dtype()
        shape = tuple(shape)
        sc = torch.full(shape, scale, dtype=f32)
        lb = torch.full(shape, 1e-6, dtype=f32)
        local_shrinkage = torch.empty(shape, dtype=f32).cauchy_(generator=generator).abs()
        eps = torch.normal(torch.zeros(shape, dtype=f32), torch.ones(shape, dtype=f32), generator=generator)
        param_sample = (eps * (local_shrinkage * sc)).abs()
        floor = lb[0] if lb.numel() > 1 else lb
        param_sample = torch.where(param_sample < floor, floor.expand_as(param_sample), param_sample)
        return param_sample.log()

    # ---- prediction --------------------------------------------------------------------------------
    @torch.no_grad()
    def _predict_parts(self, xtest):
        """models/gpregression.py:126-134 + [3P] exact prediction strategy on the joint forward over cat([train, test]):
        scaled-space mean, noise-free variance (before the clamp) and the test points' own noise, from ONE factorisation."""
        xtest = torch.as_tensor(xtest, dtype=DT)
        xall = torch.cat([self.train_x, xtest], dim=0)
        m, K = self.forward(xall)
        n = self.N
        Ky = K[:n, :n] + torch.diag(self.noise_vector(self.train_x))
        L, _ = psd_safe_cholesky(Ky)
        del Ky
        r = (self.y_sc - m[:n]).unsqueeze(-1)
        alpha = torch.cholesky_solve(r, L).squeeze(-1)
        Ksn = K[n:, :n]
        mean = m[n:] + Ksn @ alpha
        V = torch.linalg.solve_triangular(L, Ksn.T, upper=False)
        var = torch.diagonal(K[n:, n:]) - (V * V).sum(0)
        return mean, var, self.noise_vector(xtest)

    @torch.no_grad()
    def predict(self, xtest, return_std: bool = True, include_noise: bool = True):
        """models/gpregression.py:122-149 (un-scaling :142-147; [3P] settings.min_variance clamp, double: 1e-10)."""
        mean, var, noise = self._predict_parts(xtest)
        out_mean = self.y_min + self.y_std * mean
        if not return_std:
            return out_mean
        if include_noise:
            var = var + noise
        var = var.clamp_min(1e-10)  # [3P] settings.min_variance (double)
        return out_mean, var.sqrt() * self.y_std

    @torch.no_grad()
    def predict_all(self, xtest):
        """(mean, std incl. noise, std without noise) of ``predict`` from one factorisation (fixture generation at the
        BASELINE sizes, tests/golden/make_fullsize.py)."""
        mean, var, noise = self._predict_parts(xtest)
        return (self.y_min + self.y_std * mean, (var + noise).clamp_min(1e-10).sqrt() * self.y_std,
                var.clamp_min(1e-10).sqrt() * self.y_std)

    @torch.no_grad()
    def evaluation(self, xtest, ytest, alpha: float = 0.05):
        """models/gp_plus.py:889-932: ``trained_pred_dist = likelihood(self(Xtest))`` (joint predictive MVN of the test
        points incl. their own sources' noise), then [3P] gpytorch.metrics: NLPD = -log_prob(y) / M, MSE, MAE; the interval
        score of :907-913 on the +-2 sigma confidence region; MSE / MAE / IS scaled back (:916-919), RRMSE (:921)."""
        xtest = torch.as_tensor(xtest, dtype=DT)
        ytest = torch.as_tensor(ytest, dtype=DT).reshape(-1)
        m, K = self.forward(torch.cat([self.train_x, xtest], dim=0))
        n = self.N
        L, _ = psd_safe_cholesky(K[:n, :n] + torch.diag(self.noise_vector(self.train_x)))
        alpha_v = torch.cholesky_solve((self.y_sc - m[:n]).unsqueeze(-1), L).squeeze(-1)
        mean = m[n:] + K[n:, :n] @ alpha_v
        V = torch.linalg.solve_triangular(L, K[n:, :n].T, upper=False)
        cov = K[n:, n:] - V.T @ V + torch.diag(self.noise_vector(xtest))
        y_sc = (ytest - self.y_min) / self.y_std
        M = ytest.shape[0]
        Lp, _ = psd_safe_cholesky(cov)
        z = torch.linalg.solve_triangular(Lp, (y_sc - mean).unsqueeze(-1), upper=False)
        log_prob = -0.5 * ((z * z).sum() + 2 * torch.log(torch.diagonal(Lp)).sum() + M * math.log(2 * math.pi))
        std = torch.diagonal(cov).clamp_min(1e-10).sqrt()
        lo, up = mean - 2 * std, mean + 2 * std
        score = (up - lo) + (y_sc > up) * 2 / alpha * (y_sc - up) + (y_sc < lo) * 2 / alpha * (lo - y_sc)
        mse = ((y_sc - mean) ** 2).mean() * self.y_std ** 2
        return {"NLL": -log_prob / M, "MSE": mse, "MAE": (y_sc - mean).abs().mean() * self.y_std.abs(),
                "RRMSE": torch.sqrt(mse / torch.var(ytest)), "IS": score.mean() * self.y_std.abs()}


# ---------------------------------------------------------------------------------------------------
# consumers of predict(): acquisition functions, one step of the pool-based BO loop, Sobol indices
# ---------------------------------------------------------------------------------------------------
def _af_parts(mean, std, best_f, cost, maximize: bool, si: float):
    """bayesian_optimizations/AFs.py:13-26 (the same lines in every function of the file): u, sigma and the cost."""
    mean = torch.as_tensor(mean, dtype=DT).reshape(-1, 1)
    view_shape = mean.shape[:-2] if mean.shape[-2] == 1 else mean.shape[:-1]
    mean = mean.view(view_shape)
    sigma = torch.as_tensor(std, dtype=DT).view(view_shape)
    u = (mean - best_f - math.copysign(1.0, best_f) * (0.0 if best_f == 0 else 1.0) * si) / sigma  # np.sign(best_f) * si
    cost = torch.ones(u.shape, dtype=DT) if cost is None else torch.as_tensor(cost, dtype=DT).view(u.shape)
    if not maximize:
        u = -u
    return u, sigma, cost


def _std_normal_pdf_cdf(u: torch.Tensor):
    """[3P] torch.distributions.Normal(0, 1): exp(log_prob(u)) and cdf(u), written out."""
    pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
    cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
    return pdf, cdf


def oracle_af(kind: str, samples, best_f: float, oracle: "OracleGP", xmean, xstd, cost_fun, maximize: bool = False,
              si: float = 0.0) -> float:
    """AF_LF (AFs.py:1-30), AF_HF (:33-64) and AF_EI (:67-100) of ONE raw point ``samples`` = [quantitative coordinates, source]:
    standardise the coordinates (:5), predict with noise (:8), u = (mean - best_f - sign(best_f) si) / sigma (:16), negated for
    minimisation (:23-24), then  LF: -sigma pdf(u) / cost,  HF: -sigma u / cost,  EI: -sigma (pdf(u) + u cdf(u)) / cost."""
    import numpy as np

    samples = np.asarray(samples, dtype=np.float64)
    x = np.concatenate([((samples[0:-1] - np.asarray(xmean)) / np.asarray(xstd)).reshape(1, -1), samples[-1].reshape(-1, 1)], axis=-1)
    mean, std = oracle.predict(x.reshape(1, -1), return_std=True, include_noise=True)
    cost = torch.tensor([float(cost_fun(v)) for v in x[:, -1]], dtype=DT)
    u, sigma, cost = _af_parts(mean, std, best_f, cost, maximize, si)
    pdf, cdf = _std_normal_pdf_cdf(u)
    if kind == "LF":
        val = sigma * pdf
    elif kind == "HF":
        val = sigma * u
    elif kind == "EI":
        val = sigma * (pdf + u * cdf)
    else:
        raise ValueError(kind)
    return float(-1 * (val / cost))


def oracle_af_engineering(kind: str, best_f: float, mean, std, x_val, cost_fun, maximize: bool = True, si: float = 0.0):
    """AF_LF_Engineering (AFs.py:103-131: sigma pdf(u) / cost) and AF_HF_Engineering (:134-159: sigma u / cost) on a pool."""
    x_val = torch.as_tensor(x_val, dtype=DT)
    cost = torch.tensor([float(cost_fun(v)) for v in x_val[:, -1]], dtype=DT)
    u, sigma, cost = _af_parts(mean, std, best_f, cost, maximize, si)
    pdf, _ = _std_normal_pdf_cdf(u)
    return (sigma * pdf if kind == "LF" else sigma * u) / cost


def oracle_bo_pool_scores(oracle: "OracleGP", pool, best_values, cost_fun, num_fidelity: int, maximize: bool = False):
    """The selection of one iteration of the pool branch of BO (BO_GP_plus.py:183-197): for every source i the candidates
    ``pool[pool[:, -2] == i]`` (columns: inputs ..., source, response) are predicted WITHOUT noise (:187), scored with the
    high-fidelity utility for source 0 and the low-fidelity one otherwise (:188-192), the scores concatenated in source order
    (:194) and the argmax taken (:195).  Returns (scores, index).  As in the reference the index addresses the concatenation
    (which the reference then applies to the pool itself, :197 — identical when the pool is sorted by source)."""
    pool = torch.as_tensor(pool, dtype=DT)
    scores = []
    for i in range(num_fidelity):
        cand = pool[pool[:, -2] == i][:, 0:-1]
        mean, std = oracle.predict(cand, return_std=True, include_noise=False)
        kind = "HF" if i == 0 else "LF"
        scores.append(oracle_af_engineering(kind, best_values[i], mean.reshape(-1, 1), std.reshape(-1, 1), cand, cost_fun,
                                            maximize=maximize).reshape(-1))
    scores = torch.cat(scores, dim=0)
    return scores, int(torch.argmax(scores))


def oracle_sobol(oracle: "OracleGP", sequence, levels_per_cat: Sequence[int]):
    """Sobol indices of the posterior mean by Saltelli's scheme (models/gp_plus.py:1148-1224) on a GIVEN (N, 2p) low-discrepancy
    ``sequence`` in [0, 1): A from columns p.., B from columns ..p (:1177-1178), scaled to the training inputs' range (:1184-1185),
    categorical columns mapped to their level grid and rounded (:1188-1192; every categorical column, see GP_Plus.Sobol here),
    S_i = mean(FB (F_ABi - FA)) / Var, ST_i = mean((FA - F_ABi)^2) / 2 Var with Var over cat([FA, FB]) (:1212-1220)."""
    seq = torch.as_tensor(sequence, dtype=DT)
    p = oracle.train_x.shape[1]
    mins, maxs = oracle.train_x.min(dim=0)[0], oracle.train_x.max(dim=0)[0]
    halves = []
    for part in (seq[:, p:], seq[:, :p]):
        scaled = mins + (maxs - mins) * part
        for j, col in enumerate(oracle.qual_cols):
            scaled[:, col] = (part[:, col] * (levels_per_cat[j] - 1)).round()
        halves.append(scaled)
    A, B = halves
    N = A.shape[0]
    f = lambda Z: oracle.predict(Z, return_std=False).reshape(-1, 1)  # noqa: E731
    FA, FB = f(A), f(B)
    S, ST = torch.zeros(p, 1, dtype=DT), torch.zeros(p, 1, dtype=DT)
    for i in range(p):
        ABi = A.clone()
        ABi[:, i] = B[:, i]
        Fi = f(ABi)
        S[i] = (FB * (Fi - FA)).sum(0) / N
        ST[i] = ((FA - Fi) ** 2).sum(0) / (2 * N)
    varY = torch.var(torch.cat([FA, FB]), dim=0, unbiased=False)  # np.var
    return (S / varY).T.numpy(), (ST / varY).T.numpy()


# ---------------------------------------------------------------------------------------------------
# scipy multistart driver and noise continuation, restated on the oracle
# ---------------------------------------------------------------------------------------------------
def _pack(o: "OracleGP", names: Sequence[str]):
    """optim/mll_scipy.py:77-82 pack_parameters: the trainable raw parameters as one float64 vector."""
    import numpy as np

    return np.concatenate([o.params[k].detach().numpy().astype(np.float64).ravel() for k in names])


def _unpack(o: "OracleGP", names: Sequence[str], x, fp32_theta: bool = False) -> None:
    """optim/mll_scipy.py:84-98 unpack_parameters + :103-106 load_state_dict.  ``fp32_theta``: the reference casts every
    slice of theta to ``tkwargs`` = float32 before loading it (:32-35, :97), whatever the model's own dtype."""
    import numpy as np

    i = 0
    for k in names:
        n = o.params[k].numel()
        v = torch.from_numpy(np.asarray(x[i:i + n], dtype=np.float64).copy())
        if fp32_theta:
            v = v.to(torch.float32)
        o.params[k] = v.to(DT).reshape(o.params[k].shape)
        i += n


def oracle_fit_scipy(o: "OracleGP", add_prior: bool = True, num_restarts: int = 1, theta0_list=None, options: Optional[dict] = None,
                     generator: Optional[torch.Generator] = None, fp32_theta: bool = False):
    """``fit_model_scipy`` (optim/mll_scipy.py:243-307) with method 'L-BFGS-B' on the oracle: objective
    ``-(log_prob + sum of prior log-densities)``, NOT divided by N (:37-43, :120), over the trainable raw parameters; defaults
    ftol 1e-6 / gtol 1e-5 / maxfun 5000 / maxiter 2000 (:262); a start that raises NotPSDError / NanError is kept as the exception
    and scored inf (:232-236, :295); the best start's theta is loaded (:296-301).  Start points when ``theta0_list`` is None
    (:277-281): ``num_restarts + 1`` prior draws.  The reference concatenates ``prior.sample()`` in named_priors order and on the
    constrained scale (:130-137), which does not line up with its own packing; the build under test draws them with
    ``reset_parameters`` (models/gpregression.py:168-174) instead — a documented deviation — and so does this restatement, so that
    both drivers start from the same points.  Returns (list of OptimizeResult / exceptions, best objective)."""
    import numpy as np
    from scipy.optimize import minimize

    defaults = {'ftol': 1e-6, 'gtol': 1e-5, 'maxfun': 5000, 'maxiter': 2000}
    for key, val in (options or {}).items():
        if key not in defaults:
            raise RuntimeError('Unknown option %s!' % key)
        defaults[key] = val
    names = list(o.trainable)

    def fun(x):  # MLLObjective.fun, :101-127
        _unpack(o, names, x, fp32_theta)
        p = {k: v.clone().requires_grad_(k in names) for k, v in o.params.items()}
        val = o.mll(p)
        if add_prior:
            val = val + o.log_priors(p)
        obj = -val
        grads = torch.autograd.grad(obj, [p[k] for k in names])
        return obj.item(), np.concatenate([g.numpy().ravel() for g in grads]).astype(np.float64)

    if theta0_list is None:
        theta0_list = [_pack(o, names)]
        if num_restarts > -1:
            saved = {k: v.clone() for k, v in o.params.items()}
            samples = []
            for _ in range(num_restarts + 1):
                o.reset_parameters(generator)
                samples.append(_pack(o, names))
            o.params = saved
            theta0_list.extend(samples)
            theta0_list.pop(0)
    out = []
    for theta0 in theta0_list:
        try:
            out.append(minimize(fun=fun, x0=theta0, jac=True, method='L-BFGS-B', options=dict(defaults)))
        except (NotPSDError, NanError) as e:
            out.append(e)
    nlls = [np.inf if isinstance(r, Exception) or not np.isfinite(r.fun) else r.fun for r in out]
    best = int(np.argmin(nlls))
    if not isinstance(out[best], Exception) and np.isfinite(nlls[best]):
        _unpack(o, names, out[best].x, fp32_theta)
    return out, nlls[best]


def oracle_continuation(o: "OracleGP", add_prior: bool = True, num_restarts: int = 32, initial_noise_var: float = 1.0,
                        red_factor: float = math.sqrt(10), options: Optional[dict] = None, accuracy: float = 1e-2,
                        generator: Optional[torch.Generator] = None):
    """``fit_model_continuation`` (optim/mll_noise_continuation.py:45-244, criterion 'NLL') on the oracle.  The noise variance is
    FIXED (raw_noise frozen, :127-128) at each of a sequence of levels — first pass: initial / 10^i, i = 0..9 (:143-144); later
    passes: ten levels between the neighbours of the best one (:148-155) — and ``oracle_fit_scipy`` runs at each; the distinct
    optima of a level (Euclidean distance >= 1e-2 * dim, :199-207) start the next; a level at which every start fails ends the pass
    (:178-180); a pass ends the search when its best level is within ``accuracy`` of the pass's first level (:238-240) or lies at
    an end of the list (:156-160).  ``likelihood.initialize(noise=v)`` is raw = log(v - lb) ([3P] GreaterThan(lb, exp).inverse_transform,
    models/gpregression.py:59).  Returns (nll at the selected level, {'noise_history', 'nll_history'}) of the last pass and leaves
    ``o.params`` at the selected state."""
    import numpy as np
    from scipy.spatial import distance_matrix

    nk = "likelihood.noise_covar.raw_noise"
    o.trainable = [k for k in o.trainable if k != nk]
    o.fix_noise = True  # (reset_parameters then skips the noise draw without consuming random numbers: gpregression.py:172-173)

    def set_noise(v):
        # [3P] gpytorch's noise setter: torch.as_tensor(python float) is float32 (the default dtype) before it is cast to the
        # parameter's dtype — the levels are float32-rounded.  The level 1e-8 of the first pass is therefore BELOW
        # the fp64 bound 1e-8: log of a negative number, a NaN covariance, every start fails and the pass ends there (:178-180).
        v32 = torch.as_tensor(float(v), dtype=torch.float32).to(DT)
        o.params[nk] = torch.log(torch.full_like(o.params[nk], float(v32)) - o.lb_noise)

    def noise_now():
        return float(noise_transform(o.params[nk], o.lb_noise).reshape(-1)[0])

    t = 0
    theta0_list = None
    index, history, old_state = None, None, {}
    names = None
    while True:
        t += 1
        first = initial_noise_var
        if t == 1:
            noises = [first / (10 ** i) for i in range(int(10 / t))]
        else:
            n_hist = len(history['noise_history'])
            if (index >= 2 and index < n_hist - 2) or (index >= 1 and index < n_hist - 1):
                noises = np.linspace(history['noise_history'][index - 1], history['noise_history'][index + 1], 10)
                initial_noise_var = history['noise_history'][index - 1]
                o.params = {k: v.clone() for k, v in old_state[index - 1].items()}
            else:
                o.params = {k: v.clone() for k, v in old_state[index].items()}
                return history['nll_history'][index], history
        noise_list, nll_list = [], []
        t += 1
        old_state = {}
        for i in range(len(noises)):
            set_noise(noises[i])
            old_state[i] = {k: v.clone() for k, v in o.params.items()}
            reslist, nll = oracle_fit_scipy(o, add_prior, num_restarts=num_restarts, theta0_list=theta0_list, options=options,
                                            generator=generator)
            if all(isinstance(r, (RuntimeError, TypeError)) for r in reslist):
                break
            noise_list.append(noise_now())
            nll_list.append(nll)
            theta0_list = []
            for r in reslist:
                if isinstance(r, Exception):
                    continue
                if len(theta0_list) > 0:
                    d = distance_matrix(r.x.reshape(1, -1), np.vstack(theta0_list)).ravel()
                    if np.any(d < 1e-2 * r.x.shape[0]):
                        continue
                theta0_list.append(r.x)
            set_noise(noise_list[-1] / red_factor)  # (:209-216; overwritten by the next level's set_noise)
        if not nll_list:
            raise RuntimeError('oracle_continuation: every start failed at the first noise level')
        history = {'noise_history': noise_list, 'nll_history': nll_list}
        index = int(np.argmin(nll_list))
        if abs(first - noise_list[index]) < accuracy:
            o.params = {k: v.clone() for k, v in old_state[index].items()}
            break
    return history['nll_history'][index], history
