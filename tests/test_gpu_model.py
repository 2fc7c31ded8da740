"""End-to-end GPU parity: the GP+ API (GP_Plus / ExactMarginalLogLikelihood / fit_model_torch / predict) running on the
HIP back end, against the committed golden fixtures (inputs from the reference's data pipeline, expected values from
the CPU oracle) and against the oracle itself on fresh seeded inputs.

Tolerances are the north star's: MLL and grad-MLL within 1e-5 relative (fp64); predictive mean / std within 1e-4.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL_MLL = 1e-5
RTOL_PRED = 1e-4


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def build(fx, tag, **kw):
    from gpplus_amd.models import GP_Plus

    xkey = "Xtrain" if "Xtrain" in fx else "Utrain"
    m = GP_Plus(torch.tensor(fx[xkey]), torch.tensor(fx["ytrain"]), dtype=torch.float64, device="cuda", **kw)
    sd = m.state_dict()
    for k in list(sd):
        fk = f"{tag}::param::{k}"
        if fk in fx:
            sd[k] = torch.as_tensor(fx[fk]).reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    return m


def loss_and_grads(m):
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood

    m.train()
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    for p in m.parameters():
        p.grad = None
    out = m(*m.train_inputs)
    loss = -mll(out, m.train_targets)
    loss.backward()
    return loss.item(), {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}


CASES = [
    ("c1_borehole_n500.npz", "theta0", {}),
    ("c1_borehole_n500.npz", "theta1", {}),
    ("c3_borehole_mixed_n100.npz", "theta1", {"qual_dict": {0: 5, 5: 5}}),
    ("c4_wing_mf_n300.npz", "theta1", {"qual_dict": {10: 3}, "multiple_noise": True, "m_gp": "multiple_constant"}),
]


@pytest.mark.parametrize("fixture,tag,kw", CASES)
def test_loss_and_gradients_match_golden(gpu_ctx, fixture, tag, kw):
    fx = load(fixture)
    m = build(fx, tag, **kw)
    loss, grads = loss_and_grads(m)
    ref = float(fx[f"{tag}::loss"])
    assert abs(loss - ref) <= RTOL_MLL * abs(ref), (loss, ref)
    checked = 0
    for name, g in grads.items():
        key = f"{tag}::grad::{name}"
        assert key in fx, f"fixture has no gradient for {name}"
        gref = fx[key].reshape(g.shape)
        scale = max(np.abs(gref).max(), 1e-12)
        np.testing.assert_allclose(g, gref, rtol=RTOL_MLL, atol=RTOL_MLL * scale, err_msg=name)
        checked += 1
    assert checked == sum(1 for k in fx if k.startswith(f"{tag}::grad::"))


@pytest.mark.parametrize("fixture,tag,kw", CASES[1:])
def test_predict_matches_golden(gpu_ctx, fixture, tag, kw):
    fx = load(fixture)
    m = build(fx, tag, **kw)
    xt = torch.tensor(fx["Xtest"] if "Xtest" in fx else fx["Utest"])
    mean, std = m.predict(xt, return_std=True, include_noise=True)
    np.testing.assert_allclose(mean.cpu().numpy(), fx[f"{tag}::pred_mean"], rtol=RTOL_PRED, atol=1e-8)
    np.testing.assert_allclose(std.cpu().numpy(), fx[f"{tag}::pred_std"], rtol=RTOL_PRED, atol=1e-8)
    mean2, std2 = m.predict(xt, return_std=True, include_noise=False)
    np.testing.assert_allclose(std2.cpu().numpy(), fx[f"{tag}::pred_std_nonoise"], rtol=RTOL_PRED, atol=1e-7)
    only_mean = m.predict(xt, return_std=False)
    np.testing.assert_allclose(only_mean.cpu().numpy(), mean.cpu().numpy(), rtol=0, atol=0)


@pytest.mark.parametrize("n,d,seed", [(64, 3, 0), (777, 8, 1), (2048, 8, 2), (4097, 5, 3), (6200, 8, 4), (12288, 6, 5)])
def test_against_oracle_on_fresh_inputs(gpu_ctx, n, d, seed):
    """Same seeded inputs through the oracle (CPU) and the product (GPU), incl. a non-multiple-of-tile size, the sizes
    whose inverse is built by bordering inside the look-ahead factorisation (n >= 4096) and one above that range (12 288:
    look-ahead without bordering); loss, every gradient AND the predictions (models/gpregression.py:122-149) at each size."""
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    y = np.sin(X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
    o = OracleGP(X, y)
    vals = {o.ls_key: np.float32(rng.uniform(-1.5, -0.5, (1, d))), "covar_module.raw_outputscale": np.float32(0.3),
            "likelihood.noise_covar.raw_noise": np.float32([-6.0]), "mean_module.constant": np.float32([0.4])}
    for k, v in vals.items():
        o.params[k] = torch.as_tensor(np.asarray(v, dtype=np.float64)).reshape(o.params[k].shape)
    lo, go = o.loss_and_grad()
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
    sd = m.state_dict()
    for k, v in o.params.items():
        sd[k] = v.reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    loss, grads = loss_and_grads(m)
    assert abs(loss - lo.item()) <= RTOL_MLL * abs(lo.item())
    for k, g in go.items():
        gref = g.numpy().reshape(grads[k].shape)
        np.testing.assert_allclose(grads[k], gref, rtol=RTOL_MLL, atol=RTOL_MLL * max(np.abs(gref).max(), 1e-12), err_msg=k)
    # predictions from the same factor path: points at 0.05 ... 1 standard deviations from training rows
    mt = min(96, n)
    Xt = X[rng.choice(n, mt, replace=False)] + np.array([0.05, 0.3, 1.0])[np.arange(mt) % 3][:, None] * rng.standard_normal((mt, d))
    om, osd, osd0 = o.predict_all(Xt)
    m.eval()
    mean, std = m.predict(torch.tensor(Xt), return_std=True, include_noise=True)
    _, std0 = m.predict(torch.tensor(Xt), return_std=True, include_noise=False)
    np.testing.assert_allclose(mean.cpu().numpy(), om.numpy(), rtol=RTOL_PRED, atol=1e-7)
    np.testing.assert_allclose(std.cpu().numpy(), osd.numpy(), rtol=RTOL_PRED, atol=1e-7)
    np.testing.assert_allclose(std0.cpu().numpy(), osd0.numpy(), rtol=RTOL_PRED, atol=1e-7)


def test_rbfkernel_mode_and_fixed_noise(gpu_ctx):
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(7)
    X = rng.standard_normal((300, 4))
    y = np.cos(X[:, 0]) + X[:, 2]
    o = OracleGP(X, y, quant_correlation_class="RBFKernel", fix_noise=True, fix_noise_val=1e-3, m_gp="single_zero")
    o.params[o.ls_key] = torch.tensor(np.float32([[0.2, -0.1, 0.4, 0.0]]), dtype=torch.float64)
    lo, go = o.loss_and_grad()
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda", quant_correlation_class="RBFKernel",
                fix_noise=True, fix_noise_val=1e-3, m_gp="single_zero")
    sd = m.state_dict()
    sd[o.ls_key] = o.params[o.ls_key].to(sd[o.ls_key])
    m.load_state_dict(sd)
    assert not m.likelihood.raw_noise.requires_grad
    # set through an fp32 tensor in the reference too (models/gpregression.py:87): fp32-accurate only
    np.testing.assert_allclose(m.likelihood.noise.item(), 1e-3, rtol=1e-6)
    loss, grads = loss_and_grads(m)
    assert abs(loss - lo.item()) <= RTOL_MLL * abs(lo.item())
    assert "likelihood.noise_covar.raw_noise" not in grads
    np.testing.assert_allclose(grads[o.ls_key], go[o.ls_key].numpy(), rtol=RTOL_MLL, atol=1e-10)


def test_duplicate_rows_and_jitter_policy(gpu_ctx):
    """Exact duplicate rows (the reference's with-replacement shuffle): Ky is PD only through the noise.
    With the noise at its 1e-8 floor the factorisation must either succeed or go through the warn-and-retry path, and a
    matrix that stays indefinite must raise NotPSDError."""
    from gpplus_amd.gpcore import NotPSDError
    from gpplus_amd.linalg import KernelSpec, exact_mll

    rng = np.random.default_rng(3)
    U = rng.standard_normal((200, 3))
    U[100:] = U[:100]
    Ud = torch.tensor(U, device="cuda")
    spec = KernelSpec(torch.full((3,), 0.3, dtype=torch.float64, device="cuda"), torch.tensor(1.0, dtype=torch.float64, device="cuda"))
    y = torch.tensor(rng.standard_normal(200), device="cuda")
    mean = torch.zeros(200, dtype=torch.float64, device="cuda")
    ok = exact_mll(Ud, spec, torch.tensor([1e-3], dtype=torch.float64, device="cuda"), mean, y)
    assert torch.isfinite(ok)
    with pytest.warns(RuntimeWarning):
        v = exact_mll(Ud, spec, torch.tensor([0.0], dtype=torch.float64, device="cuda"), mean, y)
    assert torch.isfinite(v)
    with pytest.raises(NotPSDError):
        exact_mll(Ud, spec, torch.tensor([-0.5], dtype=torch.float64, device="cuda"), mean, y)


def test_fit_model_torch_improves_and_restores_best_state(gpu_ctx):
    from gpplus_amd.optim import fit_model_torch
    from gpplus_amd.utils import set_seed

    fx = load("c1_borehole_n500.npz")
    set_seed(0)
    m = build(fx, "theta0")
    l0, _ = loss_and_grads(m)
    f_inc, hist = fit_model_torch(m, lr_default=0.05, num_iter=30, num_restarts=1, verbose=False)
    assert len(hist) == 2 and len(hist[0]) == 30
    assert f_inc < l0
    assert abs(f_inc - min(h[-1] for h in hist)) < 1e-12
    keys = set(m.state_dict().keys())
    for k in ("likelihood.noise_covar.raw_noise", "covar_module.raw_outputscale", "covar_module.base_kernel.raw_lengthscale",
              "mean_module.constant", "y_min", "y_std", "y_scaled", "quant_index", "qual_dict_list"):
        assert k in keys, k
    mean, std = m.predict(torch.tensor(fx["Xtest"]), return_std=True)
    rmse = float(((mean.cpu() - torch.tensor(fx["ytest"])) ** 2).mean().sqrt())
    assert rmse < 0.6 * float(np.std(fx["ytest"]))  # 2 x 30 Adam steps from the init point: already far better than the mean


def test_evaluation_and_dense_evaluate(gpu_ctx):
    fx = load("c1_borehole_n500.npz")
    m = build(fx, "theta1")
    res = m.evaluation(torch.tensor(fx["Xtest"]), torch.tensor(fx["ytest"]), verbose=False)
    assert all(torch.isfinite(v).all() for v in res.values())
    m.train()
    dense = m(*m.train_inputs).lazy_covariance_matrix.evaluate()
    assert dense.shape == (500, 500)
    assert torch.allclose(dense, dense.T)
    np.testing.assert_allclose(dense.diagonal().cpu().numpy(), float(m.covar_module.outputscale), rtol=1e-14)


def test_cpu_device_fails_loudly():
    from gpplus_amd._lib import GppError
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.models import GP_Plus

    fx = load("c3_borehole_mixed_n100.npz")
    m = GP_Plus(torch.tensor(fx["Utrain"]), torch.tensor(fx["ytrain"]), qual_dict={0: 5, 5: 5}, dtype=torch.float64)
    with pytest.raises(GppError):
        ExactMarginalLogLikelihood(m.likelihood, m)(m(*m.train_inputs), m.train_targets)


def test_full_size_properties_n20000(gpu_ctx):
    """BASELINE.json's N=20000, d=8 through size-independent properties: K alpha = r, L (Linv v) = v, MLL invariance
    under a permutation of the data, and sum-rule of the noise gradient."""
    from gpplus_amd.backend import square_buffer
    from gpplus_amd.linalg import KernelSpec, exact_mll, dense_kernel, get_workspace

    N, D = 20000, 8
    g = torch.Generator(device="cuda").manual_seed(0)
    U = torch.rand(N, D, dtype=torch.float64, device="cuda", generator=g) * 3.4 - 1.7
    y = torch.sin(U[:, 0]) + 0.1 * torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    w = torch.full((D,), 0.1, dtype=torch.float64, device="cuda", requires_grad=True)
    sf2 = torch.tensor(0.85, dtype=torch.float64, device="cuda", requires_grad=True)
    tau = torch.tensor([2.5e-3], dtype=torch.float64, device="cuda", requires_grad=True)
    mean = torch.zeros(N, dtype=torch.float64, device="cuda")
    mll = exact_mll(U, KernelSpec(w, sf2), tau, mean, y)
    mll.backward()
    ws = get_workspace(gpu_ctx, N, 0)
    alpha = ws.alpha.clone()
    K = dense_kernel(U, KernelSpec(w.detach(), sf2.detach()), tau.detach(), None)
    resid = torch.mv(K, alpha) - y
    assert float(resid.norm() / y.norm()) < 1e-8
    # L (Linv v) = v on the lower factors
    v = torch.randn(N, dtype=torch.float64, device="cuda", generator=g)
    L = torch.triu(ws.A).T  # the factor is stored as U = L^T (upper triangle)
    t = torch.mv(torch.tril(ws.Li), v)
    assert float((torch.mv(L, t) - v).norm() / v.norm()) < 1e-8
    del K, L
    g_w, g_tau = w.grad.clone(), tau.grad.clone()
    # permutation invariance
    perm = torch.randperm(N, device="cuda", generator=g)
    w2 = w.detach().clone().requires_grad_(True)
    mll2 = exact_mll(U[perm].contiguous(), KernelSpec(w2, sf2.detach()), tau.detach(), mean, y[perm].contiguous())
    mll2.backward()
    assert abs(mll2.item() - mll.item()) <= 1e-9 * abs(mll.item())
    np.testing.assert_allclose(w2.grad.cpu().numpy(), g_w.cpu().numpy(), rtol=1e-6)
    # d/dtau = 0.5 (alpha'alpha - tr Ky^-1): check against the trace computed from Linv (||Linv||_F^2)
    fro = 0.0
    for r0 in range(0, N, 2000):
        blk = torch.tril(ws.Li[r0:r0 + 2000], diagonal=r0)
        fro += float((blk * blk).sum())
    expect = 0.5 * (float(alpha @ alpha) - fro)
    assert abs(g_tau.item() - expect) <= 1e-6 * abs(expect)


@pytest.mark.parametrize("cfg", ["C3", "C4"])
def test_full_size_mixed_and_multifidelity_models(gpu_ctx, cfg):
    """BASELINE.json's C3 (N=10000, two 5-level categorical inputs through the latent map) and C4 (N=15000, three
    sources with their own noise and mean) at FULL size through the GP_Plus API, checked by a size-independent property:
    the directional derivative of the loss along a random direction in parameter space, by central differences, equals
    grad . v.  C3 runs the look-ahead factorisation with the bordered inverse, C4 the one with pair merging."""
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.preprocessing import standard
    from gpplus_amd.test_functions.analytical import borehole_mixed_variables
    from gpplus_amd.test_functions.multi_fidelity import multi_fidelity_wing

    torch.manual_seed(0)
    if cfg == "C3":
        np.random.seed(4)
        qd = {0: 5, 5: 5}
        U, y = borehole_mixed_variables(n=10000, qual_dict=qd, random_state=4, shuffle=False)
        U, _, _ = standard(torch.as_tensor(U).double(), qd)
        m = GP_Plus(U, torch.tensor(y), qual_dict=qd, dtype=torch.float64, device="cuda")
    else:
        X, y = multi_fidelity_wing(n={"0": 5000, "1": 5000, "2": 5000}, noise_std={"0": 0.5, "1": 1.0, "2": 1.5},
                                   random_state=4)
        X, _, _ = standard(torch.tensor(X), {10: 3})
        m = GP_Plus(X, torch.tensor(y), qual_dict={10: 3}, multiple_noise=True, m_gp="multiple_constant",
                    dtype=torch.float64, device="cuda")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "raw_lengthscale" in n and p.requires_grad:
                p.fill_(-1.0)
            elif n == "covar_module.raw_outputscale":
                p.fill_(0.3)
            elif "raw_noise" in n:
                p.fill_(-6.0)
            elif n.endswith(".constant"):
                p.fill_(0.4 if n == "mean_module.constant" else 0.1)
    loss0, grads = loss_and_grads(m)
    assert np.isfinite(loss0) and all(np.isfinite(g).all() for g in grads.values())
    params = {n: p for n, p in m.named_parameters() if n in grads}
    gen = torch.Generator().manual_seed(1)
    v = {n: torch.randn(p.shape, generator=gen, dtype=torch.float64).to(p) for n, p in params.items()}
    dd = sum(float((torch.as_tensor(grads[n]).to(v[n]).cpu() * v[n].cpu()).sum()) for n in params)
    eps = 1e-4 if all(p.dtype == torch.float64 for p in params.values()) else 2e-3
    vals = []
    for sgn in (+1.0, -1.0):
        with torch.no_grad():
            for n, p in params.items():
                p.add_(sgn * eps * v[n])
        vals.append(loss_and_grads(m)[0])
        with torch.no_grad():
            for n, p in params.items():
                p.sub_(sgn * eps * v[n])
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - dd) <= 2e-4 * max(abs(dd), 1e-3), (fd, dd)


def test_fit_model_scipy_lbfgs(gpu_ctx):
    """the scipy multistart driver on the HIP back end; objective = -(log_prob + priors), not / N."""
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.optim import MLLObjective, fit_model_scipy
    from gpplus_amd.utils import set_seed

    fx = load("c1_borehole_n500.npz")
    m = build(fx, "theta1")
    m.train()
    obj = MLLObjective(m, True, [0, 0])
    theta = obj.pack_parameters()
    f, g = obj.fun(theta)
    o = OracleGP(fx["Xtrain"], fx["ytrain"])
    for k in list(o.params):
        o.params[k] = torch.as_tensor(fx[f"theta1::param::{k}"], dtype=torch.float64)
    lo, go = o.loss_and_grad(normalize=False)
    assert abs(f - lo.item()) <= RTOL_MLL * abs(lo.item())
    gref = np.concatenate([go[n].numpy().ravel() for n in obj.param_shapes])
    np.testing.assert_allclose(g, gref, rtol=RTOL_MLL, atol=RTOL_MLL * np.abs(gref).max())
    set_seed(3)
    res, best = fit_model_scipy(m, num_restarts=1, options={"maxiter": 25}, bounds=True)
    assert len(res) == 2 and np.isfinite(best) and best < f
    f_after, _ = MLLObjective(m, True, [0, 0]).fun(MLLObjective(m, True, [0, 0]).pack_parameters())
    assert abs(f_after - best) <= 1e-8 * abs(best)   # the model holds the best start's parameters


@pytest.mark.parametrize("kclass", ["Matern32Kernel", "Matern52Kernel"])
@pytest.mark.parametrize("mixed", [False, True])
def test_matern_models_against_oracle(gpu_ctx, kclass, mixed):
    """Matern 3/2 and 5/2 quantitative kernels (alone, and times the RBF manifold kernel)."""
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(11)
    n = 400
    X = rng.standard_normal((n, 5))
    kw = {}
    if mixed:
        X[:, 2] = rng.integers(0, 4, n)
        kw = {"qual_dict": {2: 4}}
    y = np.sin(X[:, 0]) + 0.2 * X[:, 1] + 0.1 * X[:, 2]
    o = OracleGP(X, y, quant_correlation_class=kclass, seed=2, **kw)
    o.params[o.ls_key] = torch.as_tensor(np.float32(rng.uniform(-1.0, 0.0, o.params[o.ls_key].shape)), dtype=torch.float64)
    o.params["likelihood.noise_covar.raw_noise"] = torch.tensor([-5.0], dtype=torch.float64)
    lo, go = o.loss_and_grad()
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda", quant_correlation_class=kclass, **kw)
    sd = m.state_dict()
    for k, v in o.params.items():
        sd[k] = v.reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    loss, grads = loss_and_grads(m)
    assert abs(loss - lo.item()) <= RTOL_MLL * abs(lo.item())
    for k, g in go.items():
        gref = g.numpy().reshape(grads[k].shape)
        np.testing.assert_allclose(grads[k], gref, rtol=RTOL_MLL, atol=RTOL_MLL * max(np.abs(gref).max(), 1e-12), err_msg=k)
    mean, std = m.predict(torch.tensor(X[:50] + 0.05), return_std=True, include_noise=True)
    om, os_ = o.predict(X[:50] + 0.05 if not mixed else np.column_stack([X[:50, :2] + 0.05, X[:50, 2], X[:50, 3:] + 0.05]))
    if not mixed:
        np.testing.assert_allclose(mean.cpu().numpy(), om.numpy(), rtol=RTOL_PRED, atol=1e-7)
        np.testing.assert_allclose(std.cpu().numpy(), os_.numpy(), rtol=RTOL_PRED, atol=1e-7)


def test_loocv_and_noise_continuation(gpu_ctx):
    """LOOCV error from the cached factorisation (+ gpp_lauum for diag(Ky^-1)) against dense numpy
    algebra on the oracle's covariance, and the noise-continuation driver (optim/mll_noise_continuation.py:45-244)."""
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.optim import MLLObjective, fit_model_continuation, loocv_rrmse
    from gpplus_amd.utils import set_seed

    fx = load("c1_borehole_n500.npz")
    X, y = fx["Xtrain"][:160], fx["ytrain"][:160]
    sub = {k: v for k, v in fx.items()}
    sub["Xtrain"], sub["ytrain"] = X, y
    m = build(sub, "theta1")
    # LOOCV: e_i = (Ky^-1 r)_i / (Ky^-1)_ii on the oracle's dense covariance
    o = OracleGP(X, y)
    for k in list(o.params):
        o.params[k] = torch.as_tensor(fx[f"theta1::param::{k}"], dtype=torch.float64)
    with torch.no_grad():
        mean, cov = o.forward(o.train_x)
        Ky = (cov + torch.diag(o.noise_vector(o.train_x))).numpy()
        r = (o.y_sc - mean).numpy()
    Kinv = np.linalg.inv(Ky)
    ref = float(np.sqrt(np.mean(((Kinv @ r) / np.diag(Kinv)) ** 2)))
    got = loocv_rrmse(m)
    assert abs(got - ref) <= 1e-7 * ref, (got, ref)

    # continuation: noise fixed at each level, decreasing levels, model left at the best level's optimum
    set_seed(5)
    m.train()
    nll, hist = fit_model_continuation(m, num_restarts=0, options={"maxiter": 15}, initial_noise_var=1.0, verbose=False)
    assert not m.likelihood.raw_noise.requires_grad
    levels = [float(v.reshape(-1)[0]) for v in hist["noise_history"]]
    assert len(levels) >= 2 and all(a > b for a, b in zip(levels, levels[1:]))
    assert len(hist["nll_history"]) == len(levels) == len(hist["optimization_history"]) and hist["loocv_history"][0] == "NLL"
    assert nll == min(hist["nll_history"])
    obj = MLLObjective(m, True, [0, 0])
    f_now = obj.fun(obj.pack_parameters(), return_grad=False)
    assert np.isfinite(f_now)


def test_noise_continuation_follows_the_oracle_driver(gpu_ctx):
    """f3 against a restatement, not structurally: ``fit_model_continuation`` on the HIP back end and
    ``oracle_continuation`` (optim/mll_noise_continuation.py:45-244 on the oracle's scipy objective) from the same seed — the same
    start points (a12), the same fixed noise levels, and NLLs that agree level by level."""
    from oracle.gp_oracle import OracleGP, oracle_continuation
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.optim import fit_model_continuation

    rng = np.random.default_rng(21)
    n = 96
    X = rng.standard_normal((n, 3))
    y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
    m.train()
    torch.manual_seed(6)
    nll, hist = fit_model_continuation(m, num_restarts=1, initial_noise_var=1.0, verbose=False)
    o = OracleGP(X, y)
    torch.manual_seed(6)
    nll_o, hist_o = oracle_continuation(o, num_restarts=1, initial_noise_var=1.0)
    lv = [float(v.reshape(-1)[0]) for v in hist["noise_history"]]
    np.testing.assert_allclose(lv, hist_o["noise_history"], rtol=1e-10)
    # (two L-BFGS-B runs on objectives that agree to ~1e-11 stop within their own tolerance of each other: ftol = 1e-6 relative)
    np.testing.assert_allclose(hist["nll_history"], hist_o["nll_history"], rtol=2e-5)
    assert abs(nll - nll_o) <= 2e-5 * abs(nll_o)
    # the model is left at the selected level's noise (:239); the other parameters are optimiser outputs, equal to its tolerance only
    k = "likelihood.noise_covar.raw_noise"
    np.testing.assert_allclose(m.state_dict()[k].detach().cpu().double().reshape(-1).numpy(), o.params[k].reshape(-1).numpy(), rtol=1e-10)


def test_scipy_objective_with_the_references_fp32_theta(gpu_ctx):
    """``settings.reference_fp32_theta(True)`` (optim/mll_scipy.py:32-35,97): the objective at theta equals the default mode's
    objective at float32(theta) bit for bit — eager and replayed — and the oracle's fp32 mode agrees."""
    from oracle.gp_oracle import OracleGP, _pack, _unpack
    from gpplus_amd import settings
    from gpplus_amd.optim import MLLObjective

    fx = load("c1_borehole_n500.npz")
    m = build(fx, "theta1")
    m.train()
    x = MLLObjective(m, True, [0, 0]).pack_parameters() + 1e-9
    x32 = x.astype(np.float32).astype(np.float64)
    for graphed in (False, True):
        with settings.graphed_objective(graphed):
            f_ref, g_ref = MLLObjective(m, True, [0, 0]).fun(x32)
            with settings.reference_fp32_theta(True):
                f, g = MLLObjective(m, True, [0, 0]).fun(x)
            assert f == f_ref and np.array_equal(g, g_ref), (graphed, f, f_ref)
            f64, _ = MLLObjective(m, True, [0, 0]).fun(x)
            assert f64 != f_ref  # (1e-9 off a float32 grid point is visible in fp64)
    o = OracleGP(fx["Xtrain"], fx["ytrain"])
    names = [n for n in MLLObjective(m, True, [0, 0]).param_shapes]
    assert sorted(names) == sorted(o.trainable)
    _unpack(o, names, x, fp32_theta=True)
    lo = o.loss(normalize=False).item()
    assert abs(f - lo) <= RTOL_MLL * abs(lo)
    np.testing.assert_array_equal(_pack(o, names), x32)


def test_bayesian_optimisation_consumers(gpu_ctx):
    """acquisition functions and the cost-aware multi-fidelity BO loop (bayesian_optimizations/*) on the
    HIP-backed model: one iteration of the continuous branch and one of the pool branch, plus EI against its closed form."""
    from gpplus_amd.bayesian_optimizations import AF_EI, AF_HF, AF_LF, BO
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.utils import set_seed
    from scipy.stats import norm

    set_seed(11)
    rng = np.random.default_rng(3)

    def truth(x, s):  # two sources: the cheap one is a biased version of the expensive one
        return np.sin(3.0 * x) + 0.5 * x + (0.3 * np.cos(2.0 * x) if s == 1 else 0.0)

    xs = rng.uniform(-2.0, 2.0, 24)
    src = np.array([0] * 8 + [1] * 16)
    xmean, xstd = np.array([xs.mean()]), np.array([xs.std()])
    Xtr = np.stack([(xs - xmean[0]) / xstd[0], src.astype(float)], axis=1)
    ytr = np.array([truth(x, s) for x, s in zip(xs, src)])
    costs = {"0": 10.0, "1": 1.0}

    # acquisition functions against their closed forms on one model
    m = GP_Plus(torch.tensor(Xtr), torch.tensor(ytr), qual_dict={1: 2}, dtype=torch.float64, device="cuda")
    m.eval()
    pt = np.array([0.3, 1.0])
    with torch.no_grad():
        mu, sd = m.predict(torch.tensor([[(0.3 - xmean[0]) / xstd[0], 1.0]]), return_std=True, include_noise=True)
    mu, sd = float(mu), float(sd)
    best = float(ytr.min())
    u = -(mu - best) / sd
    cf = lambda v: costs[str(int(v))]
    assert abs(AF_EI(pt, best, m, xmean, xstd, cf) + sd * (norm.pdf(u) + u * norm.cdf(u)) / 1.0) < 1e-9
    assert abs(AF_LF(pt, best, m, xmean, xstd, cf) + sd * norm.pdf(u) / 1.0) < 1e-9
    assert abs(AF_HF(pt, best, m, xmean, xstd, cf) + sd * u / 1.0) < 1e-9

    def gen(_flag, x):  # x: (1, 2) raw coordinate + source
        x = np.asarray(x, dtype=float).reshape(-1, 2)
        return torch.tensor([truth(r[0], int(round(r[1]))) for r in x])

    bestf, cum = BO(Xtrain=torch.tensor(Xtr), ytrain=torch.tensor(ytr), costs=costs, l_bound=[-2.0], u_bound=[2.0], xmean=xmean,
                    xstd=xstd, qual_index={1: 2}, data_gen_func=gen, one_iter=True, max_cost=1e9, n_starts=2,
                    fit_options={"maxiter": 10})
    assert bestf.shape == (2,) and cum.shape == (2,) and (cum[1] - cum[0]) in (10.0, 1.0)
    assert bestf[1] <= bestf[0] + 1e-12   # the incumbent (high-fidelity minimum) never gets worse

    pool_x = rng.uniform(-2.0, 2.0, 60)
    pool_s = np.array([0] * 20 + [1] * 40)
    pool = np.stack([(pool_x - xmean[0]) / xstd[0], pool_s.astype(float), [truth(x, s) for x, s in zip(pool_x, pool_s)]], axis=1)
    bestf2, cum2 = BO(costs=costs, qual_index={1: 2}, data_gen_func=pool, n_train=[6, 10], one_iter=True, max_cost=1e9,
                      fit_options={"maxiter": 10})
    assert bestf2.shape == (2,) and (cum2[1] - cum2[0]) in (10.0, 1.0)


def test_bo_consumers_match_the_oracle(gpu_ctx):
    """SURVEY.md §8 f4 against the ORACLE (oracle/gp_oracle.py: oracle_af, oracle_bo_pool_scores, oracle_sobol, restated from
    bayesian_optimizations/AFs.py:1-159, BO_GP_plus.py:183-197 and models/gp_plus.py:1148-1224 on top of OracleGP.predict): on a
    seeded two-source problem with the SAME raw parameters in both, the HIP model's acquisition values, the pool branch's scores and
    chosen candidate, and the Sobol indices on the same low-discrepancy sample agree with the oracle's."""
    import warnings as _w

    from gpplus_amd.bayesian_optimizations import AF_EI, AF_HF, AF_LF
    from gpplus_amd.bayesian_optimizations.AFs import AF_HF_Engineering, AF_LF_Engineering
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.utils import set_seed
    from oracle.gp_oracle import OracleGP, oracle_af, oracle_bo_pool_scores, oracle_sobol
    from scipy.stats import qmc

    set_seed(7)
    rng = np.random.default_rng(11)

    def truth(x1, x2, s):
        return np.sin(3.0 * x1) + 0.5 * x2 * x2 + (0.3 * np.cos(2.0 * x1) + 0.1 if s == 1 else 0.0)

    n = 60
    raw = rng.uniform(-2.0, 2.0, (n, 2))
    src = np.array([0] * 20 + [1] * 40)
    xmean, xstd = raw.mean(0), raw.std(0)
    Xtr = np.concatenate([(raw - xmean) / xstd, src[:, None].astype(float)], axis=1)
    ytr = np.array([truth(a, b, s) for (a, b), s in zip(raw, src)])
    m = GP_Plus(torch.tensor(Xtr), torch.tensor(ytr), qual_dict={2: 2}, dtype=torch.float64, device="cuda")
    o = OracleGP(Xtr, ytr, qual_dict={2: 2})
    sd = m.state_dict()
    vals = {"raw_lengthscale": -0.4, "raw_outputscale": 0.5, "raw_noise": -5.0, "constant": 0.2}
    for k in list(o.params):
        for frag, v in vals.items():
            if frag in k:
                sd[k] = torch.full_like(sd[k], v)
        o.params[k] = sd[k].detach().cpu().double().reshape(o.params[k].shape).clone()
    m.load_state_dict(sd)
    m.eval()
    costs = {"0": 10.0, "1": 1.0}
    cf = lambda v: costs[str(int(v))]  # noqa: E731
    best = float(ytr[src == 0].min())
    # point-wise acquisition functions (the continuous branch's objective, BO_GP_plus.py:68)
    for pt in ([0.3, -1.1, 1.0], [-1.7, 0.4, 0.0], [1.9, 1.9, 1.0], [0.0, 0.0, 0.0], [-0.6, 1.3, 1.0]):
        pt = np.array(pt)
        for fn, kind in ((AF_LF, "LF"), (AF_HF, "HF"), (AF_EI, "EI")):
            for maximize in (False, True):
                got = fn(pt, best, m, xmean, xstd, cf, maximize=maximize, si=0.01)
                ref = oracle_af(kind, pt, best, o, xmean, xstd, cf, maximize=maximize, si=0.01)
                assert abs(got - ref) <= 1e-6 * max(abs(ref), 1e-6), (kind, pt, maximize, got, ref)
    # one selection of the pool branch (BO_GP_plus.py:183-197): scores per source, concatenated, argmax
    praw = rng.uniform(-2.0, 2.0, (90, 2))
    psrc = np.array([0] * 30 + [1] * 60)
    pool = np.concatenate([(praw - xmean) / xstd, psrc[:, None].astype(float),
                           np.array([truth(a, b, s) for (a, b), s in zip(praw, psrc)])[:, None]], axis=1)
    best_values = [float(ytr[src == i].min()) for i in range(2)]
    ref_scores, ref_idx = oracle_bo_pool_scores(o, pool, best_values, cf, 2, maximize=False)
    scores = []
    for i in range(2):
        cand = torch.tensor(pool[pool[:, -2] == i][:, 0:-1])
        with torch.no_grad():
            ytest, ystd = m.predict(cand, return_std=True, include_noise=False)
        af = AF_HF_Engineering if i == 0 else AF_LF_Engineering
        scores.append(af(best_values[i], ytest.reshape(-1, 1), ystd.reshape(-1, 1), cand, cf, maximize=False).reshape(-1))
    scores = torch.cat(scores, dim=0)
    np.testing.assert_allclose(scores.cpu().numpy(), ref_scores.numpy(), rtol=1e-6, atol=1e-9)
    assert int(torch.argmax(scores)) == ref_idx
    # Sobol indices on the same sample
    Nq = 512
    gen = qmc.Sobol(d=6, scramble=False)
    gen.fast_forward(1)
    S_ref, ST_ref = oracle_sobol(o, gen.random(Nq), [2])
    with _w.catch_warnings():
        _w.simplefilter("ignore")
        S, ST = m.Sobol(N=Nq, batch=200)
    np.testing.assert_allclose(S, S_ref, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ST, ST_ref, rtol=1e-6, atol=1e-8)


def test_sobol_indices(gpu_ctx):
    """Sobol indices from p + 2 batched predictions (gp_plus.py:1148-1224).  For an additive truth
    y = 2 x0 + x1^2 + (level effect of a categorical x2) the analytic indices follow from the term variances."""
    import warnings as _w
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.optim import fit_model_scipy
    from gpplus_amd.utils import set_seed

    set_seed(2)
    rng = np.random.default_rng(0)
    n = 300
    X = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.integers(0, 3, n).astype(float)], axis=1)
    lev = np.array([0.0, 0.3, 0.6])
    y = 2.0 * X[:, 0] + X[:, 1] ** 2 + lev[X[:, 2].astype(int)]
    m = GP_Plus(torch.tensor(X), torch.tensor(y), qual_dict={2: 3}, dtype=torch.float64, device="cuda")
    fit_model_scipy(m, num_restarts=3, options={"maxiter": 60}, bounds=True)
    with _w.catch_warnings():
        _w.simplefilter("ignore")
        S, ST = m.Sobol(N=8192, batch=3000)
    assert S.shape == (1, 3) and ST.shape == (1, 3)
    v = np.array([4.0 / 12.0, 4.0 / 45.0, np.var(lev)])      # Var[2 x0], Var[x1^2], Var[level effect] (uniform levels)
    ref = v / v.sum()
    np.testing.assert_allclose(S[0], ref, atol=0.05)
    np.testing.assert_allclose(ST[0], ref, atol=0.05)          # additive: total = main


def test_predict_repeated_batches_with_categoricals(gpu_ctx):
    """Regression: consecutive predictions on different same-shaped batches (which the caching allocator places at the
    same address) must each encode their OWN categorical levels."""
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(5)
    n = 120
    X = np.stack([rng.uniform(0, 1, n), rng.integers(0, 3, n).astype(float)], axis=1)
    y = np.sin(4 * X[:, 0]) + np.array([0.0, 1.0, 2.0])[X[:, 1].astype(int)]
    m = GP_Plus(torch.tensor(X), torch.tensor(y), qual_dict={1: 3}, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        m.likelihood.initialize(noise=1e-4)
    m.eval()
    outs = []
    for lvl in (0.0, 1.0, 2.0, 0.0):
        Z = torch.tensor([[0.25, lvl], [0.75, lvl]])
        outs.append(m.predict(Z, return_std=False).cpu().numpy().copy())
    together = m.predict(torch.tensor([[0.25, 0.0], [0.25, 1.0], [0.25, 2.0]]), return_std=False).cpu().numpy()
    np.testing.assert_allclose([o[0] for o in outs[:3]], together, rtol=1e-10)
    np.testing.assert_allclose(outs[3], outs[0], rtol=1e-12)
    assert abs(outs[1][0] - outs[0][0]) > 0.1 and abs(outs[2][0] - outs[1][0]) > 0.1   # the level effect is visible


def test_two_models_share_the_prediction_workspace(gpu_ctx):
    """Regression: two models of the same size predicting alternately must not read each other's cached factor."""
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(9)
    X = rng.uniform(0, 1, (90, 2))
    ya, yb = np.sin(5 * X[:, 0]), np.cos(3 * X[:, 1]) + 2.0
    ma = GP_Plus(torch.tensor(X), torch.tensor(ya), dtype=torch.float64, device="cuda")
    mb = GP_Plus(torch.tensor(X), torch.tensor(yb), dtype=torch.float64, device="cuda")
    for m in (ma, mb):
        with torch.no_grad():
            m.likelihood.initialize(noise=1e-4)
        m.eval()
    Z = torch.tensor(rng.uniform(0, 1, (7, 2)))
    a1 = ma.predict(Z, return_std=False).cpu().numpy().copy()
    b1 = mb.predict(Z, return_std=False).cpu().numpy().copy()
    a2 = ma.predict(Z, return_std=False).cpu().numpy().copy()
    np.testing.assert_allclose(a2, a1, rtol=1e-12)
    assert np.abs(a1 - b1).max() > 0.5
    # the advisor's sequence: the LAZY variance of an earlier prediction is read after another model of the same size has
    # predicted (p1 = m1(x); p2 = m2(x); p1.stddev) — works in gpytorch, whose prediction strategy owns its caches
    ref_a = ma(Z).stddev.cpu().numpy().copy()
    ref_b = mb(Z).stddev.cpu().numpy().copy()
    p1 = ma(Z)
    p2 = mb(Z)
    np.testing.assert_allclose(p1.stddev.cpu().numpy(), ref_a, rtol=1e-12)
    np.testing.assert_allclose(p2.stddev.cpu().numpy(), ref_b, rtol=1e-12)
    np.testing.assert_allclose(p1.covariance_matrix.diagonal().sqrt().cpu().numpy(), ref_a, rtol=1e-9)


def _toy_model(kind, n=260, seed=0):
    from gpplus_amd.models import GP_Plus

    rng = np.random.default_rng(seed)
    if kind == "plain":
        X = rng.uniform(0, 1, (n, 4))
        y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2
        return GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
    X = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.integers(0, 3, n).astype(float)], 1)
    y = np.sin(3 * X[:, 0]) + X[:, 1] + 0.3 * X[:, 2]
    kw = dict(multiple_noise=True, m_gp="multiple_constant") if kind == "multi_fidelity" else {}
    return GP_Plus(torch.tensor(X), torch.tensor(y), qual_dict={2: 3}, dtype=torch.float64, device="cuda", **kw)


@pytest.mark.parametrize("kind", ["plain", "mixed", "multi_fidelity"])
def test_batched_objective_matches_the_model(gpu_ctx, kind):
    """optim.BatchedObjective (all restarts in one batched evaluation: vmap over the model's own forward + priors, then
    the *_batched kernels) against the model evaluated run by run: loss and every gradient."""
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.optim import BatchedObjective
    from gpplus_amd.utils import set_seed

    set_seed(1)
    m = _toy_model(kind)
    B = 5
    obj = BatchedObjective(m, B)
    obj.sample_restarts()
    loss = obj.loss()
    loss.sum().backward()
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    for b in range(B):
        st = m.state_dict()
        st.update(obj.row(b))
        m.load_state_dict(st)
        m.train()
        for p in m.parameters():
            p.grad = None
        one = -mll(m(*m.train_inputs), m.train_targets)
        one.backward()
        assert abs(one.item() - loss[b].item()) <= 1e-10 * abs(one.item())
        for name, p in m.named_parameters():
            if p.requires_grad:
                np.testing.assert_allclose(obj.theta[name].grad[b].cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-8,
                                           atol=1e-10 * float(p.grad.abs().max()) + 1e-14, err_msg=name)


def test_fit_model_torch_batched_follows_the_sequential_driver(gpu_ctx):
    """optim/mll_torch.py:99-141 with all runs advancing together: same starts (same RNG order), same Adam trajectories,
    same winner as ``fit_model_torch``."""
    from gpplus_amd.optim import fit_model_torch, fit_model_torch_batched
    from gpplus_amd.utils import set_seed

    set_seed(5)
    ma = _toy_model("mixed", n=200, seed=3)
    set_seed(5)
    mb = _toy_model("mixed", n=200, seed=3)
    set_seed(9)
    fa, ha = fit_model_torch(ma, num_restarts=3, num_iter=30, verbose=False)
    set_seed(9)
    fb, hb = fit_model_torch_batched(mb, num_restarts=3, num_iter=30)
    assert len(ha) == len(hb) == 4
    for a, b in zip(ha, hb):
        np.testing.assert_allclose(b, a, rtol=1e-7, atol=1e-9)
    assert abs(fa - fb) <= 1e-7 * abs(fa)
    for (na, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        if torch.is_tensor(pa) and pa.dtype.is_floating_point:
            np.testing.assert_allclose(pb.cpu().numpy(), pa.cpu().numpy(), rtol=1e-6, atol=1e-8, err_msg=na)


def test_fit_model_torch_batched_stops_runs_like_the_sequential_driver(gpu_ctx):
    """The reference's early stop (optim/mll_torch.py:126-128, a float32 window mean on the host) ends each run of the
    batched driver at the same iteration as in the sequential driver."""
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.optim import fit_model_torch, fit_model_torch_batched
    from gpplus_amd.utils import set_seed

    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (150, 3))
    y = np.sin(3 * X[:, 0]) + X[:, 1] + 0.05 * rng.standard_normal(150)
    out = []
    for fit in (fit_model_torch_batched, fit_model_torch):
        set_seed(1)
        m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
        set_seed(4)
        kw = {} if fit is fit_model_torch_batched else {"verbose": False}
        out.append(fit(m, num_restarts=5, num_iter=260, break_steps=50, lr_default=0.3, **kw))
    (fb, hb), (fs, hs) = out
    assert [len(h) for h in hb] == [len(h) for h in hs]
    assert any(len(h) < 260 for h in hs)            # the scenario does exercise the early stop
    assert abs(fb - fs) <= 1e-8 * abs(fs)


def test_batched_driver_replayed_graph_equals_the_eager_loop(gpu_ctx):
    """``fit_model_torch_batched`` evaluates loss + gradients of every Adam step by replaying ONE captured HIP graph
    (optim/mll_batched.py::_GraphedLossAndGrad); with ``settings.graphed_objective(False)`` the same launches are issued
    eagerly.  Same histories bit for bit, same winner, every step served by the graph — also for a model whose latent map makes
    the features trainable, and with runs stopping early (the ``active`` mask is an input of the graph)."""
    from gpplus_amd import settings
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.optim import fit_model_torch_batched
    from gpplus_amd.utils import set_seed

    def fit(on, kind):
        set_seed(5)
        if kind == "mixed":
            m = _toy_model("mixed", n=200, seed=3)
            kw = dict(num_restarts=3, num_iter=40)
        else:
            rng = np.random.default_rng(0)
            X = rng.uniform(0, 1, (150, 3))
            y = np.sin(3 * X[:, 0]) + X[:, 1] + 0.05 * rng.standard_normal(150)
            m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda")
            kw = dict(num_restarts=5, num_iter=260, break_steps=50, lr_default=0.3)
        set_seed(9)
        with settings.graphed_objective(on):
            f, h = fit_model_torch_batched(m, **kw)
        g = fit_model_torch_batched.last_graph
        return f, h, m.state_dict(), g

    for kind in ("mixed", "early-stop"):
        f1, h1, s1, g1 = fit(True, kind)
        f0, h0, s0, g0 = fit(False, kind)
        assert g0 is None and g1 is not None and g1.replays > 0 and g1.declined == 0
        assert f1 == f0 and h1 == h0
        for k, v in s0.items():
            if torch.is_tensor(v):
                assert torch.equal(s1[k], v), k
