"""Pins the CPU oracle (oracle/gp_oracle.py).  The reference pins nothing for this path (no tests, no stored outputs,
gpytorch absent), so the oracle is held by: closed forms, an independent numpy/scipy evaluation of the same formulas,
the analytic-gradient identity, central finite differences, torch.distributions for the priors, and the committed
golden fixtures whose INPUTS come from the reference's own data pipeline (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest
import scipy.linalg as sla
import torch

from oracle import gp_oracle as G

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def model_from_fixture(fx, tag, **kw):
    xkey = "Xtrain" if "Xtrain" in fx else "Utrain"
    o = G.OracleGP(fx[xkey], fx["ytrain"], **kw)
    for k in list(o.params):
        o.params[k] = torch.as_tensor(fx[f"{tag}::param::{k}"], dtype=torch.float64)
    return o


def numpy_mll(o: G.OracleGP):
    """Independent evaluation: direct-difference kernel + scipy Cholesky (no torch, no skinny-GEMM distance)."""
    p = {k: v.numpy() for k, v in o.params.items()}
    X = o.train_x.numpy()
    U = X[:, o.quant_index]
    if o.kclass == "Rough_RBF":
        w = 10.0 ** p[o.ls_key].reshape(-1)
    else:
        w = 1.0 / (2 * np.exp(2 * p[o.ls_key].reshape(-1)))
    if o.qual_cols:
        z = o.zeta.numpy()[o.cat_index.numpy()] @ p[o.latent_key].T
        U = np.hstack([z, U])
        w = np.concatenate([np.full(o.dz, 0.5), w])
    d2 = ((U[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    sf2 = np.log1p(np.exp(p["covar_module.raw_outputscale"]))
    tau = np.exp(p["likelihood.noise_covar.raw_noise"]) + o.lb_noise
    if o.noise_indices:
        tau_i = tau[X[:, -1].astype(int)]
    else:
        tau_i = np.full(len(X), tau[0])
    m = o.mean(o.train_x).numpy()
    Ky = sf2 * np.exp(-d2) + np.diag(tau_i)
    cf = sla.cho_factor(Ky, lower=True)
    r = o.y_sc.numpy() - m
    alpha = sla.cho_solve(cf, r)
    mll = -0.5 * (r @ alpha + 2 * np.log(np.diag(cf[0])).sum() + len(r) * math.log(2 * math.pi))
    return mll, Ky, alpha, U, w, sf2


def test_closed_form_n1():
    o = G.OracleGP(np.array([[0.3, -1.0]]), np.array([2.0]))
    # y-scaling with one point divides by zero in the reference too (gpregression.py:67-69); bypass with explicit y_sc
    o.y_sc = torch.tensor([0.7], dtype=torch.float64)
    o.params["mean_module.constant"] = torch.tensor([0.2], dtype=torch.float64)
    sf2, tau = math.log(2.0), 1.0 + 1e-8
    r = 0.7 - 0.2
    ref = -0.5 * (r * r / (sf2 + tau) + math.log(sf2 + tau) + math.log(2 * math.pi))
    assert abs(o.mll().item() - ref) < 1e-14


def test_closed_form_duplicate_rows():
    # two identical rows: K = sf2 * ones(2,2); PD only through the noise
    X = np.array([[0.5, 1.0, -0.2], [0.5, 1.0, -0.2], [0.1, 0.0, 0.3]])
    o = G.OracleGP(X[:2], np.array([1.0, 3.0]))
    o.params["likelihood.noise_covar.raw_noise"] = torch.tensor([-3.0], dtype=torch.float64)
    s, t = math.log(2.0), math.exp(-3.0) + 1e-8
    y = np.array([0.0, 1.0])
    det = (s + t) ** 2 - s * s
    Kinv = np.array([[s + t, -s], [-s, s + t]]) / det
    ref = -0.5 * (y @ Kinv @ y + math.log(det) + 2 * math.log(2 * math.pi))
    assert abs(o.mll().item() - ref) < 1e-12


def test_w_to_zero_sherman_morrison():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((40, 3))
    o = G.OracleGP(X, rng.standard_normal(40), m_gp="single_zero")
    o.params[o.ls_key] = torch.full((1, 3), -40.0, dtype=torch.float64)  # w = 1e-40 -> K = sf2 * 11^T
    s, t, n = math.log(2.0), 1.0 + 1e-8, 40
    y = o.y_sc.numpy()
    quad = (y @ y) / t - s * y.sum() ** 2 / (t * (t + n * s))
    logdet = (n - 1) * math.log(t) + math.log(t + n * s)
    ref = -0.5 * (quad + logdet + n * math.log(2 * math.pi))
    assert abs(o.mll().item() - ref) < 1e-9 * abs(ref)


@pytest.mark.parametrize("fixture,tag,kw", [
    ("c1_borehole_n500.npz", "theta0", {}),
    ("c1_borehole_n500.npz", "theta1", {}),
    ("c3_borehole_mixed_n100.npz", "theta1", {"qual_dict": {0: 5, 5: 5}}),
    ("c4_wing_mf_n300.npz", "theta1", {"qual_dict": {10: 3}, "multiple_noise": True, "m_gp": "multiple_constant"}),
])
def test_against_numpy_scipy_and_golden(fixture, tag, kw):
    fx = load(fixture)
    o = model_from_fixture(fx, tag, **kw)
    mll_np, Ky, alpha, U, w, sf2 = numpy_mll(o)
    mll = o.mll().item()
    assert abs(mll - mll_np) <= 1e-10 * abs(mll_np)            # GEMM-distance vs direct-difference, torch vs scipy
    assert abs(mll - fx[f"{tag}::mll"]) <= 1e-12 * abs(mll)     # committed fixture
    loss, grads = o.loss_and_grad()
    assert abs(loss.item() - fx[f"{tag}::loss"]) <= 1e-12 * abs(loss.item())
    for k, g in grads.items():
        np.testing.assert_allclose(g.numpy(), fx[f"{tag}::grad::{k}"], rtol=1e-9, atol=1e-13)
    # analytic identity dMLL/dtheta = sum W * dKy/dtheta, W = 0.5 (alpha alpha^T - Ky^-1), for the kernel weights
    Kinv = sla.cho_solve(sla.cho_factor(Ky, lower=True), np.eye(len(Ky)))
    W = 0.5 * (np.outer(alpha, alpha) - Kinv)
    Kc = Ky - np.diag(np.diag(Ky)) + np.diag(np.full(len(Ky), sf2))
    dq = len(o.quant_index)
    g_omega = np.array([(W * Kc * (-(U[:, None, o.dz + d] - U[None, :, o.dz + d]) ** 2)).sum() * math.log(10) * w[o.dz + d]
                        for d in range(dq)])
    prior_g = -(o.params[o.ls_key].numpy().reshape(-1) + 3.0) / 9.0  # d/domega log N(-3, 3)
    expect = -(g_omega + prior_g) / o.N
    np.testing.assert_allclose(grads[o.ls_key].numpy().reshape(-1), expect, rtol=1e-7, atol=1e-12)


def test_finite_differences_every_parameter():
    fx = load("c4_wing_mf_n300.npz")
    o = model_from_fixture(fx, "theta1", qual_dict={10: 3}, multiple_noise=True, m_gp="multiple_constant")
    _, grads = o.loss_and_grad()
    h = 1e-5
    for k in o.trainable:
        flat = o.params[k].reshape(-1)
        g = grads[k].reshape(-1)
        for i in range(flat.numel()):
            old = flat[i].item()
            flat[i] = old + h
            fp = o.loss().item()
            flat[i] = old - h
            fm = o.loss().item()
            flat[i] = old
            fd = (fp - fm) / (2 * h)
            assert abs(fd - g[i].item()) <= 2e-6 * max(1.0, abs(fd)), (k, i, fd, g[i].item())


def test_priors_against_torch_distributions():
    x = torch.linspace(-4, 3, 11, dtype=torch.float64)
    np.testing.assert_allclose(G.normal_log_prob(x, -3.0, 3.0), torch.distributions.Normal(torch.tensor(-3.0, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64)).log_prob(x), rtol=1e-13)
    xp = x.exp()
    np.testing.assert_allclose(G.lognormal_log_prob(xp, 1e-6, 1.0), torch.distributions.LogNormal(torch.tensor(1e-6, dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64)).log_prob(xp), rtol=1e-13)
    a, b = math.log(0.1), math.log(10)
    inside = G.mollified_uniform_log_prob(torch.tensor([0.0], dtype=torch.float64), a, b)
    expect_in = torch.distributions.Normal(torch.tensor(0.0, dtype=torch.float64), torch.tensor(0.1, dtype=torch.float64)).log_prob(torch.tensor(0.0, dtype=torch.float64)) - math.log(1 + (b - a) / (math.sqrt(2 * math.pi) * 0.1))
    assert abs(inside.item() - expect_in.item()) < 1e-13
    outside = G.mollified_uniform_log_prob(torch.tensor([b + 0.25], dtype=torch.float64), a, b)
    assert abs((inside - outside).item() - 0.5 * (0.25 / 0.1) ** 2) < 1e-12
    raw = torch.tensor([-6.0, 0.0, 2.0], dtype=torch.float64)
    hs = G.log_half_horseshoe_log_prob(raw, 0.01, 1e-8)
    np.testing.assert_allclose(hs, np.log(np.log(1 + 3 * (0.01 / (1e-8 + np.exp(raw.numpy()))) ** 2)) + raw.numpy(), rtol=1e-13)


def test_transforms():
    x = torch.tensor([-3.0, 0.0, 2.5], dtype=torch.float64)
    np.testing.assert_allclose(G.inv_softplus(G.softplus(x)), x, rtol=1e-12)
    assert abs(G.softplus(torch.tensor(0.0, dtype=torch.float64)).item() - math.log(2)) < 1e-15
    # Rough lengthscale <-> weight: 1/(2 l^2) = 10^omega  
    om = torch.tensor([-1.0, 0.0, 1.5], dtype=torch.float64)
    np.testing.assert_allclose(1 / (2 * G.rough_lengthscale(om) ** 2), 10.0 ** om.numpy(), rtol=1e-13)


def test_literal_moment_matching_is_a_noop():
    fx = load("c1_borehole_n500.npz")
    o = model_from_fixture(fx, "theta1")
    a, b = o.mll().item(), o.mll(literal_moment_matching=True).item()
    assert abs(a - b) <= 1e-9 * abs(a)


def test_jitter_retry_and_errors():
    A = torch.tensor([[1.0, 1.0], [1.0, 1.0 - 1e-12]], dtype=torch.float64)
    with pytest.warns(RuntimeWarning):
        L, jit = G.psd_safe_cholesky(A)
    assert jit == 1e-8
    with pytest.raises(G.NotPSDError):
        G.psd_safe_cholesky(torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=torch.float64))
    with pytest.raises(G.NanError):
        G.psd_safe_cholesky(torch.tensor([[float("nan"), 0.0], [0.0, 1.0]], dtype=torch.float64))


def test_predict_matches_numpy():
    fx = load("c1_borehole_n500.npz")
    o = model_from_fixture(fx, "theta1")
    mean, std = o.predict(fx["Xtest"], return_std=True, include_noise=True)
    np.testing.assert_allclose(mean.numpy(), fx["theta1::pred_mean"], rtol=1e-11)
    np.testing.assert_allclose(std.numpy(), fx["theta1::pred_std"], rtol=1e-9)
    _, Ky, alpha, U, w, sf2 = numpy_mll(o)
    Xs = fx["Xtest"]
    d2 = ((Xs[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    Ks = sf2 * np.exp(-d2)
    m = o.params["mean_module.constant"].item()
    mu = o.y_min.item() + o.y_std.item() * (m + Ks @ alpha)
    np.testing.assert_allclose(mean.numpy(), mu, rtol=1e-8)
    V = sla.solve_triangular(np.linalg.cholesky(Ky), Ks.T, lower=True)
    tau = math.exp(o.params["likelihood.noise_covar.raw_noise"].item()) + o.lb_noise
    sd = np.sqrt(np.maximum(sf2 - (V**2).sum(0) + tau, 1e-10)) * o.y_std.item()
    np.testing.assert_allclose(std.numpy(), sd, rtol=1e-6)


def test_matern_kernels_against_direct_formula():
    rng = np.random.default_rng(5)
    x = torch.tensor(rng.standard_normal((30, 4)))
    ls = torch.tensor([[0.7, 1.3, 0.9, 2.0]], dtype=torch.float64)
    d = np.sqrt((((x.numpy()[:, None, :] - x.numpy()[None, :, :]) / ls.numpy()) ** 2).sum(-1))
    k32 = (1 + math.sqrt(3) * d) * np.exp(-math.sqrt(3) * d)
    k52 = (1 + math.sqrt(5) * d + 5.0 / 3.0 * d**2) * np.exp(-math.sqrt(5) * d)
    np.testing.assert_allclose(G.matern_gpytorch(x, x, ls, 1.5).numpy(), k32, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(G.matern_gpytorch(x, x, ls, 2.5).numpy(), k52, rtol=1e-7, atol=1e-9)
    o = G.OracleGP(x.numpy(), rng.standard_normal(30), quant_correlation_class="Matern52Kernel")
    _, grads = o.loss_and_grad()
    h = 1e-5
    flat = o.params[o.ls_key].reshape(-1)
    for i in range(4):
        old = flat[i].item()
        flat[i] = old + h; fp = o.loss().item()
        flat[i] = old - h; fm = o.loss().item()
        flat[i] = old
        assert abs((fp - fm) / (2 * h) - grads[o.ls_key].reshape(-1)[i].item()) < 1e-7


def test_transforms_against_the_reference_module():
    """softplus / inv_softplus of the oracle AND of the product's ``utils.transforms`` against values produced by the reference's own
    ``utils/transforms.py`` (tests/golden/make_ref_transforms.py imports it from /root/reference): fp64 and fp32 arguments."""
    from gpplus_amd.utils.transforms import inv_softplus, softplus

    fx = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_transforms.npz")))
    x, pos = torch.tensor(fx["x"]), torch.tensor(fx["pos"])
    for sp, isp in ((G.softplus, G.inv_softplus), (softplus, inv_softplus)):
        np.testing.assert_allclose(sp(x).numpy(), fx["softplus_x"], rtol=1e-15, atol=0)
        np.testing.assert_allclose(isp(pos).numpy(), fx["inv_softplus_pos"], rtol=1e-15, atol=1e-300)
        np.testing.assert_allclose(sp(x.float()).numpy(), fx["softplus_x32"], rtol=1e-7, atol=0)
        np.testing.assert_allclose(isp(pos.float()).numpy(), fx["inv_softplus_pos32"], rtol=1e-6, atol=1e-30)


def test_oracle_bo_consumers_against_closed_forms():
    """The f4 restatements (oracle_af, oracle_af_engineering, oracle_bo_pool_scores, oracle_sobol) against scipy's normal
    distribution evaluated on OracleGP.predict's own numbers, and Sobol's sum rule on a near-additive posterior mean."""
    from scipy.stats import norm, qmc

    rng = np.random.default_rng(3)
    xs = rng.uniform(-2.0, 2.0, 40)
    src = np.array([0] * 14 + [1] * 26)
    X = np.stack([(xs - xs.mean()) / xs.std(), src.astype(float)], axis=1)
    y = np.sin(3 * xs) + 0.5 * xs + np.where(src == 1, 0.3 * np.cos(2 * xs), 0.0)
    o = G.OracleGP(X, y, qual_dict={1: 2})
    o.params["likelihood.noise_covar.raw_noise"] = torch.full_like(o.params["likelihood.noise_covar.raw_noise"], -5.0)
    cf = lambda v: {0: 10.0, 1: 1.0}[int(v)]  # noqa: E731
    best = float(y.min())
    for pt, maximize in (([0.3, 1.0], False), ([-1.2, 0.0], False), ([1.5, 1.0], True)):
        pt = np.array(pt)
        mu, sd = o.predict(np.array([[(pt[0] - xs.mean()) / xs.std(), pt[1]]]), return_std=True, include_noise=True)
        mu, sd = float(mu), float(sd)
        u = (mu - best - np.sign(best) * 0.02) / sd
        u = u if maximize else -u
        c = cf(pt[1])
        args = (pt, best, o, [xs.mean()], [xs.std()], cf)
        assert abs(G.oracle_af("EI", *args, maximize=maximize, si=0.02) + sd * (norm.pdf(u) + u * norm.cdf(u)) / c) < 1e-12
        assert abs(G.oracle_af("LF", *args, maximize=maximize, si=0.02) + sd * norm.pdf(u) / c) < 1e-12
        assert abs(G.oracle_af("HF", *args, maximize=maximize, si=0.02) + sd * u / c) < 1e-12
    pool = np.stack([rng.uniform(-1, 1, 30), np.array([0] * 10 + [1] * 20).astype(float), rng.normal(size=30)], axis=1)
    bv = [float(y[src == 0].min()), float(y[src == 1].min())]
    scores, idx = G.oracle_bo_pool_scores(o, pool, bv, cf, 2)
    assert scores.shape == (30,) and idx == int(torch.argmax(scores))
    m0, s0 = o.predict(pool[:10, :2], return_std=True, include_noise=False)
    u0 = -(m0 - bv[0]) / s0
    np.testing.assert_allclose(scores[:10].numpy(), (s0 * u0 / 10.0).numpy(), rtol=1e-12)
    gen = qmc.Sobol(d=4, scramble=False)
    gen.fast_forward(1)
    S, ST = G.oracle_sobol(o, gen.random(1024), [2])
    assert S.shape == (1, 2) and ST.shape == (1, 2) and (ST > 0).all() and (ST[0] >= S[0] - 0.05).all()


def _small_oracle(n=48, seed=7):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, 3))
    y = np.sin(1.5 * X[:, 0]) + 0.3 * X[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
    return G.OracleGP(X, y)


def test_oracle_scipy_driver_and_fp32_round_trip():
    """``oracle_fit_scipy`` (optim/mll_scipy.py:243-307): the L-BFGS-B multistart lowers the un-normalised objective from every
    start, loads the best start, and — ``fp32_theta`` (optim/mll_scipy.py:32-35,97) — evaluates at float32-rounded points: the two
    modes differ by that rounding and by nothing else."""
    o = _small_oracle()
    f0 = o.loss(normalize=False).item()
    torch.manual_seed(4)
    res, best = G.oracle_fit_scipy(o, num_restarts=1, options={"maxiter": 40})
    assert len(res) == 2 and best == min(r.fun for r in res) and best < f0
    assert abs(o.loss(normalize=False).item() - best) <= 1e-10 * abs(best)  # the best start's theta is loaded
    # a start is an L-BFGS-B run from a prior draw: the same seed gives the same starts
    o2 = _small_oracle()
    torch.manual_seed(4)
    res2, best2 = G.oracle_fit_scipy(o2, num_restarts=1, options={"maxiter": 40})
    assert best2 == best and all(np.array_equal(a.x, b.x) for a, b in zip(res, res2))
    # fp32 mode: theta is loaded as float32(theta)
    names = list(o.trainable)
    x = G._pack(o, names) + 1e-9  # not representable in float32
    G._unpack(o, names, x, fp32_theta=True)
    np.testing.assert_array_equal(G._pack(o, names), x.astype(np.float32).astype(np.float64))
    f32 = o.loss(normalize=False).item()
    G._unpack(o, names, x.astype(np.float32).astype(np.float64), fp32_theta=False)
    assert o.loss(normalize=False).item() == f32
    G._unpack(o, names, x, fp32_theta=False)
    np.testing.assert_array_equal(G._pack(o, names), x)


def test_oracle_noise_continuation_follows_the_reference_schedule():
    """``oracle_continuation`` (optim/mll_noise_continuation.py:45-244): first pass over initial / 10^i with the noise FIXED at each
    level (raw_noise = log(v - lb), never trained), the level below the bound ends the pass (every start fails on a NaN covariance,
    :178-180), refinement passes between the neighbours of the best level, the selected level's NLL returned."""
    o = _small_oracle(n=40)
    torch.manual_seed(2)
    nll, hist = G.oracle_continuation(o, num_restarts=0, options={"maxiter": 30}, initial_noise_var=1.0)
    lv = hist["noise_history"]
    assert "likelihood.noise_covar.raw_noise" not in o.trainable and o.fix_noise
    assert len(lv) == len(hist["nll_history"]) >= 2 and nll == min(hist["nll_history"])
    assert all(np.isfinite(v) for v in hist["nll_history"])
    # the levels of a pass are monotone (decreasing powers of ten, or an increasing linspace between two of them) and above the bound
    d = np.diff(lv)
    assert (np.all(d < 0) or np.all(d > 0)) and min(lv) >= o.lb_noise
    # the model is left at a state whose noise is the selected level's
    sel = lv[int(np.argmin(hist["nll_history"]))]
    assert abs(float(G.noise_transform(o.params["likelihood.noise_covar.raw_noise"], o.lb_noise)[0]) - sel) <= 1e-12 * sel
