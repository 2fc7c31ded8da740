"""GPU parity, second batch: the standalone operator classes behind ``GPR(correlation_kernel='<name>')``, the Adam
driver against an oracle-driven trajectory, the reference's literal entry call (default dtype + ``fit()``), the BASELINE
configs at FULL size against committed oracle values, and the C5 size through size-independent properties.

Tolerances: MLL / grad-MLL 1e-5 relative (fp64); predictions 1e-4; the fp32-parameter model against its fp64 twin at the
fp32-appropriate tolerance written in the test.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-5


def _loss_and_grads(m):
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood

    m.train()
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    for p in m.parameters():
        p.grad = None
    loss = -mll(m(*m.train_inputs), m.train_targets)
    loss.backward()
    return loss.item(), {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}


def _assert_grads(grads, ref, rtol=RTOL):
    assert set(grads) == set(ref), (sorted(grads), sorted(ref))
    for k, g in grads.items():
        gref = np.asarray(ref[k]).reshape(g.shape)
        np.testing.assert_allclose(g, gref, rtol=rtol, atol=rtol * max(np.abs(gref).max(), 1e-12), err_msg=k)


# ---------------------------------------------------------------------------------------------------
# (a) standalone kernels.Rough_RBF / kernels.wighted_RBF through GPR(correlation_kernel=<str>)
#     reference: models/gpregression.py:89-102, kernels/Rough_RBF.py:18-40
# ---------------------------------------------------------------------------------------------------
class _GPRWithMean:
    """GPR leaves ``mean_module`` to its subclasses (models/gpregression.py:117-120); the smallest such subclass."""

    @staticmethod
    def make(X, y, kernel, mean):
        from gpplus_amd.gpcore import ConstantMean, NormalPrior, ZeroMean
        from gpplus_amd.models.gpregression import GPR

        class Model(GPR):
            def __init__(self):
                super().__init__(X, y, kernel, [], lb_noise=1e-8)
                self.mean_module = ConstantMean(prior=NormalPrior(0., 1)) if mean == "single_constant" else ZeroMean()
                self.tkwargs = {"dtype": torch.float64, "device": torch.device("cuda")}

        torch.set_default_dtype(torch.float64)
        try:
            return Model().to(device="cuda", dtype=torch.float64)
        finally:
            torch.set_default_dtype(torch.float32)


@pytest.mark.parametrize("n,d,mean", [(300, 5, "single_zero"), (1500, 8, "single_constant"), (260, 1, "single_zero")])
def test_standalone_rough_rbf_through_gpr(gpu_ctx, n, d, mean):
    """``GPR(X, y, 'Rough_RBF', [])``: the class named by the string, exp constraint + mollified-uniform prior on its raw
    lengthscale, ScaleKernel on top.  d = 1 exercises the no-ARD branch of kernels/Rough_RBF.py:33-40 (RBFCovariance:
    exp(-dx^2 / 2 l^2)), d > 1 the sqrt(l)-scaled branch (:27-32: exp(-sum l_d dx_d^2))."""
    from oracle.gp_oracle import OracleGP

    rng = np.random.default_rng(11 + d)
    X = rng.standard_normal((n, d))
    y = np.sin(2 * X[:, 0]) + (0.3 * X[:, -1] ** 2 if d > 1 else 0) + 0.05 * rng.standard_normal(n)
    o = OracleGP(X, y, quant_correlation_class="GPR:Rough_RBF", m_gp=mean, lb_noise=1e-8, ard_num_dims=d)
    o.params[o.ls_key] = torch.as_tensor(np.float32(rng.uniform(-1.2, 0.4, (1, d))), dtype=torch.float64)
    o.params["covar_module.raw_outputscale"] = torch.tensor(float(np.float32(0.3)), dtype=torch.float64)
    o.params["likelihood.noise_covar.raw_noise"] = torch.tensor([float(np.float32(-5.0))], dtype=torch.float64)
    if mean == "single_constant":
        o.params["mean_module.constant"] = torch.tensor([float(np.float32(0.2))], dtype=torch.float64)
    lo, go = o.loss_and_grad()

    m = _GPRWithMean.make(torch.tensor(X), torch.tensor(y), "Rough_RBF", mean)
    from gpplus_amd import kernels
    assert type(m.covar_module.base_kernel) is kernels.Rough_RBF
    assert m.covar_module.base_kernel.ard_num_dims == d
    sd = m.state_dict()
    for k, v in o.params.items():
        sd[k] = v.reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    loss, grads = _loss_and_grads(m)
    assert abs(loss - lo.item()) <= RTOL * abs(lo.item()), (loss, lo.item())
    _assert_grads(grads, {k: g.numpy() for k, g in go.items()})
    # predictions from the same model (models/gpregression.py:122-149)
    Xt = rng.standard_normal((64, d))
    pm, ps = o.predict(Xt, return_std=True, include_noise=True)
    gm, gs = m.predict(torch.tensor(Xt, device="cuda"), return_std=True, include_noise=True)
    np.testing.assert_allclose(gm.detach().cpu().numpy(), pm.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(gs.detach().cpu().numpy(), ps.numpy(), rtol=1e-4, atol=1e-8)  # (GPR.predict runs outside no_grad, as in the reference)
    # the dense operator: kernel(x).evaluate() against the oracle's matrix (ScaleKernel included)
    U = torch.tensor(X[:128], device="cuda")
    Kg = m.covar_module(U).evaluate().cpu()
    with torch.no_grad():
        Ko = o.prior_cov(torch.tensor(X[:128]), torch.tensor(X[:128]))
    np.testing.assert_allclose(Kg.numpy(), Ko.numpy(), rtol=1e-9, atol=1e-12)


def test_standalone_rough_rbf_branches_as_operators(gpu_ctx):
    """``Rough_RBF.forward`` switches formula on what it is called with (kernels/Rough_RBF.py:19-26): a 1-D kernel gives
    exp(-dx^2 / 2 l^2) on plain inputs and exp(-l dx^2) when an input requires grad or ``diag`` is asked."""
    from gpplus_amd import kernels
    from oracle.gp_oracle import rough_rbf_standalone

    torch.set_default_dtype(torch.float64)
    try:
        k1 = kernels.Rough_RBF().to("cuda")
        k4 = kernels.Rough_RBF(ard_num_dims=4).to("cuda")
    finally:
        torch.set_default_dtype(torch.float32)
    k1.lengthscale = torch.tensor(0.37)
    k4.lengthscale = torch.tensor([[0.37, 1.4, 0.05, 2.2]])
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((90, 4)), rng.standard_normal((70, 4))
    for kern, cols, ard in ((k1, slice(0, 1), None), (k4, slice(0, 4), 4)):
        xa, xb = torch.tensor(a[:, cols]), torch.tensor(b[:, cols])
        ls = kern.lengthscale.detach().cpu()
        for need_grad in (False, True):
            xa_g = xa.clone().requires_grad_(need_grad)
            ref = rough_rbf_standalone(xa_g, xb, ls, ard_num_dims=ard).detach()
            got = kern(xa_g.detach().cuda().requires_grad_(need_grad), xb.cuda()).evaluate().cpu()
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-10, atol=1e-14, err_msg=f"ard={ard} grad={need_grad}")
        sq = kern(xa.cuda()).evaluate().cpu()
        np.testing.assert_allclose(sq.numpy(), rough_rbf_standalone(xa, xa, ls, ard_num_dims=ard).numpy(), rtol=1e-10, atol=1e-14)
        assert torch.allclose(kern(xa.cuda(), diag=True).cpu(), torch.ones(90, dtype=torch.float64))


def test_standalone_wighted_rbf_through_gpr(gpu_ctx):
    """``GPR(X, y, 'wighted_RBF', [])``: this build's weighted product kernel exp(-sum_d (fixed_d + l_d) dx_d^2) (the
    reference class is an unfinished stub, kernels/wighted_RBF.py:31-41 — documented deviation) against the same formula
    in the oracle, with and without fixed leading weights."""
    from oracle.gp_oracle import OracleGP

    rng = np.random.default_rng(5)
    n, d = 700, 6
    X = rng.standard_normal((n, d))
    y = np.cos(X[:, 0]) * X[:, 1] + 0.05 * rng.standard_normal(n)
    for fixed in (None, [0.5, 0.5]):
        fw = None if fixed is None else np.concatenate([fixed, np.zeros(d - len(fixed))])
        o = OracleGP(X, y, quant_correlation_class="GPR:wighted_RBF", m_gp="single_zero", lb_noise=1e-8, ard_num_dims=d,
                     fixed_weights=fw)
        o.params[o.ls_key] = torch.as_tensor(np.float32(rng.uniform(-1.5, 0.0, (1, d))), dtype=torch.float64)
        o.params["likelihood.noise_covar.raw_noise"] = torch.tensor([float(np.float32(-4.0))], dtype=torch.float64)
        lo, go = o.loss_and_grad()
        m = _GPRWithMean.make(torch.tensor(X), torch.tensor(y), "wighted_RBF", "single_zero")
        if fixed is not None:
            m.covar_module.base_kernel.fixed_weights = torch.tensor(fixed, dtype=torch.float64, device="cuda")
        sd = m.state_dict()
        for k, v in o.params.items():
            sd[k] = v.reshape(sd[k].shape).to(sd[k])
        m.load_state_dict(sd)
        loss, grads = _loss_and_grads(m)
        assert abs(loss - lo.item()) <= RTOL * abs(lo.item())
        _assert_grads(grads, {k: g.numpy() for k, g in go.items()})


# ---------------------------------------------------------------------------------------------------
# (b) fit_model_torch against an oracle-driven Adam trajectory (optim/mll_torch.py:99-137)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,fixture,xkey,kw", [
    ("c1", "c1_borehole_n500.npz", "Xtrain", {}),
    ("mixed", "c3_borehole_mixed_n100.npz", "Utrain", {"qual_dict": {0: 5, 5: 5}}),
])
def test_fit_model_torch_follows_the_oracle_trajectory(gpu_ctx, name, fixture, xkey, kw):
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.optim import fit_model_torch

    fx = dict(np.load(os.path.join(GOLD, fixture)))
    tr = dict(np.load(os.path.join(GOLD, "adam_trajectory.npz")))
    ref_hist = tr[f"{name}::loss_hist"]
    m = GP_Plus(torch.tensor(fx[xkey]), torch.tensor(fx["ytrain"]), dtype=torch.float64, device="cuda", **kw)
    sd = m.state_dict()
    for k in list(sd):
        if f"theta1::param::{k}" in fx:
            sd[k] = torch.as_tensor(fx[f"theta1::param::{k}"]).reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    f_inc, hist = fit_model_torch(m, num_iter=len(ref_hist), num_restarts=0, lr_default=0.01, break_steps=50, verbose=False)
    assert len(hist) == 1 and len(hist[0]) == len(ref_hist)
    np.testing.assert_allclose(np.asarray(hist[0]), ref_hist, rtol=RTOL, atol=0)
    assert abs(f_inc - ref_hist[-1]) <= RTOL * abs(ref_hist[-1])
    # the parameters after the last optimizer.step() (the driver restores the state of the best = only run)
    final = {n: p.detach().cpu().numpy() for n, p in m.named_parameters() if p.requires_grad}
    assert len(final) == sum(1 for k in tr if k.startswith(f"{name}::final::"))
    for k, v in final.items():
        ref = tr[f"{name}::final::{k}"].reshape(v.shape)
        np.testing.assert_allclose(v, ref, rtol=1e-5, atol=1e-7, err_msg=k)


# ---------------------------------------------------------------------------------------------------
# (c) the reference's literal entry call: GP_Plus(Xtrain, ytrain, device='cuda') [default dtype], model.fit()
#     (Examples/01 cell 5; models/gp_plus.py:83,547-567)
# ---------------------------------------------------------------------------------------------------
def test_default_dtype_model_fit_and_predict(gpu_ctx, capsys):
    from gpplus_amd.models import GP_Plus

    fx = dict(np.load(os.path.join(GOLD, "c1_borehole_n500.npz")))
    Xtrain, ytrain = torch.tensor(fx["Xtrain"]).float(), torch.tensor(fx["ytrain"]).float()
    Xtest, ytest = torch.tensor(fx["Xtest"]).float(), torch.tensor(fx["ytest"]).float()
    torch.manual_seed(0)
    model = GP_Plus(Xtrain, ytrain, device="cuda")  # dtype defaults to torch.float (models/gp_plus.py:83)
    assert all(p.dtype == torch.float32 and p.is_cuda for p in model.parameters())
    loss0, _ = _loss_and_grads(model)
    pre = model.predict(Xtest, return_std=False)
    rrmse0 = float(torch.sqrt(((pre.cpu() - ytest) ** 2).mean() / ytest.var()))
    torch.manual_seed(1)
    with pytest.warns(UserWarning, match="adam_torch"):
        f_inc, hist = model.fit()  # default optim_type='scipy' on a GPU model: warning, Adam with 4 restarts (:563-567)
    # fit() advanced the 5 runs together (settings.batched_restarts); the reference's sequential loop from the same seed
    # visits the same start points and must produce the same histories (fp32 parameters: compared at 1e-4)
    from gpplus_amd import settings as gpp_settings
    torch.manual_seed(0)
    seq = GP_Plus(Xtrain, ytrain, device="cuda")
    torch.manual_seed(1)
    with gpp_settings.batched_restarts(False), pytest.warns(UserWarning, match="adam_torch"):
        f_seq, hist_seq = seq.fit()
    assert [len(h) for h in hist_seq] == [len(h) for h in hist]
    for a, b in zip(hist, hist_seq):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)
    assert abs(f_inc - f_seq) <= 1e-4 * abs(f_seq)
    assert len(hist) == 5 and all(1 <= len(h) <= 100 for h in hist)
    assert f_inc < loss0 and abs(f_inc - min(h[-1] for h in hist)) < 1e-12
    out = capsys.readouterr().out
    assert "Learning the model's parameters has started" in out and "successfully finished" in out
    loss_fit, grads_fit = _loss_and_grads(model)
    assert all(g.dtype == np.float32 for g in grads_fit.values())
    mean32, std32 = model.predict(Xtest, return_std=True, include_noise=True)
    assert mean32.shape == (200,) and std32.shape == (200,)
    # fp64 twin: same (fp32-rounded) data, same fitted parameters.  The fp32 model stores data and raw parameters in fp32
    # and applies its constraints in fp32 (the kernels always run in fp64), so the two differ by fp32 rounding of the
    # kernel weights / noise: loss within 1e-4 relative, predictive mean and std within 1e-4 of the output range.
    twin = GP_Plus(Xtrain.double(), ytrain.double(), dtype=torch.float64, device="cuda")
    twin.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()})
    loss64, grads64 = _loss_and_grads(twin)
    assert abs(loss_fit - loss64) <= 1e-4 * abs(loss64), (loss_fit, loss64)
    for k, g in grads64.items():
        np.testing.assert_allclose(grads_fit[k], g, rtol=2e-3, atol=2e-3 * max(np.abs(g).max(), 1e-6), err_msg=k)
    mean64, std64 = twin.predict(Xtest.double(), return_std=True, include_noise=True)
    span = float(ytrain.max() - ytrain.min())
    np.testing.assert_allclose(mean32.double().cpu().numpy(), mean64.cpu().numpy(), rtol=1e-4, atol=1e-4 * span)
    np.testing.assert_allclose(std32.double().cpu().numpy(), std64.cpu().numpy(), rtol=1e-3, atol=1e-4 * span)
    # and the fit is a fit: the test RRMSE of the Borehole emulator at least halves against the untrained model's
    rrmse = float(torch.sqrt(((mean32.cpu() - ytest) ** 2).mean() / ytest.var()))
    assert rrmse < 0.5 * rrmse0, (rrmse, rrmse0)
    res = model.evaluation(Xtest, ytest, verbose=False)
    assert abs(float(res["RRMSE"]) - rrmse) < 1e-3


# ---------------------------------------------------------------------------------------------------
# (d) BASELINE configs at FULL size against committed oracle values (tests/golden/make_fullsize.py)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_full_size_configs_match_the_oracle(gpu_ctx, cfg):
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config

    path = os.path.join(GOLD, f"fullsize_{cfg.lower()}.npz")
    fx = dict(np.load(path))
    X, y, kw, theta = make_config(cfg)
    chk = np.array([float(X.sum()), float((X ** 2).sum()), float(y.sum()), float((y ** 2).sum())])
    np.testing.assert_allclose(chk, fx["checksum"], rtol=1e-12, err_msg="the config generator drifted from the fixture")
    for k, v in theta.items():
        np.testing.assert_array_equal(v.numpy().reshape(-1), fx["theta::" + k].reshape(-1))
    torch.manual_seed(0)
    m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw)
    apply_theta(m, theta)
    loss, grads = _loss_and_grads(m)
    ref = float(fx["loss"])
    assert abs(loss - ref) <= RTOL * abs(ref), (loss, ref)
    _assert_grads(grads, {k[len("grad::"):]: v for k, v in fx.items() if k.startswith("grad::")})
    # a13 at the BASELINE size (models/gpregression.py:122-149): predictive mean / std with and without noise at 256 seeded
    # test points, on the factor path N >= 3840 takes (look-ahead + cooperative panel [+ bordered inverse], gpp_predict_tn)
    from gpplus_amd.test_functions.baseline_configs import make_test_points
    Xt = make_test_points(cfg, X)
    np.testing.assert_allclose(np.array([float(Xt.sum()), float((Xt ** 2).sum())]), fx["test_checksum"], rtol=1e-12)
    m.eval()
    mean, std = m.predict(Xt, return_std=True, include_noise=True)
    np.testing.assert_allclose(mean.cpu().numpy(), fx["pred_mean"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(std.cpu().numpy(), fx["pred_std"], rtol=1e-4, atol=1e-8)
    _, std0 = m.predict(Xt, return_std=True, include_noise=False)
    np.testing.assert_allclose(std0.cpu().numpy(), fx["pred_std_nonoise"], rtol=1e-4, atol=1e-7)
    np.testing.assert_array_equal(m.predict(Xt, return_std=False).cpu().numpy(), mean.cpu().numpy())
    del m
    import gpplus_amd.linalg as L
    L._workspaces.clear()
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------
# (e) C5 size (N = 60 000, d = 16) on one GPU: size-independent properties
# ---------------------------------------------------------------------------------------------------
def test_c5_size_properties_single_gpu(gpu_ctx):
    """The C5 workload through GP_Plus on ONE MI355X (3 x 28.8 GB matrices): Ky alpha = r on sampled rows, the
    gradient against a central difference of the loss along a random direction, invariance under a permutation of the
    data, and d/dtau = (alpha'alpha - tr Ky^-1)/2 with the trace taken from the factor's inverse."""
    from gpplus_amd import linalg
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config

    if torch.cuda.get_device_properties(0).total_memory < 120 * 2 ** 30:
        pytest.skip("needs ~90 GiB of device memory")
    X, y, kw, theta = make_config("C5")
    N, D = X.shape
    assert (N, D) == (60000, 16)
    m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw)
    apply_theta(m, theta)
    loss, grads = _loss_and_grads(m)
    assert np.isfinite(loss)
    ws = linalg.get_workspace(gpu_ctx, N, 0)
    alpha = ws.alpha.clone()
    # (1) Ky alpha = r on 512 sampled rows (cross-covariance rows rebuilt by gpp_cross_kernel)
    g = torch.Generator().manual_seed(0)
    rows = torch.randperm(N, generator=g)[:512].cuda()
    with torch.no_grad():
        U = m.train_inputs[0]
        lazy = m.covar_module(U[rows], U)
        Krows = lazy.evaluate()
        tau = m.likelihood.noise.reshape(())
        r = m.train_targets - m.mean_module(U)
        lhs = Krows @ alpha + tau * alpha[rows]
        assert float((lhs - r[rows]).norm() / r[rows].norm()) < 1e-7
        # (2) d loss / d raw_noise from the trace identity: dMLL/dtau = (alpha'alpha - tr Ky^-1)/2, tr Ky^-1 = ||L^-1||_F^2
        fro = 0.0
        for r0 in range(0, N, 4000):
            blk = torch.tril(ws.Li[r0:r0 + 4000], diagonal=r0)
            fro += float((blk * blk).sum())
            del blk
        dmll_dtau = 0.5 * (float(alpha @ alpha) - fro)
        raw = m.likelihood.noise_covar.raw_noise.detach()
    # prior part of the same derivative through autograd on the prior alone
    raw_p = raw.clone().requires_grad_(True)
    prior = None
    for name, module, pr, closure, _ in m.named_priors():
        if "noise" in name:
            prior = pr.log_prob(raw_p).sum()
    (gp,) = torch.autograd.grad(prior, raw_p)
    expect = -(dmll_dtau * float(torch.exp(raw)) + float(gp)) / N
    got = float(grads["likelihood.noise_covar.raw_noise"].reshape(-1)[0])
    assert abs(got - expect) <= 1e-6 * abs(expect), (got, expect)
    # (3) directional derivative: central difference of the loss along a random direction in parameter space
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    gen = torch.Generator().manual_seed(1)
    direction = {n: torch.randn(dict(m.named_parameters())[n].shape, generator=gen, dtype=torch.float64) for n in names}
    slope = sum(float((torch.as_tensor(grads[n]) * direction[n]).sum()) for n in names)
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    h = 1e-4
    vals = []
    base = {n: dict(m.named_parameters())[n].detach().clone() for n in names}
    for sgn in (+1, -1):
        with torch.no_grad():
            for n in names:
                dict(m.named_parameters())[n].copy_(base[n] + sgn * h * direction[n].cuda())
            vals.append(float(-mll(m(*m.train_inputs), m.train_targets)))
    with torch.no_grad():
        for n in names:
            dict(m.named_parameters())[n].copy_(base[n])
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(fd - slope) <= 1e-5 * max(abs(slope), 1e-8), (fd, slope)
    # (4) permutation invariance of value and gradients
    perm = torch.randperm(N, generator=g)
    del m
    linalg._workspaces.clear()
    torch.cuda.empty_cache()
    m2 = GP_Plus(X[perm].contiguous(), y[perm].contiguous(), dtype=torch.float64, device="cuda", **kw)
    apply_theta(m2, theta)
    loss2, grads2 = _loss_and_grads(m2)
    assert abs(loss2 - loss) <= 1e-9 * abs(loss)
    _assert_grads(grads2, grads, rtol=1e-6)
    del m2
    linalg._workspaces.clear()
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------
# multi-noise model: fit-like training evaluations, then evaluation() / predict() / train again
# (models/gp_plus.py:889-932; the training covariance must keep the TRAINING points' noise groups whatever the last
#  prediction left in likelihood.fidel_indices)
# ---------------------------------------------------------------------------------------------------
def test_multi_noise_evaluation_after_training_matches_oracle(gpu_ctx):
    from oracle.gp_oracle import OracleGP
    from gpplus_amd.models import GP_Plus

    fx = dict(np.load(os.path.join(GOLD, "c4_wing_mf_n300.npz")))
    kw = {"qual_dict": {10: 3}, "multiple_noise": True, "m_gp": "multiple_constant"}
    X, y = fx["Xtrain"], fx["ytrain"]
    Xt = fx["Xtest"][:47]  # M < N and M not a multiple of anything: a stale test-sized index would be out of bounds
    yt = fx["theta1::pred_mean"][:47] + 0.3 * np.sin(np.arange(47))
    o = OracleGP(X, y, **kw)
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device="cuda", **kw)
    sd = m.state_dict()
    for k in list(o.params):
        o.params[k] = torch.as_tensor(fx[f"theta1::param::{k}"]).reshape(o.params[k].shape).clone()
        sd[k] = o.params[k].reshape(sd[k].shape).to(sd[k])
    m.load_state_dict(sd)
    ref = {k: float(v) for k, v in o.evaluation(Xt, yt).items()}
    loss0, grads0 = _loss_and_grads(m)                      # what fit() does: training-mode evaluations ...
    res = m.evaluation(torch.tensor(Xt), torch.tensor(yt), verbose=False)   # ... then straight into evaluation()
    for k, v in ref.items():
        assert abs(float(res[k]) - v) <= 1e-4 * max(abs(v), 1e-8), (k, float(res[k]), v)
    # predict() with a different batch leaves test-sized indices behind; training and the next evaluation are unaffected
    pm, ps = m.predict(torch.tensor(fx["Xtest"][:13]), return_std=True, include_noise=True)
    om, os_ = o.predict(fx["Xtest"][:13], return_std=True, include_noise=True)
    np.testing.assert_allclose(pm.cpu().numpy(), om.numpy(), rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(ps.cpu().numpy(), os_.numpy(), rtol=1e-4, atol=1e-8)
    m.likelihood.fidel_indices = torch.tensor(fx["Xtest"][:13, -1], device="cuda")  # (what GPR.predict leaves behind)
    m.train()
    m.prediction_strategy = None
    res2 = m.evaluation(torch.tensor(Xt), torch.tensor(yt), verbose=False)
    for k, v in ref.items():
        assert abs(float(res2[k]) - v) <= 1e-4 * max(abs(v), 1e-8), (k, float(res2[k]), v)
    # the C ABI front refuses a group index of the wrong length instead of reading past it
    from gpplus_amd._lib import GppError
    from gpplus_amd.backend import square_buffer
    U = torch.rand(64, 3, dtype=torch.float64, device="cuda")
    w = torch.ones(3, dtype=torch.float64, device="cuda")
    one = torch.ones(1, dtype=torch.float64, device="cuda")
    tau = torch.full((2,), 1e-3, dtype=torch.float64, device="cuda")
    with pytest.raises(GppError, match="noise-group index"):
        gpu_ctx.kernel_build(U, w, one, tau, torch.zeros(10, dtype=torch.int32, device="cuda"), square_buffer(64, "cuda"))
    with pytest.raises(GppError, match="out of range"):
        gpu_ctx.kernel_build(U, w, one, tau, torch.full((64,), 2, dtype=torch.int32, device="cuda"), square_buffer(64, "cuda"))
    with pytest.raises(GppError, match="feature columns"):
        gpu_ctx.kernel_build(torch.rand(8, 65, dtype=torch.float64, device="cuda"), torch.ones(65, dtype=torch.float64, device="cuda"),
                             one, None, None, square_buffer(8, "cuda"))


# ---------------------------------------------------------------------------------------------------
# bench.py contract (one JSON line, the fields the driver reads), on a small problem
# ---------------------------------------------------------------------------------------------------
def test_bench_line_contract(gpu_ctx):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--n", "4096",
                        "--cpu-baseline", "none"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "stages"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["unit"] == "evals/s"
    assert rec["dtype"] == "f64" and rec["data"] == "synthetic" and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert "workload" in rec["config"] and rec["config"]["N"] == 4096
    assert abs(rec["value"] - 1e3 / rec["ms_per_step"]) <= 1e-6 * rec["value"]
    rf = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "entries"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["peak"] == 78.6 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    stages = {e["stage"] for e in rf["entries"]}
    assert {"potrf", "trtri", "lauum"} <= stages
    # the stage events lie inside the step: their sum cannot exceed ms_per_step (5 % allowance for event resolution at this size)
    assert sum(rec["stages"]["ms"].values()) <= 1.05 * rec["ms_per_step"]


@pytest.mark.gpu
def test_bench_two_ranks_through_both_legs_on_one_gpu():
    """``bench.py --gpus 2 --share-gpu`` (test mode: both ranks on cuda:0 over gloo, small sizes): the N > 1 code path of the
    benchmark — self-launch, the replica leg with its barrier / MAX timing, the sharded leg, ONE line carrying both — runs on
    the 1-GPU box before a multi-GPU node ever sees it."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1",
                        "--n", "4096", "--sharded-n", "3000", "--nb", "256", "--sharded-steps", "1", "--sharded-warmup", "1",
                        "--sharded-timeout", "300"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["N"] == 4096
    assert abs(rec["value"] - 2 * 1e3 / rec["ms_per_step"]) <= 1e-6 * rec["value"]  # whole-job rate: both replicas
    sh = rec["sharded"]
    assert "error" not in sh, sh
    assert sh["n_gpus"] == 2 and sh["scaling"] == "strong" and sh["config"]["N"] == 3000 and sh["config"]["backend"] == "gloo"
    assert {"shard_factor", "shard_inverse", "shard_backsolve"} <= set(sh["stages"]["ms"])
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
