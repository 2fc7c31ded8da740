"""Restart start points: ``torch.manual_seed(s); model.reset_parameters()`` of the product against the
oracle's restatement of models/gpregression.py:168-174, priors/horseshoe.py:68-79 and priors/mollified_uniform.py:84-85 —
the SAME values from the same seed, for a plain, a mixed-input and a 3-noise multi-fidelity model, a fixed-noise model
and the kernels chosen by name.  CPU only: building a model and drawing its start points needs no GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.gp_oracle import OracleGP  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def _pair(kind, **extra):
    from gpplus_amd.models import GP_Plus

    if kind == "plain":
        fx = load("c1_borehole_n500.npz")
        X, y, kw = fx["Xtrain"][:60], fx["ytrain"][:60], {}
    elif kind == "mixed":
        fx = load("c3_borehole_mixed_n100.npz")
        X, y, kw = fx["Utrain"], fx["ytrain"], dict(qual_dict={0: 5, 5: 5})
    else:
        fx = load("c4_wing_mf_n300.npz")
        X, y, kw = fx["Xtrain"], fx["ytrain"], dict(qual_dict={10: 3}, multiple_noise=True, m_gp="multiple_constant")
    kw.update(extra)
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, **kw)
    o = OracleGP(X, y, **kw)
    return m, o


def _assert_same_state(m, o, exact=True):
    sd = m.state_dict()
    for k, v in o.params.items():
        got = sd[k].reshape(v.shape)
        if exact:
            assert torch.equal(got, v), (k, got, v)
        else:
            torch.testing.assert_close(got, v, rtol=1e-14, atol=0)


@pytest.mark.parametrize("kind", ["plain", "mixed", "multifidelity"])
@pytest.mark.parametrize("seed", [0, 7, 1234])
def test_reset_parameters_draws_the_oracles_values(kind, seed):
    m, o = _pair(kind)
    torch.manual_seed(seed)
    m.reset_parameters()
    torch.manual_seed(seed)
    o.reset_parameters()
    _assert_same_state(m, o)
    # a second restart continues the same stream on both sides
    state = torch.get_rng_state()
    m.reset_parameters()
    torch.set_rng_state(state)
    o.reset_parameters()
    _assert_same_state(m, o)


@pytest.mark.parametrize("kclass", ["RBFKernel", "Matern32Kernel", "Matern52Kernel"])
def test_reset_parameters_other_quantitative_kernels(kclass):
    # RBFKernel: MollifiedUniformPrior.rsample = a plain uniform draw on [log 0.1, log 10) (models/gp_plus.py:274-277);
    # the Matern classes keep N(-3, 3) (:287-295)
    m, o = _pair("mixed", quant_correlation_class=kclass)
    torch.manual_seed(3)
    m.reset_parameters()
    torch.manual_seed(3)
    o.reset_parameters()
    _assert_same_state(m, o)
    raw = m.state_dict()["covar_module.base_kernel.kernels.1.raw_lengthscale"]
    if kclass == "RBFKernel":
        assert (raw >= np.log(0.1)).all() and (raw < np.log(10)).all()


def test_reset_parameters_named_kernel_through_gpr():
    from gpplus_amd.models import GPR

    fx = load("c1_borehole_n500.npz")
    X, y = fx["Xtrain"][:40], fx["ytrain"][:40]
    g = GPR(torch.tensor(X), torch.tensor(y), "Rough_RBF", [])
    g.tkwargs = {"dtype": torch.float64, "device": torch.device("cpu")}
    g.double()
    o = OracleGP(X, y, quant_correlation_class="GPR:Rough_RBF", m_gp="single_zero", lb_noise=1e-12)
    torch.manual_seed(11)
    g.reset_parameters()
    torch.manual_seed(11)
    o.reset_parameters()
    _assert_same_state(g, o)


def test_fixed_noise_is_skipped_without_consuming_random_numbers():
    m, o = _pair("plain", fix_noise=True)
    before = m.state_dict()["likelihood.noise_covar.raw_noise"].clone()
    torch.manual_seed(5)
    m.reset_parameters()
    torch.manual_seed(5)
    o.reset_parameters()
    assert torch.equal(m.state_dict()["likelihood.noise_covar.raw_noise"], before)
    _assert_same_state(m, o)
    # the same seed WITH a trainable noise gives different outputscale / lengthscale draws: the noise prior drew first
    m2, _ = _pair("plain")
    torch.manual_seed(5)
    m2.reset_parameters()
    assert not torch.equal(m2.state_dict()["covar_module.raw_outputscale"], m.state_dict()["covar_module.raw_outputscale"])


def test_horseshoe_restart_floor_is_the_default_lb_not_the_models():
    """SURVEY B-7 (priors/horseshoe.py:77-79): ``expand`` drops ``lb``, so restart draws of the noise are clamped at the
    constructor default 1e-6 — although the model was built with lb_noise = 1e-8 — and with several noise levels the floor
    is read from ``lb[0]``."""
    from gpplus_amd.priors import LogHalfHorseshoePrior

    # a scale so small that nearly every draw falls below 1e-6: with the constructor's own lb (1e-12) almost none of them
    # would be clamped; after expand() they all sit exactly on log(1e-6)
    p = LogHalfHorseshoePrior(1e-9, 1e-12)
    assert float(p.lb) == pytest.approx(1e-12)
    floor = torch.log(torch.tensor(1e-6))
    for shape in ((1,), (3,)):
        e = p.expand(shape)
        assert e.lb.shape == shape and torch.all(e.lb == torch.tensor(1e-6))
        torch.manual_seed(0)
        own = torch.stack([p.sample() for _ in range(200)])
        torch.manual_seed(0)
        lo = torch.stack([e.sample() for _ in range(200)])
        assert lo.dtype == torch.get_default_dtype()
        assert torch.all(lo >= floor) and (lo == floor).float().mean() > 0.9
        assert (own < floor).float().mean() > 0.9  # the un-expanded prior does go below: its floor is log(1e-12)
        assert torch.all(own >= torch.log(torch.tensor(1e-12)))
    # through the model: no restart ever starts below log(1e-6) although lb_noise = 1e-8 would allow it
    m, _ = _pair("multifidelity")
    for s in range(20):
        torch.manual_seed(s)
        m.reset_parameters()
        assert torch.all(m.state_dict()["likelihood.noise_covar.raw_noise"] >= np.log(1e-6) - 1e-6)


def test_oracle_draws_are_the_torch_distributions_draws():
    """Pins the oracle's written-out draws to torch.distributions (what [3P] gpytorch's priors inherit their ``sample``
    from, and what priors/horseshoe.py:69-70 calls)."""
    from torch.distributions import HalfCauchy, HalfNormal, LogNormal, Normal, Uniform

    for shape in ((1,), (3,)):
        torch.manual_seed(21)
        got = OracleGP._horseshoe_draw(0.01, shape)
        torch.manual_seed(21)
        scale = torch.tensor(0.01).expand(shape)
        local = HalfCauchy(1).rsample(scale.shape)
        ps = HalfNormal(local * scale).rsample(torch.Size([]))
        ps[ps < 1e-6] = 1e-6
        assert torch.equal(got, ps.log())
    m, o = _pair("mixed")
    torch.manual_seed(2)
    o.reset_parameters()
    torch.manual_seed(2)
    lat = Normal(torch.tensor(0.0), torch.tensor(1.0)).expand((2, 10)).sample()
    OracleGP._horseshoe_draw(0.01, (1,))
    os_ = LogNormal(torch.tensor(1e-6), torch.tensor(1.0)).expand(()).sample()
    ls = Normal(torch.tensor(-3.0), torch.tensor(3.0)).expand((1, 6)).sample()
    mean = Normal(torch.tensor(0.0), torch.tensor(1.0)).expand((1,)).sample()
    assert torch.equal(o.params["latent[0, 5]"], lat.double())
    assert torch.equal(o.params["covar_module.base_kernel.kernels.1.raw_lengthscale"], ls.double())
    assert torch.equal(o.params["mean_module.constant"], mean.double())
    v = os_.double()
    torch.testing.assert_close(o.params["covar_module.raw_outputscale"], v + torch.log(-torch.expm1(-v)), rtol=0, atol=0)
    # RBFKernel: the uniform draw sits behind the latent map's, the noise's and the outputscale's
    _, o2 = _pair("mixed", quant_correlation_class="RBFKernel")
    torch.manual_seed(9)
    Normal(torch.tensor(0.0), torch.tensor(1.0)).expand((2, 10)).sample()
    OracleGP._horseshoe_draw(0.01, (1,))
    LogNormal(torch.tensor(1e-6), torch.tensor(1.0)).expand(()).sample()
    lo, hi = torch.tensor(np.log(0.1), dtype=torch.float32), torch.tensor(np.log(10), dtype=torch.float32)
    u = Uniform(lo, hi).expand((1, 6)).rsample()
    torch.manual_seed(9)
    o2.reset_parameters()
    assert torch.equal(o2.params["covar_module.base_kernel.kernels.1.raw_lengthscale"], u.double())
