"""The replayed L-BFGS objective (gp-plus_amd/graphed.py) against the eager evaluation it was captured from:
same value and gradient at every point, the eager path for the points the replay cannot serve (an indefinite Ky, NaN), the same
fit from ``fit_model_scipy`` with the replay on and off — and a prediction cache that notices its factors were overwritten."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.gpu


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def _model(kind, n=None, **extra):
    from gpplus_amd.models import GP_Plus

    if kind == "plain":
        fx = load("c1_borehole_n500.npz")
        X, y, kw = fx["Xtrain"], fx["ytrain"], {}
    elif kind == "mixed":
        fx = load("c3_borehole_mixed_n100.npz")
        X, y, kw = fx["Utrain"], fx["ytrain"], dict(qual_dict={0: 5, 5: 5})
    else:
        fx = load("c4_wing_mf_n300.npz")
        X, y, kw = fx["Xtrain"], fx["ytrain"], dict(qual_dict={10: 3}, multiple_noise=True, m_gp="multiple_constant")
    if n is not None:
        X, y = X[:n], y[:n]
    kw.update(extra)
    return GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device=torch.device("cuda:0"), **kw)


def _objective(model):
    from gpplus_amd.optim.mll_scipy import MLLObjective

    model.train()
    return MLLObjective(model, True, [0, 0])


@pytest.mark.parametrize("kind", ["plain", "mixed", "multifidelity"])
def test_replay_equals_eager_evaluation(kind):
    from gpplus_amd import settings

    m = _model(kind)
    obj = _objective(m)
    x0 = obj.pack_parameters()
    rng = np.random.default_rng(0)
    points = [x0] + [x0 + 0.3 * rng.standard_normal(x0.shape) for _ in range(6)]
    with settings.graphed_objective(False):
        eager = [_objective(m).fun(x) for x in points]
    got = [obj.fun(x) for x in points]
    assert obj._graph is not None and obj._graph.replays == len(points) and obj._graph.declined == 0
    for (fe, ge), (fg, gg) in zip(eager, got):
        # the same launches on the same data in the same order
        assert fg == pytest.approx(fe, rel=1e-13, abs=0)
        np.testing.assert_allclose(gg, ge, rtol=1e-11, atol=1e-13)
    # the replay leaves the evaluated point in the model, as load_state_dict does on the eager path
    np.testing.assert_array_equal(obj.pack_parameters(), points[-1])
    # visiting the points again in another order gives the same numbers again (nothing carried over between replays)
    again = [obj.fun(x) for x in reversed(points)]
    for (f1, g1), (f2, g2) in zip(reversed(got), again):
        assert f1 == f2 and np.array_equal(g1, g2)
    # ... and eager evaluations of the same model, device-wide synchronisations and other GPU work in between change nothing:
    # every replay is served by the graph (a replayed memset node that turned to garbage after a device synchronisation once
    # made every later replay decline — the library now launches no memsets inside an evaluation)
    with settings.graphed_objective(False):
        other = _objective(m)
    for k in range(12):
        other.fun(points[k % len(points)])
        torch.cuda.synchronize()
        torch.randn(512, 512, device="cuda").sum().item()
        f, g = obj.fun(points[k % len(points)])
        assert f == got[k % len(points)][0] and np.array_equal(g, got[k % len(points)][1])
    assert obj._graph.declined == 0


def test_points_the_replay_cannot_serve_take_the_eager_path():
    from gpplus_amd import settings
    from gpplus_amd.errors import NanError

    # 300 points on a line, a very long lengthscale and (almost) no noise: Ky is numerically singular, the factorisation
    # without jitter fails and the answer has to come from the eager path's jitter retries
    from gpplus_amd.models import GP_Plus

    x = torch.linspace(0, 1, 300, dtype=torch.float64).reshape(-1, 1)
    y = torch.sin(6 * x[:, 0])
    m = GP_Plus(x, y, dtype=torch.float64, device=torch.device("cuda:0"), lb_noise=1e-14)
    obj = _objective(m)
    names = list(obj.param_shapes)
    x0 = obj.pack_parameters()
    good = obj.fun(x0)
    assert obj._graph is not None and obj._graph.replays == 1
    bad = x0.copy()
    for i, n in enumerate(names):
        if "raw_lengthscale" in n:
            bad[i] = -9.0   # omega = 10^-9: a constant kernel
        if "raw_noise" in n:
            bad[i] = -40.0
    assert obj._graph.evaluate(bad) is None
    with warnings.catch_warnings(record=True) as w1:
        warnings.simplefilter("always")
        f_g, g_g = obj.fun(bad)
    with settings.graphed_objective(False), warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        f_e, g_e = _objective(m).fun(bad)
    assert any("jitter" in str(w.message) for w in w1) and any("jitter" in str(w.message) for w in w2)
    assert f_g == f_e and np.array_equal(g_g, g_e)
    # the graph is still good for the next point
    f2, g2 = obj.fun(x0)
    assert f2 == good[0] and np.array_equal(g2, good[1])
    # NaN in theta: NanError from the eager path, as without the replay
    nan = x0.copy()
    nan[0] = np.nan
    with pytest.raises(NanError):
        obj.fun(nan)


def test_fit_model_scipy_is_the_same_fit_with_and_without_replay():
    from gpplus_amd import settings
    from gpplus_amd.optim.mll_scipy import fit_model_scipy

    out = {}
    for on in (True, False):
        m = _model("mixed")
        torch.manual_seed(4)
        with settings.graphed_objective(on):
            res, nll = fit_model_scipy(m, num_restarts=2, options={"maxfun": 300})
        out[on] = (nll, [r.nfev for r in res], m.state_dict())
    assert out[True][0] == pytest.approx(out[False][0], rel=1e-10)
    assert out[True][1] == out[False][1]
    for k, v in out[False][2].items():
        torch.testing.assert_close(out[True][2][k], v, rtol=1e-8, atol=1e-10)


def test_prediction_cache_sees_the_replays():
    m = _model("plain", n=200)
    fx = load("c1_borehole_n500.npz")
    Xt = torch.tensor(fx["Xtrain"][200:260], device="cuda:0")
    m.eval()
    with torch.no_grad():
        before = m(Xt).mean.clone()
    cache = m.prediction_strategy
    assert cache is not None and not cache.stale()
    other = _model("plain", n=200)  # same N: same evaluation workspace
    obj = _objective(other)
    x0 = obj.pack_parameters()
    obj.fun(x0)
    obj.fun(x0 + 0.1)
    assert obj._graph.replays == 2
    if cache._ws is obj._graph.ws:
        assert cache.stale()
    with torch.no_grad():
        after = m(Xt).mean
    torch.testing.assert_close(after, before, rtol=1e-12, atol=1e-12)


def test_large_problems_and_the_switch_leave_the_objective_eager():
    from gpplus_amd import settings

    m = _model("plain", n=100)
    with settings.graphed_objective(False):
        obj = _objective(m)
        obj.fun(obj.pack_parameters())
        assert obj._graphed() is None
    from gpplus_amd.graphed import GraphedObjective

    with pytest.raises(RuntimeError, match="N <"):
        GraphedObjective(lambda: None, [], 5000, torch.device("cuda:0"))


@pytest.mark.parametrize("kind", ["plain", "mixed", "multifidelity"])
def test_fit_model_torch_replayed_graph_equals_the_eager_loop(kind):
    """The sequential Adam driver as the reference's notebooks and BO loop call it (optim/mll_torch.py:104-137): with the
    evaluation replayed as one HIP graph (Adam outside it) the loss histories, the winner and the final parameters are bit for bit
    those of the eager loop — restarts included; the graph serves every iteration."""
    from gpplus_amd import settings
    from gpplus_amd.optim.mll_torch import fit_model_torch

    out = {}
    for on in (True, False):
        torch.manual_seed(5)  # (the latent map of the categorical inputs is drawn from the global generator at construction)
        m = _model(kind, n=260)
        torch.manual_seed(11)
        with settings.graphed_objective(on):
            f, hist = fit_model_torch(m, num_iter=14, num_restarts=2, verbose=False)
        out[on] = (f, hist, {k: v.clone() for k, v in m.state_dict().items()})
        g = fit_model_torch.last_graph
        if on:
            assert g is not None and g.replays == 3 * 14 and g.declined == 0, g
        else:
            assert g is None
    assert out[True][0] == out[False][0]
    assert out[True][1] == out[False][1]
    for k, v in out[False][2].items():
        assert torch.equal(out[True][2][k], v), k


def test_fit_model_torch_hands_an_indefinite_step_to_the_eager_path():
    """A start point whose covariance needs jitter: the replay reports the status, the iteration runs eagerly (jitter warning as
    without the graph) and the fit goes on."""
    from gpplus_amd.optim.mll_torch import fit_model_torch

    m = _model("plain", n=200)
    with torch.no_grad():
        m.likelihood.noise_covar.raw_noise.fill_(-40.0)  # tau = lb = 1e-8 on duplicated rows' scale: indefinite in fp64 rounding
        m.covar_module.base_kernel.raw_lengthscale.fill_(-6.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f, hist = fit_model_torch(m, num_iter=12, num_restarts=0, verbose=False)
    g = fit_model_torch.last_graph
    assert g is not None and g.replays == 12
    assert np.isfinite(f) and len(hist[0]) == 12
    if g.declined:
        assert any("jitter" in str(x.message) for x in w)


def test_two_captured_graphs_on_one_handle_alternate():
    """Two evaluations captured on the SAME library handle (each holds one cooperative panel launch, N <= 2048), replayed in turn
    and interleaved with eager evaluations of a third model: every captured panel launch owns a flag block of the capture ring
    (csrc/gpp_api.hip::launch_panel) that neither the other graph nor an eager launch uses, so the numbers are those of the eager
    evaluations, bit for bit, in any order."""
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.graphed import GraphedLossAndGrad

    def eager(m):
        m.train()
        mll = ExactMarginalLogLikelihood(m.likelihood, m)
        for p in m.parameters():
            p.grad = None
        loss = -mll(m(*m.train_inputs), m.train_targets)
        loss.backward()
        return loss.item(), [p.grad.clone() for p in m.parameters() if p.grad is not None]

    ma, mb, mc = _model("plain", n=300), _model("plain", n=480), _model("plain", n=350)
    ref = {k: eager(m) for k, m in (("a", ma), ("b", mb), ("c", mc))}
    graphs = {}
    for k, m in (("a", ma), ("b", mb)):
        mll = ExactMarginalLogLikelihood(m.likelihood, m)
        plist = [p for p in m.parameters() if p.requires_grad]
        graphs[k] = GraphedLossAndGrad(lambda m=m, mll=mll: -mll(m(*m.train_inputs), m.train_targets), plist, int(m.train_targets.shape[0]),
                                       torch.device("cuda:0"))
    for k in ("a", "b", "c", "b", "a", "a", "c", "b"):
        if k == "c":
            got = eager(mc)
            assert got[0] == ref["c"][0]
            continue
        m = ma if k == "a" else mb
        v = graphs[k].step()
        assert v is not None and v == ref[k][0], (k, v, ref[k][0])
        grads = [p.grad for p in m.parameters() if p.grad is not None]
        for g, r in zip(grads, ref[k][1]):
            assert torch.equal(g, r), k


# ---------------------------------------------------------------------------------------------------
# round 6: the host segments of an evaluation above N = 3840 as replayed graphs (graphed.GraphedSegment)
# ---------------------------------------------------------------------------------------------------
def _big_model(cfg, n):
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config

    X, y, kw, theta = make_config(cfg, n)
    torch.manual_seed(0)
    m = GP_Plus(X, y, dtype=torch.float64, device="cuda", **kw)
    apply_theta(m, theta)
    m.train()
    return m


_MLLS = {}


def _eval(m):
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood

    mll = _MLLS.get(id(m))
    if mll is None:
        mll = _MLLS[id(m)] = ExactMarginalLogLikelihood(m.likelihood, m)  # (not an attribute of m: a module cycle)
    for p in m.parameters():
        p.grad = None
    loss = -mll(m(*m.train_inputs), m.train_targets)
    loss.backward()
    return loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("cfg,n", [("C2", 4096), ("C3", 4200), ("C4", 4500)])
def test_graphed_host_segments_give_the_eager_numbers(cfg, n):
    """The plain API above N = 3840 — ``model(*x)``, ``-mll(...)``, ``backward()`` (optim/mll_torch.py:114-117) — with the model's
    forward and the likelihood / prior terms replayed as graphs (``settings.graphed_segments(True)``; off by default: measured
    slower on this stack) against the same calls issued op by op: bitwise the same loss and gradients (same kernels, same data,
    same order), over parameter updates, and the graphs really replay.  C3: the manifold map (gradients w.r.t. the latent matrix
    through dMLL/dU); C4: per-source noise and means."""
    from gpplus_amd import settings

    m = _big_model(cfg, n)
    l0, g0 = _eval(m)
    assert getattr(m, "_prior_segment", None) is None  # (the default is off)
    with settings.graphed_segments(True):
        l1, g1 = _eval(m)
        seg, tail = m._prior_segment["seg"], _MLLS[id(m)]._tail_segment["seg"]
        assert seg is not None and tail is not None
        assert l1 == l0 and set(g0) == set(g1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
        # an optimizer moves the parameters in place: the replays read the new values
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.05)
        for _ in range(3):
            _eval(m)
            opt.step()
        r0 = seg.replays
        l2, g2 = _eval(m)
        assert seg.replays == r0 + 1 and tail.replays >= 4
        with settings.graphed_segments(False):
            l3, g3 = _eval(m)
        assert l2 == l3 and l2 != l1
        for k in g2:
            assert torch.equal(g2[k], g3[k]), k
        # a parameter that stops being trained is a different segment (the continuation driver freezes the noise)
        m.likelihood.raw_noise.requires_grad_(False)
        l4, g4 = _eval(m)
        assert m._prior_segment["seg"] is not seg and "likelihood.noise_covar.raw_noise" not in g4
        with settings.graphed_segments(False):
            l5, g5 = _eval(m)
        assert l4 == l5 and all(torch.equal(g4[k], g5[k]) for k in g4)
        # prediction (eval mode) and gradient-free evaluations do not go through the segments
        m.eval()
        mean, std = m.predict(m.train_inputs[0][:16], return_std=True)
        assert torch.isfinite(mean).all() and torch.isfinite(std).all()
