"""The handful of gpytorch.settings the path touches.  ``fast_computations`` is accepted and ignored: this back end is
always the exact Cholesky path (the reference forces it with ``fast_computations(log_prob=False)`` at
models/gpregression.py:127,161 and optim/mll_scipy.py:217; optim/mll_torch.py does not, SURVEY.md hazard B-2)."""
from contextlib import contextmanager


class _Value:
    def __init__(self, default):
        self._v = default

    def value(self, *_):
        return self._v

    @contextmanager
    def __call__(self, v):
        old, self._v = self._v, v
        try:
            yield
        finally:
            self._v = old


cholesky_jitter = _Value(1e-8)      # gpytorch.settings.cholesky_jitter (double)
cholesky_max_tries = _Value(3)      # gpytorch.settings.cholesky_max_tries
min_variance = _Value(1e-10)        # gpytorch.settings.min_variance (double)


# Restart-parallel fits on ONE GPU: ``GP_Plus.fit()`` advances the reference's sequential restarts (optim/mll_torch.py:99-141)
# together, one batched evaluation per Adam iteration (optim/mll_batched.py: same start points, same per-run Adam and early
# stop, same winner), whenever the problem is small enough that one evaluation leaves the MI355X mostly idle.
# ``with settings.batched_restarts(False):`` runs the restarts one after the other, exactly as the reference does.
batched_restarts = _Value(True)


# Graph replay of the L-BFGS objective (optim/mll_scipy.py): for small problems, where one evaluation is a chain of ~100 short
# launches issued by ~2 ms of Python, ``fit_model_scipy`` captures objective + gradient once as a HIP graph and replays it per
# evaluation (gp-plus_amd/graphed.py).  ``with settings.graphed_objective(False):`` evaluates eagerly, as the reference does.
graphed_objective = _Value(True)


# Above N = 3840 (where the whole evaluation cannot be one graph) the model's own host code — parameter transforms, manifold map, mean,
# priors and their backward: 130 (C1) to 172 (C3) element-wise launches per evaluation, gpurun census of round 6 — CAN be replayed as
# HIP graphs around the library's call (gp-plus_amd/graphed.py::GraphedSegment; gpcore/models.py, gpcore/mlls.py): same kernels, the
# same numbers bit for bit (tests/test_gpu_graphed.py).  OFF by default, because it does not pay on this stack.  The first form lost
# 0.2 / 1.5 / 1.9 ms at C3 / C4 / C2: the backward was captured from a stream other than the one the leaves' AccumulateGrad nodes were
# created on, autograd forked the capture onto that stream (PyTorch's warning "The AccumulateGrad node's stream does not match ..."),
# and a replayed graph with two branches occupies a second hardware queue — which perturbs the factorisation's CU-masked streams (the
# effect linalg._forward documents for side streams).  Captured on the warm-up stream, on shadow copies of the parameters, the warning
# and that loss are gone — and what remains is that a hipGraphLaunch of ~60 small nodes costs about what issuing them does: same box,
# alternating, three times each: the mixed-input C3 model (172 launches) 20.44 -> 20.15 ms; C4 58.30 -> 58.29; C2 127.40 -> 127.57; the
# plain 8-input model through bench.py: N = 4096 4.41 -> 5.28 ms, 6144 7.13 -> 10.2 (the launch-per-product range is the most
# sensitive), 8192 11.85 -> 12.14, 10 000 19.79 -> 19.98, 15 000 57.1 -> 57.3, 20 000 127.5 -> 127.8 (tools/attic/dev/segments_ab*.sh).
# ``with settings.graphed_segments(True):`` (or GPP_GRAPHED_SEGMENTS=1) switches it on.
graphed_segments = _Value(__import__("os").environ.get("GPP_GRAPHED_SEGMENTS", "0") not in ("", "0"))


# The reference's scipy driver casts every slice of theta to float32 before loading it into the model (optim/mll_scipy.py:32-35
# ``tkwargs``, :97 ``torch.from_numpy(param).to(**tkwargs)``), whatever the model's dtype: an fp64 model's L-BFGS trajectory is
# evaluated at fp32-rounded points.  Off by default (this build keeps the model's dtype);
# ``with settings.reference_fp32_theta(True):`` reproduces the reference's round trip in ``MLLObjective`` — objective, gradient
# and the final ``load_state_dict`` then see float32(theta) — so that a trajectory can be compared with the reference's.
reference_fp32_theta = _Value(False)


# Sharded single evaluation (gp-plus_amd/sharded.py): ``with settings.sharded_evaluation({"group": None, "nb": 1024}):``
# makes every exact-GP log-likelihood inside the block a cooperative evaluation by all ranks of the process group
# (None = the default group).  Every rank must run the same model code with the same parameters.
sharded_evaluation = _Value(None)


@contextmanager
def fast_computations(covar_root_decomposition=True, log_prob=True, solves=True):
    yield


@contextmanager
def max_cholesky_size(_n):
    yield
