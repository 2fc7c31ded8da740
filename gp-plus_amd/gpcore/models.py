"""``ExactGP`` (gpytorch.models.ExactGP subset): train-mode call returns the prior at the training inputs, eval-mode
call returns the exact predictive distribution through a cached factorisation (gpytorch's prediction strategy),
computed by the HIP back end (linalg.factorize / predict_from_cache)."""
from __future__ import annotations

import warnings

import torch

from .distributions import DenseCovariance, MultivariateNormal
from .kernels import LazyKernelMatrix
from .module import Module


class GP(Module):
    pass


class ExactGP(GP):
    def __init__(self, train_inputs, train_targets, likelihood):
        if train_inputs is not None and torch.is_tensor(train_inputs):
            train_inputs = (train_inputs,)
        super().__init__()
        self.train_inputs = None if train_inputs is None else tuple(
            (i.unsqueeze(-1) if i.ndimension() == 1 else i) for i in train_inputs)
        self.train_targets = train_targets
        self.likelihood = likelihood
        self.prediction_strategy = None

    def _apply(self, fn, *args, **kwargs):
        if self.train_inputs is not None:
            self.train_inputs = tuple(fn(t) for t in self.train_inputs)
            self.train_targets = fn(self.train_targets)
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        if mode:
            self.prediction_strategy = None
        return super().train(mode)

    def _ensure_prediction_cache(self, **kwargs):
        """Factor the training covariance once per eval() phase (gpytorch's prediction strategy caches)."""
        from ..linalg import factorize

        if self.prediction_strategy is not None and self.prediction_strategy.stale():
            self.prediction_strategy = None  # another model reused the prediction workspace: factor again
        if self.prediction_strategy is None:
            with torch.no_grad():
                train_out = Module.__call__(self, *self.train_inputs, **kwargs)
                cov = train_out.lazy_covariance_matrix
                if not isinstance(cov, LazyKernelMatrix):
                    raise RuntimeError("exact prediction needs the model's forward to return a lazy kernel covariance")
                # The training covariance takes its noise groups from the TRAINING inputs' source column, whatever a
                # previous predict()/evaluation() left in likelihood.fidel_indices (models/gpregression.py:136-139
                # overwrites it with the test points' sources): set it around the call and put the caller's value back.
                lik = self.likelihood
                swap = hasattr(lik, "fidel_indices")
                if swap:
                    saved, lik.fidel_indices = lik.fidel_indices, self.train_inputs[0][:, -1]
                try:
                    noisy = lik(train_out).lazy_covariance_matrix
                finally:
                    if swap:
                        lik.fidel_indices = saved
                self.prediction_strategy = factorize(cov.U1, cov.spec, noisy.tau, noisy.grp, train_out.mean, self.train_targets)
        return self.prediction_strategy

    def _graphed_prior_call(self, inputs, kwargs):
        """The training-mode call — the model's own ``forward``: manifold map, mean, parameter transforms, kernel weights
        (models/gp_plus.py:386-484) — as a replayed pair of HIP graphs (gp-plus_amd/graphed.py::GraphedSegment) where that applies
        (N >= 3840, autograd on, a lazy kernel covariance); None otherwise: the caller runs ``forward`` op by op.  Same kernels on
        the same data: the same numbers."""
        from ..graphed import GraphedSegment, segment_key, segments_apply
        from ..linalg import KernelSpec

        if kwargs or len(inputs) != 1 or not torch.is_tensor(inputs[0]) or not segments_apply(inputs[0].shape[0], inputs[0].device):
            return None
        params = [p for p in self.parameters()]
        if not any(p.requires_grad for p in params) or any(p.device != inputs[0].device for p in params):
            return None
        key = segment_key(params, inputs[0])
        st = getattr(self, "_prior_segment", None)
        if st is None or st["key"] != key:
            meta = {}

            def fn():
                out = Module.__call__(self, *inputs)
                cov = out.lazy_covariance_matrix
                if not isinstance(cov, LazyKernelMatrix) or not cov.is_square or cov.tau is not None:
                    raise TypeError("not a lazy kernel covariance")
                meta.update(kind=cov.spec.kind, d_split=cov.spec.d_split, n_grad_dims=cov.n_grad_dims,
                            U=None if cov.U1.requires_grad else cov.U1)
                tens = [out.mean, cov.spec.w, cov.spec.sf2.reshape(1)]
                if cov.U1.requires_grad:
                    tens.append(cov.U1)
                return tuple(tens)

            st = {"key": key, "seg": None, "meta": meta}
            try:
                st["seg"] = GraphedSegment(fn, params, inputs[0].device, module=self)
            except (TypeError, RuntimeError) as exc:  # a forward the stack cannot capture: op by op, and say so once
                warnings.warn(f"the model's forward could not be captured as a HIP graph ({exc}); evaluating it op by op", RuntimeWarning)
            self._prior_segment = st
        if st["seg"] is None:
            return None
        outs, meta = st["seg"](), st["meta"]
        U = outs[3] if meta["U"] is None else meta["U"]
        cov = LazyKernelMatrix(U, None, KernelSpec(outs[1], outs[2].reshape(()), meta["kind"], meta["d_split"]), n_grad_dims=meta["n_grad_dims"])
        return MultivariateNormal(outs[0], cov)

    def __call__(self, *args, **kwargs):
        inputs = [a.unsqueeze(-1) if torch.is_tensor(a) and a.ndimension() == 1 else a for a in args]
        if self.training:
            if self.train_inputs is None:
                raise RuntimeError("train_inputs, train_targets cannot be None in training mode.")
            if not all(ti is x or torch.equal(ti, x) for ti, x in zip(self.train_inputs, inputs)):  # (equal() waits for the GPU)
                raise RuntimeError("You must train on the training inputs!")
            out = self._graphed_prior_call(inputs, kwargs)
            return out if out is not None else Module.__call__(self, *inputs, **kwargs)
        # ---- posterior mode -----------------------------------------------------------------------
        from ..linalg import factorize, predict_from_cache, cross_kernel

        with torch.no_grad():
            self._ensure_prediction_cache(**kwargs)
            cache = self.prediction_strategy
            test_out = Module.__call__(self, *inputs, **kwargs)
            tcov = test_out.lazy_covariance_matrix
            Us = tcov.U1.to(torch.float64).contiguous()
            # the mean is O(M N) and computed now; the variance needs V = K_*N L^-T (M N^2 flops) and is computed when — if —
            # somebody asks for it (predict(return_std=False), the acquisition means of a BO loop and Sobol's p + 2 batches do not)
            mean_c, _, _ = predict_from_cache(cache, Us, need_var=False)
            pred_mean = test_out.mean.to(torch.float64) + mean_c
            lazy = {}

            def var_and_v(cache=cache, Us=Us):
                if "V" not in lazy:
                    # another model of the same size may have factored into the shared workspace since this prediction was
                    # made (p1 = m1(x); p2 = m2(x); p1.stddev): the cache rebuilds its factor from its own inputs
                    cache.refresh()
                    with torch.no_grad():
                        _, lazy["var"], lazy["V"] = predict_from_cache(cache, Us, need_var=True, need_V=True)
                return lazy["var"], lazy["V"]

            def full_cov(Us=Us, spec=cache.spec, gctx=cache.gctx):
                M = Us.shape[0]
                V = var_and_v()[1]
                Kss = LazyKernelMatrix(Us, None, spec).evaluate()
                # Kss - V V^T through the MFMA GEMM (NT, lower + mirrored by symmetry)
                gctx.gemm(0, 1, M, M, V.shape[1], -1.0, V, V, 1.0, Kss)
                return Kss

            return MultivariateNormal(pred_mean, DenseCovariance(lambda: var_and_v()[0], full_cov, n=Us.shape[0]))
