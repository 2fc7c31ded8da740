"""Per-source homoskedastic noise (reference: likelihoods_noise/multifidelity.py:26-136):
    diag_i = sum_k 1[fidel_i == noise_indices[k]] * noise_k
held as (noise vector, int32 group index) so the tile kernel adds it while it writes the diagonal."""
from typing import Optional

import torch

from ..gpcore.kernels import DiagNoise
from ..gpcore.likelihoods import HomoskedasticNoise, _GaussianLikelihoodBase


class Multifidelity_noise(HomoskedasticNoise):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), num_noises=1):
        super().__init__(noise_prior, noise_constraint, batch_shape, num_tasks=num_noises)

    def forward(self, *params, shape: Optional[torch.Size] = None, fidel_indices: torch.Tensor = None,
                noise_indices: list = None, **kwargs) -> DiagNoise:
        if fidel_indices is None or len(fidel_indices) == 0:
            raise ValueError('You need to specify a list of indices for noise such as [1,3]')
        if self.raw_noise.shape[-1] != len(noise_indices):
            raise ValueError('Something is wrong, number of noise and indices are not the same')
        # group id of every point = position of its fidelity level in noise_indices; levels outside the list get a
        # zero noise in the reference (no indicator matches) -> extra group with zero variance
        fid = fidel_indices.reshape(-1)
        # the grouping depends on the data only: computed once per index tensor (the any() below reads a device value, i.e.
        # it synchronises the host with everything enqueued so far — once per fit instead of once per evaluation)
        key = (id(fidel_indices), fidel_indices._version, tuple(noise_indices))
        cached = getattr(self, "_grp_cache", None)
        if cached is None or cached[0] != key or cached[1] is not fidel_indices:
            grp = torch.full(fid.shape, len(noise_indices), dtype=torch.int32, device=fid.device)
            for k, lvl in enumerate(noise_indices):
                grp = torch.where(fid == lvl, torch.full_like(grp, k), grp)
            cached = (key, fidel_indices, grp, bool((grp == len(noise_indices)).any()))
            self._grp_cache = cached
        grp, extra = cached[2], cached[3]
        noise = self.noise.reshape(-1)
        if extra:
            noise = torch.cat([noise, torch.zeros(1, dtype=noise.dtype, device=noise.device)])
        return DiagNoise(noise, grp.to(noise.device), fid.shape[0])


class Multifidelity_likelihood(_GaussianLikelihoodBase):
    def __init__(self, fidel_indices: torch.Tensor, noise_indices: list = [1], noise_prior=None, noise_constraint=None,
                 learn_additional_noise=False, batch_shape=torch.Size(), **kwargs) -> None:
        super().__init__(Multifidelity_noise(noise_prior=noise_prior, noise_constraint=noise_constraint,
                                             batch_shape=batch_shape, num_noises=len(noise_indices)))
        self.fidel_indices = fidel_indices
        self.noise_indices = noise_indices

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.initialize(noise=value)

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    @raw_noise.setter
    def raw_noise(self, value):
        self.noise_covar.initialize(raw_noise=value)

    def _noise_for(self, mvn):
        return self.noise_covar(fidel_indices=self.fidel_indices, noise_indices=self.noise_indices)
