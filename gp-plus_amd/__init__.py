"""gp-plus_amd — MI355X-native exact-GP hot path of GP+ (kernel build -> Cholesky -> MLL + gradients).

Import as ``gpplus_amd`` (see ``gpplus_amd/__init__.py``).  The sub-packages mirror the reference's import
surface (``models``, ``kernels``, ``likelihoods_noise``, ``priors``, ``optim``, ``preprocessing``, ``utils``,
``test_functions``); the arithmetic lives in ``csrc/`` behind the C ABI of ``include/gpp.h``.
"""
__version__ = "0.1.0"
