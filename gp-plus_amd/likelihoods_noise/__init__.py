from .multifidelity import Multifidelity_likelihood, Multifidelity_noise  # noqa: F401
