"""Import alias for the package whose sources live in ``gp-plus_amd/`` (a hyphen cannot appear in a Python
module name).  ``import gpplus_amd`` executes ``gp-plus_amd/__init__.py`` and resolves every submodule
(``gpplus_amd.models``, ``gpplus_amd.kernels`` ...) from that directory."""
from pathlib import Path as _Path

_real = _Path(__file__).resolve().parent.parent / "gp-plus_amd"
__path__ = [str(_real)]
__file__ = str(_real / "__init__.py")
exec(compile((_real / "__init__.py").read_text(), __file__, "exec"))
