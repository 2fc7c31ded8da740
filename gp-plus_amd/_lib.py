"""ctypes binding of ``libgpp_hip.so`` (C ABI declared in ``include/gpp.h``).

This is the only door between the Python host and the HIP kernels.  There is deliberately no CPU fallback: if the
library is missing, cannot be loaded, or a call is made without a GPU, an exception is raised (the reference's own
equivalent of this layer is gpytorch/ATen, reached from ``optim/mll_torch.py:112-117``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

# PyTorch must be imported before libgpp_hip.so is loaded: torch bundles its own libamdhip64 and both copies cannot
# initialise in one process (loading /opt/rocm's first makes every later HIP call report "no device").
import torch  # noqa: F401  (import order matters, see above)

_PKG_DIR = Path(__file__).resolve().parent
_LIB_PATH = _PKG_DIR / "libgpp_hip.so"
_CSRC = _PKG_DIR / "csrc"

#: every symbol ``include/gpp.h`` declares: name -> (restype, argtypes)
_SIGNATURES = {
    "gpp_version": (c_char_p, []),
    "gpp_create": (c_int, [POINTER(c_void_p), c_int]),
    "gpp_destroy": (c_int, [c_void_p]),
    "gpp_set_stream": (c_int, [c_void_p, c_void_p]),
    "gpp_internal_stream": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "gpp_set_option": (c_int, [c_void_p, c_int, c_int]),
    "gpp_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int64, c_int64, c_int, c_int]),
    "gpp_set_workspace": (c_int, [c_void_p, c_void_p, c_size_t]),
    "gpp_kernel_build": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_double, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int64]),
    "gpp_cross_kernel": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p, c_int64]),
    "gpp_potrf": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gpp_potrf_ws": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "gpp_trtri": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "gpp_lauum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64]),
    "gpp_syrk_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int]),
    "gpp_gemm_lower_cols": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_double,
                                    c_double, c_int64, c_int64, c_int, c_int, c_int64, c_int64, c_int]),
    "gpp_shard_list_begin": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, POINTER(c_int)]),
    "gpp_shard_piece_cols": (c_int64, []),
    "gpp_shard_list_gate": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gpp_shard_list_signal": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gpp_shard_list_end": (c_int, [c_void_p]),
    "gpp_set_comm": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gpp_comm_unique_id": (c_int, [c_void_p]),
    "gpp_comm_init_rccl": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gpp_shard_buffer_doubles": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int]),
    "gpp_shard_eval": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_double, c_int, c_int, c_void_p, POINTER(c_int)]),
    "gpp_push_create": (c_int, [c_int, c_int, c_int, c_int64, POINTER(c_void_p), c_void_p]),
    "gpp_push_connect": (c_int, [c_void_p, c_void_p]),
    "gpp_push_send": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64),
                              POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "gpp_push_recv": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64),
                              POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "gpp_push_ack": (c_int, [c_void_p, c_void_p, c_int64]),
    "gpp_push_info": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int)]),
    "gpp_push_destroy": (c_int, [c_void_p, c_int]),
    "gpp_shard_back_list": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_int, POINTER(c_int)]),
    "gpp_trmv_lower_cols": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int]),
    "gpp_mll_scalars": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpp_grad_reduce_cols": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int]),
    "gpp_lauum_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int]),
    "gpp_lauum_rows_range": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_int64, c_int64]),
    "gpp_transpose": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64]),
    "gpp_mll_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpp_alpha": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpp_grad_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpp_grad_reduce_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "gpp_predict": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                            c_void_p, c_int64, c_void_p, c_void_p]),
    "gpp_predict_tn": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                               c_void_p, c_int64, c_void_p, c_void_p]),
    "gpp_gemm": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_void_p,
                         c_int64, c_double, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int]),
    "gpp_gemm_batched": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64,
                                 c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                 c_int, c_int]),
    "gpp_kernel_build_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_double, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_int]),
    "gpp_potrf_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "gpp_trtri_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                  c_int64, c_int]),
    "gpp_lauum_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int]),
    "gpp_mll_reduce_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_int]),
    "gpp_alpha_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int]),
    "gpp_grad_reduce_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int]),
}

_lib = None


class GppError(RuntimeError):
    """A libgpp_hip call returned a non-zero status."""


def build(force: bool = False) -> Path:
    """Compile ``libgpp_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", str(_CSRC), "clean"], check=True, capture_output=True)
    proc = subprocess.run(["make", "-C", str(_CSRC), "-j4"], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libgpp_hip.so failed:\n" + proc.stdout + proc.stderr)
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the shared library and bind every entry point; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise GppError(
            f"{_LIB_PATH} not found: the HIP extension is required (build it with "
            f"`make -C {_CSRC}` or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback."
        )
    lib = ctypes.CDLL(os.fspath(_LIB_PATH))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def exported_symbols() -> list:
    return sorted(_SIGNATURES)


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status < 0:
        raise GppError(f"{what}: bad argument #{-status}")
    raise GppError(f"{what}: HIP runtime error {status - 1000}")
