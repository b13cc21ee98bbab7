"""ctypes binding of ``libsgrender.so`` (the C ABI declared in ``include/sgrender.h``).

The shared library is the product.  Since round 4 the package's operators reach it through the C++ torch extension
(``csrc/sgr_torch.cpp``); this module is the binding INTEGRATION.md shows a maintainer -- it finds the library, declares the
argument types and turns non-zero return codes into Python exceptions -- and what the ABI tests (``tests/test_abi.py``,
raw C-ABI calls in the GPU tests) and the host-side queries (``sgr_fused_recon_supported`` ...) go through.  There
is deliberately no fallback: if the library is missing, every entry point of
the package raises ``SgrenderUnavailable`` (build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C
inverserenderingofindoorscene_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGR_LIB", os.path.join(_HERE, "libsgrender.so"))

ABI_VERSION = 5


class SgrenderUnavailable(RuntimeError):
    pass


class SgrenderError(RuntimeError):
    pass


_P = c_void_p   # device / host pointers travel as plain addresses
_I = c_int
_F = c_float

# symbol -> argument types; must mirror include/sgrender.h exactly (tests/test_abi.py checks
# that every function declared in the header is exported and listed here).
SIGNATURES = {
    "sgr_abi_version": ([], c_int),
    "sgr_last_error": ([], c_char_p),
    "sgr_dirs_padded": ([_I], c_int),
    "sgr_dirs_floats": ([_I, _I], c_int),
    "sgr_fill_direction_table": ([_P, _I, _I], c_int),
    "sgr_fill_view_vectors": ([_P, _I, _I, _F, _P], c_int),
    "sgr_sg_to_env_fwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], c_int),
    "sgr_render_env_fwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P], c_int),
    "sgr_fused_fwd": ([_P] * 11 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_fused_fwd_tan": ([_P] * 13 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_sg_to_env_bwd": ([_P] * 8 + [_I] * 7 + [_P], c_int),
    "sgr_fused_bwd_sg": ([_P] * 14 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_render_env_bwd_env": ([_P] * 8 + [_I] * 7 + [_F, _P], c_int),
    "sgr_render_bwd_brdf": ([_P] * 14 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_loss_workspace_floats": ([_I], c_int),
    "sgr_render_loss_fwd": ([_P] * 10 + [_I] * 5 + [_P], c_int),
    "sgr_render_loss_fwd_total": ([_P] * 11 + [_F, _P] + [_I] * 5 + [_P], c_int),
    "sgr_light_objective_fwd": ([_P] * 25 + [_F] + [_P] * 4 + [_I] * 10 + [_F, _I, _P], c_int),
    "sgr_render_loss_fwd_total_grads": ([_P] * 11 + [_F, _F, _P, _P, _P] + [_I] * 5 + [_P], c_int),
    "sgr_render_loss_bwd": ([_P] * 8 + [_I] * 3 + [_P], c_int),
    "sgr_loss_finalize": ([_P, _P, _P, _F, _P], c_int),
    "sgr_objective_finalize": ([_P, _P, _F, _F, _F, _P, _P, _P], c_int),
    "sgr_render_loss_bwd_scaled": ([_P, _F] + [_P] * 8 + [_I] * 3 + [_P], c_int),
    "sgr_lsregress_coef": ([_P] * 4 + [_I, ctypes.c_longlong, _P], c_int),
    "sgr_lsregress_diffspec_coef": ([_P] * 5 + [_I, _I, _P], c_int),
    "sgr_sg_shading": ([_P] * 5 + [_I] * 7 + [_P], c_int),
    "sgr_recon_workspace_floats": ([_I, _I, _I], c_int),
    "sgr_recon_loss_fwd": ([_P] * 8 + [_I] * 5 + [_F, _P], c_int),
    "sgr_recon_loss_bwd": ([_P] * 6 + [_I] * 5 + [_F, _P], c_int),
    "sgr_fused_recon_supported": ([_I] * 5, c_int),
    "sgr_heads_prologue_supported": ([_I] * 5, c_int),
    "sgr_fused_recon_workspace_floats": ([_I, _I, _I], c_int),
    "sgr_fused_fwd_recon": ([_P] * 17 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_fused_fwd_recon_tan": ([_P] * 19 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_fused_fwd_recon_seg": ([_P] * 10 + [_I, _I] + [_P] * 9 + [_I] * 8 + [_F, _I, _P], c_int),
    "sgr_light_heads_fwd": ([_P] * 7 + [_I] * 4 + [_P], c_int),
    "sgr_light_heads_bwd": ([_P] * 10 + [_I] * 4 + [_P], c_int),
    "sgr_rescale_inplace": ([_P, _P, _I, _P, _P, _P], c_int),
    "sgr_rescale_inplace_flip": ([_P, _P, _I, _P, _P, _I, _P], c_int),
    "sgr_fused_bwd_recon": ([_P] * 19 + [_I] * 8 + [_F, _I, _F, _F, _P], c_int),
    "sgr_fused_bwd_recon_total": ([_P] * 18 + [_I] * 8 + [_F, _I, _F, _F, _P, _F, _P, _P, _P, _P], c_int),
    "sgr_glue_workspace_floats": ([_I], c_int),
    "sgr_light_albedo_scale": ([_P] * 7 + [ctypes.c_longlong, ctypes.c_longlong, _P], c_int),
    "sgr_light_input_fwd": ([_P] * 9 + [_I] * 5 + [_P], c_int),
}

_lib = None


def load():
    """Load (once) and return the ctypes library handle."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SgrenderUnavailable(
            f"{LIB_PATH} not found: the HIP library has not been built "
            "(run __graft_entry__.build() or `make -C inverserenderingofindoorscene_amd/csrc`). "
            "This package has no CPU / PyTorch fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. missing libamdhip64
        raise SgrenderUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (argtypes, restype) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SgrenderUnavailable(f"{LIB_PATH} does not export {name}; stale build?") from e
        fn.argtypes = argtypes
        fn.restype = restype
    ver = lib.sgr_abi_version()
    if ver != ABI_VERSION:
        raise SgrenderUnavailable(f"{LIB_PATH} has ABI version {ver}, this package needs {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sgr_last_error()
        raise SgrenderError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
