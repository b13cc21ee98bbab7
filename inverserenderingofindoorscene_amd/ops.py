"""``torch.ops.sgrender.*``: the render path as a C++ torch extension.

The operators -- schemas, HIP-device kernels (one call each into the C ABI of ``libsgrender.so``), fake-tensor (Meta) shape
functions and the autograd nodes -- are registered by ``libsgrender_torch.so`` (``csrc/sgr_torch.cpp``,
``TORCH_LIBRARY(sgrender, ...)``); this module only loads that library and keeps the two host-side knobs.  Rounds 2-3 had this
layer in Python (``torch.library.custom_op`` wrappers plus a second, eager ``autograd.Function`` around ctypes calls, because the
wrappers cost ~0.15 ms per step); the C++ extension is the one implementation of both.

  operator (forward)                      reference call it stands for                              C entry point
  sgrender::sg_to_env                     output2env.output2env / fromSGtoIm, models.py:371-404     sgr_sg_to_env_fwd
  sgrender::render_env                    renderingLayer.forwardEnv, models.py:461-522              sgr_render_env_fwd
  sgrender::fused_render                  both back to back, wrapperBRDFLight.py:177+194            sgr_fused_fwd
  sgrender::render_loss                   wrapperBRDFLight.py:170-171,192,197-207                   sgr_render_loss_fwd(_total)
  sgrender::recon_loss_parts              wrapperBRDFLight.py:172-188                               sgr_recon_loss_fwd
  sgrender::light_heads                   models.py:336-346                                         sgr_light_heads_fwd
  sgrender::light_objective               wrapperBRDFLight.py:167-207, trainLight.py:237            sgr_fused_fwd_recon_seg ... sgr_fused_bwd_recon_total
  (backward)  sg_to_env_bwd, render_env_bwd_env, render_bwd_brdf, fused_render_bwd_sg, render_loss_bwd, recon_loss_bwd, light_heads_bwd

There is no CPU implementation: CPU tensors raise, a missing library raises ``SgrenderUnavailable``.
"""
from __future__ import annotations

import os

import torch

from ._lib import LIB_PATH, SgrenderUnavailable

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.environ.get("SGR_TORCH_EXT", os.path.join(_HERE, "libsgrender_torch.so"))

_loaded = False


def load() -> None:
    """Load (once) the C++ extension that registers ``torch.ops.sgrender.*``."""
    global _loaded
    if _loaded:
        return
    if not os.path.isfile(EXT_PATH):
        raise SgrenderUnavailable(
            f"{EXT_PATH} not found: the torch extension has not been built (run __graft_entry__.build() or "
            "`python inverserenderingofindoorscene_amd/csrc/build_torch_ext.py`). This package has no CPU / PyTorch fallback.")
    if not os.path.isfile(LIB_PATH):
        raise SgrenderUnavailable(
            f"{LIB_PATH} not found: the HIP library has not been built (run __graft_entry__.build() or "
            "`make -C inverserenderingofindoorscene_amd/csrc`). This package has no CPU / PyTorch fallback.")
    try:
        torch.ops.load_library(EXT_PATH)
    except OSError as e:
        raise SgrenderUnavailable(f"cannot load {EXT_PATH}: {e}") from e
    _loaded = True


load()
ops = torch.ops.sgrender


def tan_handoff() -> bool:
    """Whether the fused forward passes hand the post-tan sharpness / intensity to their backward (premap mode 2) instead
    of the backward re-evaluating the pre-map.  ``SGR_TAN_HANDOFF=0|1`` overrides the default (read per call: a tuning knob)."""
    v = os.environ.get("SGR_TAN_HANDOFF")
    return _TAN_HANDOFF_DEFAULT if v is None else v not in ("0", "")


_TAN_HANDOFF_DEFAULT = False     # measured at config 2: +45 us of stores in the (write-bound) forward for -12 us in the backward


# --------------------------------------------------------------------------- #
# helpers for callers of the RAW C ABI (tests/test_gpu_heads.py, tools/): the    #
# extension does all of this itself, in C++                                      #
# --------------------------------------------------------------------------- #
def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _dirs(dev, eh: int, ew: int) -> torch.Tensor:
    """Device copy of the packed direction table (include/sgrender.h) from the numpy builder of tables.py."""
    from . import tables
    return torch.from_numpy(tables.packed_direction_table(eh, ew)).to(dev)


def _view(dev, R: int, C: int, fov: float, cam=(0.0, 0.0, 0.0)) -> torch.Tensor:
    from . import tables
    return torch.from_numpy(tables.view_vectors(C, R, fov, tuple(float(c) for c in cam))).to(dev)
