"""MI355X-native SG x microfacet render layer (drop-in for the reference's models.renderingLayer /
models.output2env call boundary; the compute lives in libsgrender.so, see include/sgrender.h)."""
from ._lib import SgrenderError, SgrenderUnavailable  # noqa: F401
from . import ops  # noqa: F401  (torch.ops.sgrender.* registration)
from .layers import light_albedo_scale, light_encoder_input, light_heads, output2env, output_radiance, predToShading, render_from_sg, renderingLayer, renderLayer, unpack_envmaps  # noqa: F401
from .graphs import CapturedStep, capture_step  # noqa: F401
from .handoff import loadH5, read_cascade_handoff, write_cascade_handoff, writeH5ToFile  # noqa: F401
from .losses import (LSregress, LSregressDiffSpec, combine_loss_parts, ddp_loss_scale, disable_native_allreduce,  # noqa: F401
                     enable_native_allreduce, light_objective, light_objective_supported, native_allreduce_enabled, recon_loss, render_loss)

__all__ = ["light_albedo_scale", "light_encoder_input", "light_heads", "unpack_envmaps", "output2env", "renderingLayer", "render_from_sg", "renderLayer", "output_radiance", "predToShading",
           "LSregress", "LSregressDiffSpec", "render_loss", "recon_loss", "combine_loss_parts", "ddp_loss_scale", "light_objective",
           "light_objective_supported", "enable_native_allreduce", "disable_native_allreduce", "native_allreduce_enabled", "capture_step", "CapturedStep", "writeH5ToFile", "loadH5", "write_cascade_handoff", "read_cascade_handoff", "SgrenderError", "SgrenderUnavailable"]
