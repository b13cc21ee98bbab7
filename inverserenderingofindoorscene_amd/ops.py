"""``torch.ops.sgrender.*``: the render path registered with the PyTorch dispatcher.

Each operator is one C-ABI call into ``libsgrender.so`` (include/sgrender.h) on the current HIP stream; this module adds
what makes it a torch extension rather than a ctypes call -- an operator schema, a fake-tensor (meta) shape function and
an autograd formula whose backward is itself made of registered operators -- so the layer is visible to the dispatcher,
works under ``FakeTensorMode`` / ``torch.compile`` (as opaque calls: there is nothing for a compiler to fuse into a
hand-written kernel) and composes with other autograd code.  The C ABI stays the boundary: nothing here computes.

  operator (forward)                      reference call it stands for                       C entry point
  sgrender::sg_to_env                     output2env.output2env / fromSGtoIm, models.py:371-404    sgr_sg_to_env_fwd
  sgrender::render_env                    renderingLayer.forwardEnv, models.py:461-522             sgr_render_env_fwd
  sgrender::fused_render                  both back to back, wrapperBRDFLight.py:177+194           sgr_fused_fwd
  (backward)  sg_to_env_bwd, render_env_bwd_env, render_bwd_brdf, fused_render_bwd_sg              sgr_*_bwd*

There is no CPU implementation: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, tables


# --------------------------------------------------------------------------- #
# host-side helpers shared by the package                                      #
# --------------------------------------------------------------------------- #
def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream(dev: torch.device):
    return torch.cuda.current_stream(dev).cuda_stream


def _require_hip(*ts: Tensor) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not torch.is_tensor(t):
            raise TypeError("sgrender: expected torch tensors")
        if not t.is_cuda:
            raise RuntimeError(
                "sgrender: this layer runs only on HIP device tensors (MI355X); there is no CPU path. "
                "Move the inputs to the GPU (the reference's isCuda=True mode).")
        if t.dtype != torch.float32:
            raise RuntimeError(f"sgrender: fp32 tensors required, got {t.dtype}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"sgrender: tensors on different devices ({dev} vs {t.device})")
    return dev


class _DeviceTables:
    """Constant tables, created lazily on whichever device the inputs live on.

    The reference keeps them as bare attributes on the layer object, created on the current
    device when ``isCuda`` (models.py:454-459) and never moved by ``.to()``; keying by device
    keeps that behaviour while making one object usable from several ranks / devices."""

    MAX_ENTRIES = 64      # testReal.py builds a layer per image size: keep the cache bounded

    def __init__(self):
        self._cache: Dict[Tuple, Tensor] = {}

    def get(self, key: Tuple, dev: torch.device, make):
        k = key + (str(dev),)
        t = self._cache.get(k)
        if t is None:
            if len(self._cache) >= self.MAX_ENTRIES:
                self._cache.pop(next(iter(self._cache)))      # oldest entry
            t = torch.from_numpy(np.ascontiguousarray(make())).to(dev)
            self._cache[k] = t
        return t


_TABLES = _DeviceTables()


def _dirs(dev, eh: int, ew: int) -> Tensor:
    return _TABLES.get(("dirs", eh, ew), dev, lambda: tables.packed_direction_table(eh, ew))


def _view(dev, R: int, C: int, fov: float, cam: Sequence[float]) -> Tensor:
    cam = tuple(float(c) for c in cam)
    return _TABLES.get(("view", R, C, float(fov), cam), dev, lambda: tables.view_vectors(C, R, fov, cam))


def _check_sg(axis, lamb, weight, K: Optional[int]):
    if axis.dim() != 5 or axis.shape[2] != 3:
        raise RuntimeError(f"sgrender: axis must be [bn,SGNum,3,envRow,envCol], got {tuple(axis.shape)}")
    bn, k, _, R, C = axis.shape
    if K is not None and k != K:
        raise RuntimeError(f"sgrender: axis has {k} lobes, layer was built with SGNum={K}")
    if tuple(lamb.shape) != (bn, k, R, C):
        raise RuntimeError(f"sgrender: lamb must be [bn,SGNum,envRow,envCol]={(bn, k, R, C)}, got {tuple(lamb.shape)}")
    if tuple(weight.shape) != (bn, 3 * k, R, C):
        raise RuntimeError(f"sgrender: weight must be [bn,3*SGNum,envRow,envCol]={(bn, 3 * k, R, C)}, got {tuple(weight.shape)}")
    if k > 32:
        raise RuntimeError("sgrender: SGNum > 32 is not supported")
    return bn, k, R, C


def _check_brdf(albedo, normal, rough):
    if albedo.dim() != 4 or albedo.shape[1] != 3:
        raise RuntimeError(f"sgrender: diffusePred must be [bn,3,h,w], got {tuple(albedo.shape)}")
    bn, _, h, w = albedo.shape
    if tuple(normal.shape) != (bn, 3, h, w):
        raise RuntimeError(f"sgrender: normalPred must be {(bn, 3, h, w)}, got {tuple(normal.shape)}")
    if tuple(rough.shape) != (bn, 1, h, w):
        raise RuntimeError(f"sgrender: roughPred must be {(bn, 1, h, w)}, got {tuple(rough.shape)}")
    return bn, h, w


def tan_handoff() -> bool:
    """Whether the fused forward passes hand the post-tan sharpness / intensity to their backward (premap mode 2) instead
    of the backward re-evaluating the pre-map.  ``SGR_TAN_HANDOFF=0|1`` overrides the default (read per call: a tuning knob)."""
    import os
    v = os.environ.get("SGR_TAN_HANDOFF")
    return _TAN_HANDOFF_DEFAULT if v is None else v not in ("0", "")


_TAN_HANDOFF_DEFAULT = False     # measured at config 2: +45 us of stores in the (write-bound) forward for -12 us in the backward


def _none(t: Tensor) -> Optional[Tensor]:
    """Operators return tensors only: an output that was not asked for is an empty tensor."""
    return None if t.numel() == 0 else t


# --------------------------------------------------------------------------- #
# SG -> env image                                                              #
# --------------------------------------------------------------------------- #
@torch.library.custom_op("sgrender::sg_to_env", mutates_args=())
def sg_to_env(axis: Tensor, lamb: Tensor, weight: Tensor, eh: int, ew: int, premap: bool, want_tan: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """``(env [bn,3,R,C,eh,ew], lamb_tan, weight_tan)``; the post-tan tensors are empty unless ``premap and want_tan``."""
    dev = _require_hip(axis, lamb, weight)
    axis_c, lamb_c, weight_c = axis.contiguous(), lamb.contiguous(), weight.contiguous()
    bn, K, R, C = _check_sg(axis_c, lamb_c, weight_c, None)
    env = torch.empty((bn, 3, R, C, eh, ew), device=dev, dtype=torch.float32)
    tan = premap and want_tan
    lam_t = torch.empty_like(lamb_c) if tan else lamb_c.new_empty(0)
    w_t = torch.empty_like(weight_c) if tan else weight_c.new_empty(0)
    d = _dirs(dev, eh, ew)
    with torch.cuda.device(dev):
        _lib.call("sgr_sg_to_env_fwd", _ptr(axis_c), _ptr(lamb_c), _ptr(weight_c), _ptr(d), _ptr(env),
                  _ptr(lam_t) if tan else None, _ptr(w_t) if tan else None, bn, K, R, C, eh, ew, int(premap), _stream(dev))
    return env, lam_t, w_t


@sg_to_env.register_fake
def _(axis, lamb, weight, eh, ew, premap, want_tan):
    bn, K, R, C = _check_sg(axis, lamb, weight, None)
    tan = premap and want_tan
    return (axis.new_empty((bn, 3, R, C, eh, ew)), torch.empty_like(lamb, memory_format=torch.contiguous_format) if tan else lamb.new_empty(0),
            torch.empty_like(weight, memory_format=torch.contiguous_format) if tan else weight.new_empty(0))


@torch.library.custom_op("sgrender::sg_to_env_bwd", mutates_args=())
def sg_to_env_bwd(g_env: Tensor, axis: Tensor, lamb: Tensor, weight: Tensor, eh: int, ew: int, premap: int) -> Tuple[Tensor, Tensor, Tensor]:
    """``premap``: 0 post-tan inputs (gradients w.r.t. them), 1 raw decoder outputs, 2 post-tan inputs saved by the forward with
    the gradients still w.r.t. the raw ones (include/sgrender.h)."""
    dev = _require_hip(g_env, axis, lamb, weight)
    g_env, axis, lamb, weight = g_env.contiguous(), axis.contiguous(), lamb.contiguous(), weight.contiguous()
    bn, K, R, C = _check_sg(axis, lamb, weight, None)
    g_axis, g_lamb, g_weight = torch.empty_like(axis), torch.empty_like(lamb), torch.empty_like(weight)
    d = _dirs(dev, eh, ew)
    with torch.cuda.device(dev):
        _lib.call("sgr_sg_to_env_bwd", _ptr(g_env), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d),
                  _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight), bn, K, R, C, eh, ew, int(premap), _stream(dev))
    return g_axis, g_lamb, g_weight


@sg_to_env_bwd.register_fake
def _(g_env, axis, lamb, weight, eh, ew, premap):
    c = torch.contiguous_format
    return torch.empty_like(axis, memory_format=c), torch.empty_like(lamb, memory_format=c), torch.empty_like(weight, memory_format=c)


def _sg_to_env_setup(ctx, inputs, output):
    axis, lamb, weight, eh, ew, premap, want_tan = inputs
    if premap and want_tan:
        # the post-tan tensors exist anyway (the reference returns them): the backward reads them instead of re-evaluating
        # 4K tangents per cell (premap mode 2)
        ctx.save_for_backward(axis, output[1], output[2])
        ctx.cfg = (eh, ew, 2)
    else:
        ctx.save_for_backward(axis, lamb, weight)
        ctx.cfg = (eh, ew, int(premap))
    ctx.set_materialize_grads(False)


def _sg_to_env_backward(ctx, g_env, g_lam_t, g_w_t):
    axis, lamb, weight = ctx.saved_tensors
    eh, ew, premap = ctx.cfg
    g_axis = g_lamb = g_weight = None
    if g_env is not None:
        g_axis, g_lamb, g_weight = torch.ops.sgrender.sg_to_env_bwd(g_env, axis, lamb, weight, eh, ew, premap)
    # cotangents of the returned post-tan tensors (nobody in the reference differentiates through them,
    # wrapperBRDFLight.py:177; handled for completeness with elementwise torch).  Only produced in mode 2,
    # where `lamb` / `weight` are the saved post-tan tensors themselves.
    scale = 0.999 * (np.pi / 2)
    if g_lam_t is not None and g_lam_t.numel():
        extra = g_lam_t * scale * (1 + lamb * lamb)
        g_lamb = extra if g_lamb is None else g_lamb + extra
    if g_w_t is not None and g_w_t.numel():
        extra = g_w_t * scale * (1 + weight * weight)
        g_weight = extra if g_weight is None else g_weight + extra
    return g_axis, g_lamb, g_weight, None, None, None, None


sg_to_env.register_autograd(_sg_to_env_backward, setup_context=_sg_to_env_setup)


# --------------------------------------------------------------------------- #
# env image -> (diffuse, specular)                                             #
# --------------------------------------------------------------------------- #
@torch.library.custom_op("sgrender::render_env", mutates_args=())
def render_env(albedo: Tensor, normal: Tensor, rough: Tensor, env: Tensor, fov: float, F0: float, cam: List[float]) -> Tuple[Tensor, Tensor]:
    dev = _require_hip(albedo, normal, rough, env)
    albedo_c, normal_c, rough_c, env_c = albedo.contiguous(), normal.contiguous(), rough.contiguous(), env.contiguous()
    bn, h, w = _check_brdf(albedo_c, normal_c, rough_c)
    if env_c.dim() != 6 or env_c.shape[0] != bn or env_c.shape[1] != 3:
        raise RuntimeError(f"sgrender: envmap must be [bn,3,envRow,envCol,envHeight,envWidth], got {tuple(env_c.shape)}")
    _, _, R, C, eh, ew = env_c.shape
    diffuse = torch.empty((bn, 3, R, C), device=dev, dtype=torch.float32)
    spec = torch.empty_like(diffuse)
    d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
    with torch.cuda.device(dev):
        _lib.call("sgr_render_env_fwd", _ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(env_c), _ptr(d), _ptr(v),
                  _ptr(diffuse), _ptr(spec), bn, R, C, eh, ew, h, w, float(F0), _stream(dev))
    return diffuse, spec


@render_env.register_fake
def _(albedo, normal, rough, env, fov, F0, cam):
    bn, h, w = _check_brdf(albedo, normal, rough)
    R, C = env.shape[2], env.shape[3]
    return albedo.new_empty((bn, 3, R, C)), albedo.new_empty((bn, 3, R, C))


@torch.library.custom_op("sgrender::render_env_bwd_env", mutates_args=())
def render_env_bwd_env(g_diffuse: Tensor, g_spec: Tensor, albedo: Tensor, normal: Tensor, rough: Tensor, eh: int, ew: int,
                       fov: float, F0: float, cam: List[float]) -> Tensor:
    """dL/dEnv of forwardEnv (dense, [bn,3,R,C,eh,ew])."""
    dev = _require_hip(g_diffuse, g_spec, albedo, normal, rough)
    g_diffuse, g_spec = g_diffuse.contiguous(), g_spec.contiguous()
    albedo, normal, rough = albedo.contiguous(), normal.contiguous(), rough.contiguous()
    bn, h, w = _check_brdf(albedo, normal, rough)
    R, C = g_diffuse.shape[2], g_diffuse.shape[3]
    g_env = torch.empty((bn, 3, R, C, eh, ew), device=dev, dtype=torch.float32)
    d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
    with torch.cuda.device(dev):
        _lib.call("sgr_render_env_bwd_env", _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough),
                  _ptr(d), _ptr(v), _ptr(g_env), bn, R, C, eh, ew, h, w, float(F0), _stream(dev))
    return g_env


@render_env_bwd_env.register_fake
def _(g_diffuse, g_spec, albedo, normal, rough, eh, ew, fov, F0, cam):
    bn, _, R, C = g_diffuse.shape
    return g_diffuse.new_empty((bn, 3, R, C, eh, ew))


@torch.library.custom_op("sgrender::render_bwd_brdf", mutates_args=())
def render_bwd_brdf(g_diffuse: Tensor, g_spec: Tensor, albedo: Tensor, normal: Tensor, rough: Tensor, env: Optional[Tensor],
                    axis: Optional[Tensor], lamb: Optional[Tensor], weight: Optional[Tensor], eh: int, ew: int, fov: float, F0: float,
                    cam: List[float], premap: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """d/d{albedo, normal, rough} of forwardEnv; the env image is given, or re-evaluated from the SG parameters."""
    dev = _require_hip(g_diffuse, g_spec, albedo, normal, rough, env, axis, lamb, weight)
    g_diffuse, g_spec = g_diffuse.contiguous(), g_spec.contiguous()
    albedo, normal, rough = albedo.contiguous(), normal.contiguous(), rough.contiguous()
    env = None if env is None else env.contiguous()
    if env is None and (axis is None or lamb is None or weight is None):
        raise RuntimeError("sgrender: render_bwd_brdf needs the env image or the SG parameters")
    axis, lamb, weight = (None, None, None) if env is not None else (axis.contiguous(), lamb.contiguous(), weight.contiguous())
    bn, h, w = _check_brdf(albedo, normal, rough)
    R, C = g_diffuse.shape[2], g_diffuse.shape[3]
    g_alb, g_nrm, g_rgh = torch.empty_like(albedo), torch.empty_like(normal), torch.empty_like(rough)
    K = 0 if axis is None else axis.shape[1]
    d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
    with torch.cuda.device(dev):
        _lib.call("sgr_render_bwd_brdf", _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough),
                  _ptr(env), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(v),
                  _ptr(g_alb), _ptr(g_nrm), _ptr(g_rgh), bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _stream(dev))
    return g_alb, g_nrm, g_rgh


@render_bwd_brdf.register_fake
def _(g_diffuse, g_spec, albedo, normal, rough, env, axis, lamb, weight, eh, ew, fov, F0, cam, premap):
    c = torch.contiguous_format
    return torch.empty_like(albedo, memory_format=c), torch.empty_like(normal, memory_format=c), torch.empty_like(rough, memory_format=c)


def _render_env_setup(ctx, inputs, output):
    albedo, normal, rough, env, fov, F0, cam = inputs
    ctx.save_for_backward(albedo, normal, rough, env)
    ctx.cfg = (fov, F0, list(cam))
    ctx.set_materialize_grads(False)


def _render_env_backward(ctx, g_diffuse, g_spec):
    albedo, normal, rough, env = ctx.saved_tensors
    fov, F0, cam = ctx.cfg
    eh, ew = env.shape[4], env.shape[5]
    if g_diffuse is None and g_spec is None:
        return (None,) * 7
    zeros = None
    if g_diffuse is None or g_spec is None:
        zeros = torch.zeros((env.shape[0], 3, env.shape[2], env.shape[3]), device=env.device, dtype=torch.float32)
    g_diffuse = zeros if g_diffuse is None else g_diffuse
    g_spec = zeros if g_spec is None else g_spec
    g_env = g_alb = g_nrm = g_rgh = None
    if ctx.needs_input_grad[3]:
        g_env = torch.ops.sgrender.render_env_bwd_env(g_diffuse, g_spec, albedo, normal, rough, eh, ew, fov, F0, cam)
    if any(ctx.needs_input_grad[:3]):
        ga, gn, gr = torch.ops.sgrender.render_bwd_brdf(g_diffuse, g_spec, albedo, normal, rough, env, None, None, None,
                                                         eh, ew, fov, F0, cam, False)
        g_alb, g_nrm, g_rgh = (ga if ctx.needs_input_grad[0] else None, gn if ctx.needs_input_grad[1] else None,
                               gr if ctx.needs_input_grad[2] else None)
    return g_alb, g_nrm, g_rgh, g_env, None, None, None


render_env.register_autograd(_render_env_backward, setup_context=_render_env_setup)


# --------------------------------------------------------------------------- #
# fused: SG -> (env image), diffuse, specular                                  #
# --------------------------------------------------------------------------- #
@torch.library.custom_op("sgrender::fused_render", mutates_args=())
def fused_render(albedo: Tensor, normal: Tensor, rough: Tensor, axis: Tensor, lamb: Tensor, weight: Tensor, eh: int, ew: int,
                 fov: float, F0: float, cam: List[float], premap: bool, need_env: bool, want_tan: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """``(env, diffuse, spec, lamb_tan, weight_tan)``; ``env`` is empty when ``need_env`` is false (the env image is then never
    materialised); the post-tan tensors are empty unless ``premap and want_tan`` (they are what the backward reads)."""
    dev = _require_hip(albedo, normal, rough, axis, lamb, weight)
    albedo_c, normal_c, rough_c = albedo.contiguous(), normal.contiguous(), rough.contiguous()
    axis_c, lamb_c, weight_c = axis.contiguous(), lamb.contiguous(), weight.contiguous()
    bn, K, R, C = _check_sg(axis_c, lamb_c, weight_c, None)
    bn2, h, w = _check_brdf(albedo_c, normal_c, rough_c)
    if bn2 != bn:
        raise RuntimeError("sgrender: BRDF maps and SG parameters disagree on the batch size")
    env = torch.empty((bn, 3, R, C, eh, ew), device=dev, dtype=torch.float32) if need_env else albedo_c.new_empty(0)
    diffuse = torch.empty((bn, 3, R, C), device=dev, dtype=torch.float32)
    spec = torch.empty_like(diffuse)
    tan = premap and want_tan
    lam_t = torch.empty_like(lamb_c) if tan else lamb_c.new_empty(0)
    w_t = torch.empty_like(weight_c) if tan else weight_c.new_empty(0)
    d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
    with torch.cuda.device(dev):
        _lib.call("sgr_fused_fwd_tan", _ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(axis_c), _ptr(lamb_c),
                  _ptr(weight_c), _ptr(d), _ptr(v), _ptr(env) if need_env else None, _ptr(lam_t) if tan else None,
                  _ptr(w_t) if tan else None, _ptr(diffuse), _ptr(spec),
                  bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _stream(dev))
    return env, diffuse, spec, lam_t, w_t


@fused_render.register_fake
def _(albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap, need_env, want_tan):
    bn, K, R, C = _check_sg(axis, lamb, weight, None)
    _check_brdf(albedo, normal, rough)
    env = albedo.new_empty((bn, 3, R, C, eh, ew)) if need_env else albedo.new_empty(0)
    tan = premap and want_tan
    c = torch.contiguous_format
    return (env, albedo.new_empty((bn, 3, R, C)), albedo.new_empty((bn, 3, R, C)),
            torch.empty_like(lamb, memory_format=c) if tan else lamb.new_empty(0),
            torch.empty_like(weight, memory_format=c) if tan else weight.new_empty(0))


@torch.library.custom_op("sgrender::fused_render_bwd_sg", mutates_args=())
def fused_render_bwd_sg(g_env: Optional[Tensor], g_diffuse: Tensor, g_spec: Tensor, albedo: Tensor, normal: Tensor, rough: Tensor,
                        axis: Tensor, lamb: Tensor, weight: Tensor, eh: int, ew: int, fov: float, F0: float, cam: List[float],
                        premap: int) -> Tuple[Tensor, Tensor, Tensor]:
    """SG gradients of the fused pass; ``g_env`` is the env image's cotangent from its other consumers, if any.
    ``premap`` as in :func:`sg_to_env_bwd` (2: ``lamb`` / ``weight`` are the post-tan tensors the forward returned)."""
    dev = _require_hip(g_env, g_diffuse, g_spec, albedo, normal, rough, axis, lamb, weight)
    g_env = None if g_env is None else g_env.contiguous()
    g_diffuse, g_spec = g_diffuse.contiguous(), g_spec.contiguous()
    albedo, normal, rough = albedo.contiguous(), normal.contiguous(), rough.contiguous()
    axis, lamb, weight = axis.contiguous(), lamb.contiguous(), weight.contiguous()
    bn, K, R, C = _check_sg(axis, lamb, weight, None)
    _, h, w = _check_brdf(albedo, normal, rough)
    g_axis, g_lamb, g_weight = torch.empty_like(axis), torch.empty_like(lamb), torch.empty_like(weight)
    d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
    with torch.cuda.device(dev):
        _lib.call("sgr_fused_bwd_sg", _ptr(g_env), _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal),
                  _ptr(rough), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(v),
                  _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight),
                  bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _stream(dev))
    return g_axis, g_lamb, g_weight


@fused_render_bwd_sg.register_fake
def _(g_env, g_diffuse, g_spec, albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap):
    c = torch.contiguous_format
    return torch.empty_like(axis, memory_format=c), torch.empty_like(lamb, memory_format=c), torch.empty_like(weight, memory_format=c)


def _fused_setup(ctx, inputs, output):
    albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap, need_env, want_tan = inputs
    env = output[0]
    mode = int(premap)
    if premap and want_tan:      # the backward reads the post-tan tensors the forward produced (premap mode 2)
        lamb, weight, mode = output[3], output[4], 2
    # the env image, when it exists, feeds the BRDF-map gradients (the env-given kernel is faster than re-evaluating the SG)
    if need_env and any(ctx.needs_input_grad[:3]):
        ctx.save_for_backward(albedo, normal, rough, axis, lamb, weight, env)
    else:
        ctx.save_for_backward(albedo, normal, rough, axis, lamb, weight)
    ctx.cfg = (eh, ew, fov, F0, list(cam), mode)
    ctx.mark_non_differentiable(output[3], output[4])      # internal hand-off to the backward, not part of the layer's interface
    ctx.set_materialize_grads(False)


def _fused_backward(ctx, g_env, g_diffuse, g_spec, _g_lam_t=None, _g_w_t=None):
    saved = ctx.saved_tensors
    albedo, normal, rough, axis, lamb, weight = saved[:6]
    env_saved = saved[6] if len(saved) > 6 else None
    eh, ew, fov, F0, cam, premap = ctx.cfg
    if g_env is not None and g_env.numel() == 0:
        g_env = None
    if g_env is None and g_diffuse is None and g_spec is None:
        return (None,) * 14
    bn, R, C = axis.shape[0], axis.shape[3], axis.shape[4]
    zeros = None
    if g_diffuse is None or g_spec is None:
        zeros = torch.zeros((bn, 3, R, C), device=axis.device, dtype=torch.float32)
    g_diffuse = zeros if g_diffuse is None else g_diffuse
    g_spec = zeros if g_spec is None else g_spec
    g_axis = g_lamb = g_weight = g_alb = g_nrm = g_rgh = None
    if any(ctx.needs_input_grad[3:6]):
        g_axis, g_lamb, g_weight = torch.ops.sgrender.fused_render_bwd_sg(g_env, g_diffuse, g_spec, albedo, normal, rough,
                                                                          axis, lamb, weight, eh, ew, fov, F0, cam, premap)
    if any(ctx.needs_input_grad[:3]):
        if env_saved is not None:
            ga, gn, gr = torch.ops.sgrender.render_bwd_brdf(g_diffuse, g_spec, albedo, normal, rough, env_saved, None, None, None,
                                                             eh, ew, fov, F0, cam, premap == 1)
        else:
            ga, gn, gr = torch.ops.sgrender.render_bwd_brdf(g_diffuse, g_spec, albedo, normal, rough, None, axis, lamb, weight,
                                                             eh, ew, fov, F0, cam, premap == 1)
        g_alb, g_nrm, g_rgh = (ga if ctx.needs_input_grad[0] else None, gn if ctx.needs_input_grad[1] else None,
                               gr if ctx.needs_input_grad[2] else None)
    return (g_alb, g_nrm, g_rgh, g_axis, g_lamb, g_weight) + (None,) * 8


fused_render.register_autograd(_fused_backward, setup_context=_fused_setup)
