"""Host-side mirror of the reference's render-path interface, backed by libsgrender.so.

Drop-in for these members of the reference's ``models.py`` (same names, constructor
kwargs, argument meaning and return values):

  ``output2env(SGNum, envWidth=16, envHeight=8, isCuda=True)``        models.py:348-404
      ``.output2env(axisOrig, lambOrig, weightOrig) -> (envmaps, axis, lamb, weight)``
      ``.fromSGtoIm(axis, lamb, weight) -> envmaps``
  ``renderingLayer(imWidth=160, imHeight=120, fov=57, F0=0.05, cameraPos=[0,0,0],
                   envWidth=16, envHeight=8, isCuda=True)``           models.py:407-522
      ``.forwardEnv(diffusePred, normalPred, roughPred, envmap) -> (colorDiffuse, colorSpec)``

plus the additive fused entry ``render_from_sg`` and ``nn.Module`` aliases
``renderLayer`` / ``output_radiance`` (the names BASELINE.json uses).

Everything here is plumbing: shape checks, output allocation with torch, the current HIP
stream, and ``torch.autograd.Function`` wrappers whose forward/backward are single calls
into the C ABI.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, tables

__all__ = ["light_albedo_scale", "light_encoder_input", "light_heads", "unpack_envmaps", "output2env", "renderingLayer", "render_from_sg", "renderLayer", "output_radiance", "predToShading"]


# --------------------------------------------------------------------------- #
# small helpers                                                                #
# --------------------------------------------------------------------------- #
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(dev: torch.device):
    return torch.cuda.current_stream(dev).cuda_stream


def _require_hip(*ts: torch.Tensor) -> torch.device:
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
        self._cache: Dict[Tuple, torch.Tensor] = {}

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


def _dirs(dev, eh: int, ew: int) -> torch.Tensor:
    return _TABLES.get(("dirs", eh, ew), dev, lambda: tables.packed_direction_table(eh, ew))


def _view(dev, R: int, C: int, fov: float, cam: Tuple[float, float, float]) -> torch.Tensor:
    return _TABLES.get(("view", R, C, float(fov), tuple(float(c) for c in cam)), dev,
                       lambda: tables.view_vectors(C, R, fov, cam))


class _SpanWorkspaces:
    """Workspace of the row-span launches (include/sgrender.h: sgr_fused_fwd_ws / sgr_fused_bwd_sg_ws): one zero-filled
    buffer per (device, stream), created on first use and reused -- every call leaves it as it found it.  Inside a
    HIP-graph capture nothing may be cached (the memory belongs to the graph's pool), so a fresh buffer is made per call
    and only its flag words are cleared."""

    FLAG_BYTES = 16384

    def __init__(self):
        self._cache: Dict[Tuple, torch.Tensor] = {}

    def get(self, dev: torch.device) -> Tuple[Optional[torch.Tensor], int]:
        nbytes = int(_lib.load().sgr_span_workspace_bytes())
        if nbytes <= self.FLAG_BYTES:
            return None, 0
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            ws[:self.FLAG_BYTES].zero_()
            return ws, nbytes
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        ws = self._cache.get(key)
        if ws is None or ws.numel() < nbytes:
            if len(self._cache) >= 16:
                self._cache.pop(next(iter(self._cache)))
            ws = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
            self._cache[key] = ws
        return ws, nbytes


_SPAN_WS = _SpanWorkspaces()


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


def _prepool(albedo, normal, rough, R: int, C: int):
    """BRDF maps whose size is not 1x or 2x the env grid (testReal.py:320-346 resizes freely) are
    average-pooled to the grid first -- with torch's own adaptive pooling, so that autograd and
    the window arithmetic are exactly those of models.py:465-469; the kernels then run with
    ratio 1."""
    h, w = albedo.shape[2], albedo.shape[3]
    if (h, w) == (R, C) or (h, w) == (2 * R, 2 * C):
        return albedo, normal, rough
    return (F.adaptive_avg_pool2d(albedo, (R, C)), F.adaptive_avg_pool2d(normal, (R, C)),
            F.adaptive_avg_pool2d(rough, (R, C)))


# --------------------------------------------------------------------------- #
# autograd functions: one C-ABI call per direction                             #
# --------------------------------------------------------------------------- #
class _SGToEnv(torch.autograd.Function):
    """sgr_sg_to_env_fwd / sgr_sg_to_env_bwd."""

    @staticmethod
    def forward(ctx, axis, lamb, weight, eh: int, ew: int, premap: bool, want_tan: bool):
        dev = _require_hip(axis, lamb, weight)
        axis_c, lamb_c, weight_c = axis.contiguous(), lamb.contiguous(), weight.contiguous()
        bn, K, R, C = _check_sg(axis_c, lamb_c, weight_c, None)
        env = torch.empty((bn, 3, R, C, eh, ew), device=dev, dtype=torch.float32)
        lam_t = torch.empty_like(lamb_c) if (premap and want_tan) else None
        w_t = torch.empty_like(weight_c) if (premap and want_tan) else None
        d = _dirs(dev, eh, ew)
        with torch.cuda.device(dev):
            _lib.call("sgr_sg_to_env_fwd", _ptr(axis_c), _ptr(lamb_c), _ptr(weight_c), _ptr(d), _ptr(env),
                      _ptr(lam_t), _ptr(w_t), bn, K, R, C, eh, ew, int(premap), _stream(dev))
        ctx.save_for_backward(axis_c, lamb_c, weight_c)
        ctx.cfg = (eh, ew, premap)
        ctx.set_materialize_grads(False)
        if lam_t is None:
            return env
        return env, lam_t, w_t

    @staticmethod
    def backward(ctx, g_env, g_lam_t=None, g_w_t=None):
        axis, lamb, weight = ctx.saved_tensors
        eh, ew, premap = ctx.cfg
        dev = axis.device
        bn, K, _, R, C = axis.shape
        g_axis = g_lamb = g_weight = None
        if g_env is not None:
            g_env = g_env.contiguous()
            g_axis, g_lamb, g_weight = torch.empty_like(axis), torch.empty_like(lamb), torch.empty_like(weight)
            d = _dirs(dev, eh, ew)
            with torch.cuda.device(dev):
                _lib.call("sgr_sg_to_env_bwd", _ptr(g_env), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d),
                          _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight), bn, K, R, C, eh, ew, int(premap), _stream(dev))
        # cotangents of the returned post-tan tensors (nobody in the reference differentiates
        # through them, wrapperBRDFLight.py:177; handled for completeness with elementwise torch)
        if g_lam_t is not None or g_w_t is not None:
            scale = 0.999 * (np.pi / 2)
            if g_lam_t is not None:
                y = torch.tan(np.pi / 2 * (0.999 * lamb))
                extra = g_lam_t * scale * (1 + y * y)
                g_lamb = extra if g_lamb is None else g_lamb + extra
            if g_w_t is not None:
                y = torch.tan(np.pi / 2 * (0.999 * weight))
                extra = g_w_t * scale * (1 + y * y)
                g_weight = extra if g_weight is None else g_weight + extra
        return g_axis, g_lamb, g_weight, None, None, None, None


class _RenderEnv(torch.autograd.Function):
    """sgr_render_env_fwd / sgr_render_env_bwd_env (+ BRDF-map gradients)."""

    @staticmethod
    def forward(ctx, albedo, normal, rough, env, fov, F0, cam):
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
        ctx.save_for_backward(albedo_c, normal_c, rough_c, env_c)
        ctx.cfg = (fov, F0, cam)
        return diffuse, spec

    @staticmethod
    def backward(ctx, g_diffuse, g_spec):
        albedo, normal, rough, env = ctx.saved_tensors
        fov, F0, cam = ctx.cfg
        dev = albedo.device
        bn, _, h, w = albedo.shape
        _, _, R, C, eh, ew = env.shape
        g_diffuse, g_spec = g_diffuse.contiguous(), g_spec.contiguous()
        d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
        g_env = g_alb = g_nrm = g_rgh = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[3]:
                g_env = torch.empty_like(env)
                _lib.call("sgr_render_env_bwd_env", _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough),
                          _ptr(d), _ptr(v), _ptr(g_env), bn, R, C, eh, ew, h, w, float(F0), _stream(dev))
            if any(ctx.needs_input_grad[:3]):
                g_alb, g_nrm, g_rgh = _brdf_grads(g_diffuse, g_spec, albedo, normal, rough, env, None, None, None,
                                                  d, v, R, C, eh, ew, F0, False, ctx.needs_input_grad[:3])
        return g_alb, g_nrm, g_rgh, g_env, None, None, None


class _FusedRender(torch.autograd.Function):
    """sgr_fused_fwd / sgr_fused_bwd_sg (+ BRDF-map gradients)."""

    @staticmethod
    def forward(ctx, albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap, need_env):
        dev = _require_hip(albedo, normal, rough, axis, lamb, weight)
        albedo_c, normal_c, rough_c = albedo.contiguous(), normal.contiguous(), rough.contiguous()
        axis_c, lamb_c, weight_c = axis.contiguous(), lamb.contiguous(), weight.contiguous()
        bn, K, R, C = _check_sg(axis_c, lamb_c, weight_c, None)
        bn2, h, w = _check_brdf(albedo_c, normal_c, rough_c)
        if bn2 != bn:
            raise RuntimeError("sgrender: BRDF maps and SG parameters disagree on the batch size")
        env = torch.empty((bn, 3, R, C, eh, ew), device=dev, dtype=torch.float32) if need_env else None
        diffuse = torch.empty((bn, 3, R, C), device=dev, dtype=torch.float32)
        spec = torch.empty_like(diffuse)
        d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
        with torch.cuda.device(dev):
            ws, ws_bytes = _SPAN_WS.get(dev)
            _lib.call("sgr_fused_fwd_ws", _ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(axis_c), _ptr(lamb_c),
                      _ptr(weight_c), _ptr(d), _ptr(v), _ptr(env), _ptr(diffuse), _ptr(spec),
                      bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _ptr(ws), ws_bytes, _stream(dev))
        # the env image, when it exists, feeds the BRDF-map gradients (the env-given kernel is faster than re-evaluating the SG)
        if need_env and any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(albedo_c, normal_c, rough_c, axis_c, lamb_c, weight_c, env)
        else:
            ctx.save_for_backward(albedo_c, normal_c, rough_c, axis_c, lamb_c, weight_c)
        ctx.cfg = (eh, ew, fov, F0, cam, premap)
        ctx.set_materialize_grads(False)
        if need_env:
            return env, diffuse, spec
        return diffuse, spec

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        albedo, normal, rough, axis, lamb, weight = saved[:6]
        env_saved = saved[6] if len(saved) > 6 else None
        eh, ew, fov, F0, cam, premap = ctx.cfg
        if len(grads) == 3:
            g_env, g_diffuse, g_spec = grads
        else:
            g_env = None
            g_diffuse, g_spec = grads
        dev = albedo.device
        bn, _, h, w = albedo.shape
        _, K, _, R, C = axis.shape
        d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
        zeros = None
        if g_diffuse is None or g_spec is None:
            zeros = torch.zeros((bn, 3, R, C), device=dev, dtype=torch.float32)
        g_diffuse = zeros if g_diffuse is None else g_diffuse.contiguous()
        g_spec = zeros if g_spec is None else g_spec.contiguous()
        g_env = None if g_env is None else g_env.contiguous()
        g_axis = g_lamb = g_weight = g_alb = g_nrm = g_rgh = None
        with torch.cuda.device(dev):
            if any(ctx.needs_input_grad[3:6]):
                g_axis, g_lamb, g_weight = torch.empty_like(axis), torch.empty_like(lamb), torch.empty_like(weight)
                ws, ws_bytes = _SPAN_WS.get(dev)
                _lib.call("sgr_fused_bwd_sg_ws", _ptr(g_env), _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal),
                          _ptr(rough), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(v),
                          _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight),
                          bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _ptr(ws), ws_bytes, _stream(dev))
            if any(ctx.needs_input_grad[:3]):
                g_alb, g_nrm, g_rgh = _brdf_grads(g_diffuse, g_spec, albedo, normal, rough, env_saved, axis, lamb, weight,
                                                  d, v, R, C, eh, ew, F0, premap, ctx.needs_input_grad[:3])
        return (g_alb, g_nrm, g_rgh, g_axis, g_lamb, g_weight) + (None,) * 7


def _brdf_grads(g_diffuse, g_spec, albedo, normal, rough, env, axis, lamb, weight, d, v, R, C, eh, ew, F0, premap, needs):
    """d/d{albedo, normal, rough}: sgr_render_bwd_brdf (env given, or re-evaluated from the SG)."""
    dev = albedo.device
    bn, _, h, w = albedo.shape
    g_alb, g_nrm, g_rgh = torch.empty_like(albedo), torch.empty_like(normal), torch.empty_like(rough)
    K = 0 if axis is None else axis.shape[1]
    _lib.call("sgr_render_bwd_brdf", _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough),
              _ptr(env), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(v),
              _ptr(g_alb), _ptr(g_nrm), _ptr(g_rgh), bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _stream(dev))
    return (g_alb if needs[0] else None, g_nrm if needs[1] else None, g_rgh if needs[2] else None)


# --------------------------------------------------------------------------- #
# the reference's classes                                                      #
# --------------------------------------------------------------------------- #
class output2env:
    """SG parameters -> per-pixel hemisphere image.  Mirror of models.py:348-404."""

    def __init__(self, SGNum, envWidth=16, envHeight=8, isCuda=True):
        self.SGNum = SGNum
        self.envWidth = envWidth
        self.envHeight = envHeight
        self.isCuda = isCuda
        ls, _ = tables.direction_table(envHeight, envWidth)
        # same attribute the reference exposes: ls [1,1,3,1,1,envHeight,envWidth] (models.py:362-363)
        self.ls = torch.from_numpy(ls.T.reshape(1, 1, 3, 1, 1, envHeight, envWidth).copy())

    def fromSGtoIm(self, axis, lamb, weight):
        """models.py:371-389 (lamb / weight already post-tan)."""
        bn, K, R, C = _check_sg(axis, lamb, weight, self.SGNum)
        return _SGToEnv.apply(axis, lamb, weight, self.envHeight, self.envWidth, False, False)

    def output2env(self, axisOrig, lambOrig, weightOrig):
        """models.py:391-404: returns ``(envmaps, axis, lamb_tan, weight_tan)``."""
        _check_sg(axisOrig, lambOrig, weightOrig, self.SGNum)
        env, lamb, weight = _SGToEnv.apply(axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth, True, True)
        return env, axisOrig, lamb, weight


class renderingLayer:
    """Microfacet quadrature over the env image.  Mirror of models.py:407-522."""

    def __init__(self, imWidth=160, imHeight=120, fov=57, F0=0.05, cameraPos=[0, 0, 0],
                 envWidth=16, envHeight=8, isCuda=True):
        self.imHeight = imHeight
        self.imWidth = imWidth
        self.envWidth = envWidth
        self.envHeight = envHeight
        self.fov_deg = float(fov)
        self.fov = fov / 180.0 * np.pi
        self.F0 = F0
        self.cameraPos = np.array(cameraPos, dtype=np.float32).reshape([1, 3, 1, 1])
        self._cam = tuple(float(c) for c in np.asarray(cameraPos, dtype=np.float32).reshape(-1))
        self.isCuda = isCuda
        # host copies of the attributes the reference exposes (models.py:432-452)
        self.v = torch.from_numpy(tables.view_vectors(imWidth, imHeight, fov, self._cam))[None]
        self.up = torch.Tensor([0, 1, 0])
        ls, omega = tables.direction_table(envHeight, envWidth)
        self.ls = torch.from_numpy(ls)
        self.envWeight = torch.from_numpy(omega).reshape(1, -1, 1, 1, 1)

    def _check_grid(self, R, C):
        if (R, C) != (self.imHeight, self.imWidth):
            raise RuntimeError(f"sgrender: env grid {R}x{C} does not match the layer's imHeight x imWidth "
                               f"{self.imHeight}x{self.imWidth}")

    def forwardEnv(self, diffusePred, normalPred, roughPred, envmap):
        """models.py:461-522: ``(colorDiffuse, colorSpec)``, each ``[bn,3,envRow,envCol]``."""
        if envmap.dim() != 6:
            raise RuntimeError(f"sgrender: envmap must be 6-D, got {tuple(envmap.shape)}")
        R, C, eh, ew = envmap.shape[2:]
        self._check_grid(R, C)
        if (eh, ew) != (self.envHeight, self.envWidth):
            raise RuntimeError("sgrender: envmap direction grid does not match the layer's envHeight x envWidth")
        a, n, r = _prepool(diffusePred, normalPred, roughPred, R, C)
        return _RenderEnv.apply(a, n, r, envmap, self.fov_deg, self.F0, self._cam)

    def forwardSG(self, diffusePred, normalPred, roughPred, axisOrig, lambOrig, weightOrig, need_env=True, premap=True):
        """Fused ``output2env.output2env`` + ``forwardEnv``: ``(env or None, colorDiffuse, colorSpec)``."""
        bn, K, R, C = _check_sg(axisOrig, lambOrig, weightOrig, None)
        self._check_grid(R, C)
        a, n, r = _prepool(diffusePred, normalPred, roughPred, R, C)
        out = _FusedRender.apply(a, n, r, axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth,
                                 self.fov_deg, self.F0, self._cam, bool(premap), bool(need_env))
        if need_env:
            return out
        return (None,) + tuple(out)


def render_from_sg(albedo, normal, rough, axisOrig, lambOrig, weightOrig, need_env=True, fov=57, F0=0.05,
                   cameraPos=(0, 0, 0), envWidth=16, envHeight=8):
    """Functional form of :meth:`renderingLayer.forwardSG` (env grid taken from the SG tensors)."""
    R, C = axisOrig.shape[3], axisOrig.shape[4]
    key = (C, R, float(fov), float(F0), tuple(float(c) for c in cameraPos), envWidth, envHeight)
    layer = _LAYERS.get(key)
    if layer is None:
        if len(_LAYERS) >= 32:
            _LAYERS.pop(next(iter(_LAYERS)))
        layer = _LAYERS[key] = renderingLayer(imWidth=C, imHeight=R, fov=fov, F0=F0, cameraPos=list(cameraPos),
                                              envWidth=envWidth, envHeight=envHeight)
    return layer.forwardSG(albedo, normal, rough, axisOrig, lambOrig, weightOrig, need_env=need_env)


_LAYERS: Dict[Tuple, "renderingLayer"] = {}


class _LightHeads(torch.autograd.Function):
    """sgr_light_heads_fwd / sgr_light_heads_bwd."""

    @staticmethod
    def forward(ctx, xa, xl, xw, need_packed):
        dev = _require_hip(xa, xl, xw)
        xa_c, xl_c, xw_c = xa.contiguous(), xl.contiguous(), xw.contiguous()
        if xa_c.dim() != 4 or xl_c.dim() != 4 or xw_c.dim() != 4 or xa_c.shape[1] % 3 != 0:
            raise RuntimeError("sgrender: light_heads takes the three decoders' [bn,3K,R,C], [bn,K,R,C], [bn,3K,R,C] outputs")
        bn, K3, R, C = xa_c.shape
        K = K3 // 3
        if tuple(xl_c.shape) != (bn, K, R, C) or tuple(xw_c.shape) != (bn, 3 * K, R, C):
            raise RuntimeError(f"sgrender: light_heads shapes disagree: {tuple(xa_c.shape)}, {tuple(xl_c.shape)}, {tuple(xw_c.shape)}")
        axis = torch.empty((bn, K, 3, R, C), device=dev, dtype=torch.float32)
        lamb = torch.empty((bn, K, R, C), device=dev, dtype=torch.float32)
        weight = torch.empty((bn, 3 * K, R, C), device=dev, dtype=torch.float32)
        packed = torch.empty((bn, 7 * K, R, C), device=dev, dtype=torch.float32) if need_packed else None
        with torch.cuda.device(dev):
            _lib.call("sgr_light_heads_fwd", _ptr(xa_c), _ptr(xl_c), _ptr(xw_c), _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(packed),
                      bn, K, R, C, _stream(dev))
        ctx.save_for_backward(xa_c, xl_c, xw_c)
        ctx.set_materialize_grads(False)
        if need_packed:
            return axis, lamb, weight, packed
        return axis, lamb, weight

    @staticmethod
    def backward(ctx, *grads):
        xa, xl, xw = ctx.saved_tensors
        dev = xa.device
        bn, K3, R, C = xa.shape
        g = [None if t is None else t.contiguous() for t in grads] + [None] * (4 - len(grads))
        if all(t is None for t in g):
            return None, None, None, None
        gxa, gxl, gxw = torch.empty_like(xa), torch.empty_like(xl), torch.empty_like(xw)
        with torch.cuda.device(dev):
            _lib.call("sgr_light_heads_bwd", _ptr(xa), _ptr(xl), _ptr(xw), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]),
                      _ptr(gxa), _ptr(gxl), _ptr(gxw), bn, K3 // 3, R, C, _stream(dev))
        return gxa, gxl, gxw, None


def light_heads(xAxis, xLamb, xWeight, need_packed=False):
    """Output activations of the reference's three light decoders (``models.decoderLight`` modes 0 / 1 / 2,
    models.py:336-346) applied to their last convolution's outputs, in one HIP pass each way:
    ``(axisPred [bn,K,3,R,C], lambPred [bn,K,R,C], weightPred [bn,3K,R,C], envmapsPred or None)``.

    ``envmapsPred [bn,7K,R,C]`` (``need_packed=True``) is the packed cascade hand-off tensor of
    wrapperBRDFLight.py:167-168 (what ``outputBRDFLight.py`` stores as ``imenv_*.h5`` and cascade 1 reads)."""
    out = _LightHeads.apply(xAxis, xLamb, xWeight, bool(need_packed))
    if need_packed:
        return out
    return tuple(out) + (None,)


def unpack_envmaps(envmapsPred, SGNum=12):
    """Views ``(axis [bn,K,3,R,C], lamb [bn,K,R,C], weight [bn,3K,R,C])`` of the packed ``[bn,7K,R,C]`` light
    prediction -- the cascade hand-off layout written by wrapperBRDFLight.py:167-168 / ``light_heads(need_packed=True)``
    and read back as ``envmapsPreBatch`` by cascade 1 (models.py:229, dataLoader.py:277-283)."""
    if envmapsPred.dim() != 4 or envmapsPred.shape[1] != 7 * SGNum:
        raise RuntimeError(f"sgrender: envmapsPred must be [bn,{7 * SGNum},envRow,envCol], got {tuple(envmapsPred.shape)}")
    bn, _, R, C = envmapsPred.shape
    K = SGNum
    return (envmapsPred[:, :3 * K].reshape(bn, K, 3, R, C), envmapsPred[:, 3 * K:4 * K], envmapsPred[:, 4 * K:])


def predToShading(pred, envWidth=32, envHeight=16, SGNum=12):
    """GPU version of ``utils.predToShading`` (utils.py:156-195): cosine-weighted irradiance per env cell from
    the packed ``[.., 7*SGNum, envRow, envCol]`` light prediction (axis 3K, lamb K, weight 3K channels, the
    cascade hand-off layout of wrapperBRDFLight.py:167-168).

    numpy in -> numpy ``[3,envRow,envCol]`` out, like the reference (which takes a batch-1 array);
    a HIP tensor ``[bn,7K,R,C]`` in -> tensor ``[bn,3,R,C]`` out.  Forward only."""
    is_np = isinstance(pred, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)).cuda() if is_np else pred
    dev = _require_hip(t)
    if t.dim() != 4 or t.shape[1] != 7 * SGNum:
        raise RuntimeError(f"sgrender: pred must be [bn,{7 * SGNum},envRow,envCol], got {tuple(t.shape)}")
    bn, _, R, C = t.shape
    K = SGNum
    axis = t[:, 0:3 * K].reshape(bn, K, 3, R, C).contiguous()
    lamb = t[:, 3 * K:4 * K].contiguous()
    weight = t[:, 4 * K:7 * K].contiguous()
    out = torch.empty((bn, 3, R, C), device=dev, dtype=torch.float32)
    d = _dirs(dev, envHeight, envWidth)
    with torch.cuda.device(dev):
        _lib.call("sgr_sg_shading", _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(out), bn, K, R, C,
                  envHeight, envWidth, 1, _stream(dev))
    if is_np:
        return out[0].cpu().numpy() if bn == 1 else out.cpu().numpy()
    return out


def light_albedo_scale(diffuseScaled, diffuse, specScaled, spec, albedoPred):
    """``(cLight, cAlbedo)`` of testReal.py:421-432 as 0-d device tensors (no ``.item()`` round trips): the global light /
    albedo scale derived from the ratio of the LSregressDiffSpec-scaled to the unscaled render images, clipped by the
    brightest albedo.  ``envmapsPredImage * cLight`` (testReal.py:431) then stays an asynchronous device multiply."""
    dev = _require_hip(diffuseScaled, diffuse, specScaled, spec, albedoPred)
    dn, d, sn, s = (t.detach().contiguous() for t in (diffuseScaled, diffuse, specScaled, spec))
    if not (dn.shape == d.shape == sn.shape == s.shape):
        raise RuntimeError("sgrender: light_albedo_scale needs four render images of one shape")
    alb = albedoPred.detach().contiguous()
    out = torch.empty(4, device=dev, dtype=torch.float32)
    ws = torch.empty(_lib.load().sgr_glue_workspace_floats(1), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.call("sgr_light_albedo_scale", _ptr(dn), _ptr(d), _ptr(sn), _ptr(s), _ptr(alb), _ptr(out), _ptr(ws),
                  d.numel(), alb.numel(), _stream(dev))
    return out[0], out[1]


def light_encoder_input(imBatch, albedoPred, normalPred, roughPred, depthPred, size=(480, 640)):
    """The light encoder's input of wrapperBRDFLight.py:138-156 in two HIP launches: per-image mean-normalisation of
    albedo and depth, bilinear resize of the five maps to ``size`` and their concatenation.
    Returns ``(inputBatch [bn,11,H,W], albedoPredNormalised, depthPredNormalised)`` (the wrapper returns the normalised
    maps, :139-147).  Forward only: the reference feeds ``inputBatch.detach()`` to the encoder (:158-161)."""
    dev = _require_hip(imBatch, albedoPred, normalPred, roughPred, depthPred)
    im, alb, nrm, rgh, dep = (t.detach().contiguous() for t in (imBatch, albedoPred, normalPred, roughPred, depthPred))
    bn, _, h, w = im.shape
    if tuple(alb.shape) != (bn, 3, h, w) or tuple(nrm.shape) != (bn, 3, h, w) or tuple(rgh.shape) != (bn, 1, h, w) or \
            tuple(dep.shape) != (bn, 1, h, w) or im.shape[1] != 3:
        raise RuntimeError("sgrender: light_encoder_input takes im/albedo/normal [bn,3,h,w] and rough/depth [bn,1,h,w]")
    H, W = int(size[0]), int(size[1])
    out = torch.empty((bn, 11, H, W), device=dev, dtype=torch.float32)
    alb_n, dep_n = torch.empty_like(alb), torch.empty_like(dep)
    ws = torch.empty(_lib.load().sgr_glue_workspace_floats(bn), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.call("sgr_light_input_fwd", _ptr(im), _ptr(alb), _ptr(nrm), _ptr(rgh), _ptr(dep), _ptr(out), _ptr(alb_n), _ptr(dep_n),
                  _ptr(ws), bn, h, w, H, W, _stream(dev))
    return out, alb_n, dep_n


# --------------------------------------------------------------------------- #
# nn.Module aliases under the names BASELINE.json's north_star uses             #
# --------------------------------------------------------------------------- #
class renderLayer(torch.nn.Module):
    """``nn.Module`` face of :class:`renderingLayer` (no parameters, no buffers)."""

    def __init__(self, imWidth=160, imHeight=120, fov=57, F0=0.05, cameraPos=[0, 0, 0], envWidth=16, envHeight=8,
                 isCuda=True):
        super().__init__()
        self.impl = renderingLayer(imWidth, imHeight, fov, F0, cameraPos, envWidth, envHeight, isCuda)

    def forwardEnv(self, diffusePred, normalPred, roughPred, envmap):
        return self.impl.forwardEnv(diffusePred, normalPred, roughPred, envmap)

    def forward(self, diffusePred, normalPred, roughPred, envmap):
        return self.impl.forwardEnv(diffusePred, normalPred, roughPred, envmap)

    def forwardSG(self, *args, **kwargs):
        return self.impl.forwardSG(*args, **kwargs)


class output_radiance(torch.nn.Module):
    """``nn.Module`` face of :class:`output2env`."""

    def __init__(self, SGNum, envWidth=16, envHeight=8, isCuda=True):
        super().__init__()
        self.impl = output2env(SGNum, envWidth, envHeight, isCuda)

    def output2env(self, axisOrig, lambOrig, weightOrig):
        return self.impl.output2env(axisOrig, lambOrig, weightOrig)

    def fromSGtoIm(self, axis, lamb, weight):
        return self.impl.fromSGtoIm(axis, lamb, weight)

    def forward(self, axisOrig, lambOrig, weightOrig):
        return self.impl.output2env(axisOrig, lambOrig, weightOrig)
