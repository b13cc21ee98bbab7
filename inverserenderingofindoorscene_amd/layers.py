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

Everything here is plumbing: shape checks and calls of the operators registered in ``ops.py``
(``torch.ops.sgrender.*``: schema + fake-tensor shape function + autograd formula, each a single call into the C ABI
on the current HIP stream).  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, tables

__all__ = ["light_albedo_scale", "light_encoder_input", "light_heads", "unpack_envmaps", "output2env", "renderingLayer", "render_from_sg", "renderLayer", "output_radiance", "predToShading"]


# host-side helpers and the registered operators live in ops.py (torch.ops.sgrender.*); re-exported here for the loss module
from . import ops  # noqa: E402,F401  (registers the operators)
from .ops import _check_brdf, _check_sg, _dirs, _ptr, _require_hip, _stream, _view  # noqa: E402,F401


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
        _require_hip(axis, lamb, weight)
        _check_sg(axis, lamb, weight, self.SGNum)
        return torch.ops.sgrender.sg_to_env(axis, lamb, weight, self.envHeight, self.envWidth, False, False)[0]

    def output2env(self, axisOrig, lambOrig, weightOrig):
        """models.py:391-404: returns ``(envmaps, axis, lamb_tan, weight_tan)``."""
        _require_hip(axisOrig, lambOrig, weightOrig)
        _check_sg(axisOrig, lambOrig, weightOrig, self.SGNum)
        env, lamb, weight = torch.ops.sgrender.sg_to_env(axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth, True, True)
        return env, axisOrig, lamb, weight


def _dispatcher_needed(*ts) -> bool:
    """The registered operators (torch.ops.sgrender.*) are what FakeTensorMode / torch.compile / functorch see; eager calls on
    real tensors take :class:`_FusedRender` below -- the same two C-ABI calls behind a plain ``autograd.Function``, without the
    ~0.15 ms per step of Python that ``torch.library.custom_op``'s autograd wrapper, schema handling and redispatch add (measured:
    host enqueue 0.28-0.35 ms per layer step against 0.37 ms of GPU work, i.e. a with-loss step was host-bound)."""
    if torch.compiler.is_compiling():
        return True
    return any(type(t) is not torch.Tensor and not isinstance(t, torch.nn.Parameter) for t in ts if t is not None)


class _FusedRender(torch.autograd.Function):
    """sgr_fused_fwd(_tan) / sgr_fused_bwd_sg / sgr_render_bwd_brdf on the current stream; the eager twin of
    ``torch.ops.sgrender.fused_render`` (ops.py) -- same argument checks, same saved tensors, same kernels."""

    @staticmethod
    def forward(ctx, albedo, normal, rough, axis, lamb, weight, eh, ew, fov, F0, cam, premap, need_env, want_tan):
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
        tan = bool(premap and want_tan)
        lam_t, w_t = (torch.empty_like(lamb_c), torch.empty_like(weight_c)) if tan else (None, None)
        d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
        with torch.cuda.device(dev):
            _lib.call("sgr_fused_fwd_tan", _ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(axis_c), _ptr(lamb_c), _ptr(weight_c),
                      _ptr(d), _ptr(v), _ptr(env), _ptr(lam_t), _ptr(w_t), _ptr(diffuse), _ptr(spec),
                      bn, K, R, C, eh, ew, h, w, float(F0), int(premap), _stream(dev))
        brdf_grads = any(ctx.needs_input_grad[:3])
        ctx.save_for_backward(albedo_c, normal_c, rough_c, axis_c, lam_t if tan else lamb_c, w_t if tan else weight_c,
                              env if (need_env and brdf_grads) else None)
        ctx.cfg = (eh, ew, float(fov), float(F0), cam, 2 if tan else int(premap), d, v)
        ctx.set_materialize_grads(False)
        return env, diffuse, spec

    @staticmethod
    def backward(ctx, g_env, g_diffuse, g_spec):
        albedo, normal, rough, axis, lamb, weight, env_saved = ctx.saved_tensors
        eh, ew, fov, F0, cam, premap, d, v = ctx.cfg
        none = (None,) * 14
        if g_env is None and g_diffuse is None and g_spec is None:
            return none
        dev = axis.device
        bn, K, R, C = axis.shape[0], axis.shape[1], axis.shape[3], axis.shape[4]
        h, w = albedo.shape[2], albedo.shape[3]
        if g_diffuse is None or g_spec is None:
            zeros = torch.zeros((bn, 3, R, C), device=dev, dtype=torch.float32)
            g_diffuse = zeros if g_diffuse is None else g_diffuse
            g_spec = zeros if g_spec is None else g_spec
        g_env = None if g_env is None else g_env.contiguous()
        g_diffuse, g_spec = g_diffuse.contiguous(), g_spec.contiguous()
        out = [None] * 14
        with torch.cuda.device(dev):
            if any(ctx.needs_input_grad[3:6]):
                g_axis, g_lamb, g_weight = torch.empty_like(axis), torch.empty_like(lamb), torch.empty_like(weight)
                _lib.call("sgr_fused_bwd_sg", _ptr(g_env), _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough),
                          _ptr(axis), _ptr(lamb), _ptr(weight), _ptr(d), _ptr(v), _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight),
                          bn, K, R, C, eh, ew, h, w, F0, premap, _stream(dev))
                out[3], out[4], out[5] = g_axis, g_lamb, g_weight
            if any(ctx.needs_input_grad[:3]):
                ga, gn, gr = torch.empty_like(albedo), torch.empty_like(normal), torch.empty_like(rough)
                sg = (None, None, None) if env_saved is not None else (axis, lamb, weight)
                _lib.call("sgr_render_bwd_brdf", _ptr(g_diffuse), _ptr(g_spec), _ptr(albedo), _ptr(normal), _ptr(rough), _ptr(env_saved),
                          _ptr(sg[0]), _ptr(sg[1]), _ptr(sg[2]), _ptr(d), _ptr(v), _ptr(ga), _ptr(gn), _ptr(gr),
                          bn, 0 if env_saved is not None else K, R, C, eh, ew, h, w, F0, int(premap == 1), _stream(dev))
                out[0] = ga if ctx.needs_input_grad[0] else None
                out[1] = gn if ctx.needs_input_grad[1] else None
                out[2] = gr if ctx.needs_input_grad[2] else None
        return tuple(out)


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
        _require_hip(diffusePred, normalPred, roughPred, envmap)
        a, n, r = _prepool(diffusePred, normalPred, roughPred, R, C)
        return torch.ops.sgrender.render_env(a, n, r, envmap, self.fov_deg, float(self.F0), list(self._cam))

    def forwardSG(self, diffusePred, normalPred, roughPred, axisOrig, lambOrig, weightOrig, need_env=True, premap=True):
        """Fused ``output2env.output2env`` + ``forwardEnv``: ``(env or None, colorDiffuse, colorSpec)``."""
        bn, K, R, C = _check_sg(axisOrig, lambOrig, weightOrig, None)
        self._check_grid(R, C)
        _require_hip(diffusePred, normalPred, roughPred, axisOrig, lambOrig, weightOrig)
        a, n, r = _prepool(diffusePred, normalPred, roughPred, R, C)
        # the post-tan sharpness / intensity leave the forward kernel only when a backward will read them
        want_tan = bool(premap) and ops.tan_handoff() and torch.is_grad_enabled() and (axisOrig.requires_grad or lambOrig.requires_grad or weightOrig.requires_grad)
        if _dispatcher_needed(a, n, r, axisOrig, lambOrig, weightOrig):
            env, d, s, _, _ = torch.ops.sgrender.fused_render(a, n, r, axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth,
                                                              self.fov_deg, float(self.F0), list(self._cam), bool(premap), bool(need_env), want_tan)
            return (env if need_env else None), d, s
        return _FusedRender.apply(a, n, r, axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth, self.fov_deg, float(self.F0),
                                  self._cam, bool(premap), bool(need_env), want_tan)


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
