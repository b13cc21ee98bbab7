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

Everything here is plumbing: grid checks and calls of the operators the C++ torch extension registers
(``torch.ops.sgrender.*``, csrc/sgr_torch.cpp: schema + HIP-device kernel + fake-tensor shape function + autograd node, each a
single call into the C ABI on the current HIP stream).  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import tables
from . import ops as _ops      # loads libsgrender_torch.so (registers torch.ops.sgrender.*)

__all__ = ["light_albedo_scale", "light_encoder_input", "light_heads", "unpack_envmaps", "output2env", "renderingLayer", "render_from_sg", "renderLayer", "output_radiance", "predToShading"]

_sg = torch.ops.sgrender


def _sg_dims(axis):
    if axis.dim() != 5 or axis.shape[2] != 3:
        raise RuntimeError(f"sgrender: axis must be [bn,SGNum,3,envRow,envCol], got {tuple(axis.shape)}")
    return axis.shape[0], axis.shape[1], axis.shape[3], axis.shape[4]


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

    def _check_lobes(self, axis):
        _, k, _, _ = _sg_dims(axis)
        if k != self.SGNum:
            raise RuntimeError(f"sgrender: axis has {k} lobes, layer was built with SGNum={self.SGNum}")

    def fromSGtoIm(self, axis, lamb, weight):
        """models.py:371-389 (lamb / weight already post-tan)."""
        self._check_lobes(axis)
        return _sg.sg_to_env(axis, lamb, weight, self.envHeight, self.envWidth, False, False)[0]

    def output2env(self, axisOrig, lambOrig, weightOrig):
        """models.py:391-404: returns ``(envmaps, axis, lamb_tan, weight_tan)``."""
        self._check_lobes(axisOrig)
        env, lamb, weight = _sg.sg_to_env(axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth, True, True)
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
        return _sg.render_env(a, n, r, envmap, self.fov_deg, float(self.F0), self._cam)

    def forwardSG(self, diffusePred, normalPred, roughPred, axisOrig, lambOrig, weightOrig, need_env=True, premap=True):
        """Fused ``output2env.output2env`` + ``forwardEnv``: ``(env or None, colorDiffuse, colorSpec)``.
        ``premap``: True / 1 = ``lambOrig, weightOrig`` are the decoders' outputs in [0, 1] (what ``output2env.output2env`` takes);
        False / 0 = post-tan values (``fromSGtoIm``); 3 = the three decoders' last-convolution outputs, their output activations
        (models.py:336-346) run as the kernels' prologue (SG gradients only)."""
        _, _, R, C = _sg_dims(axisOrig)
        self._check_grid(R, C)
        a, n, r = _prepool(diffusePred, normalPred, roughPred, R, C)
        pm = 3 if premap == 3 and premap is not True else int(bool(premap))
        # the post-tan sharpness / intensity leave the forward kernel only when a backward will read them
        want_tan = pm == 1 and _ops.tan_handoff() and torch.is_grad_enabled() and (axisOrig.requires_grad or lambOrig.requires_grad or weightOrig.requires_grad)
        env, d, s, _, _ = _sg.fused_render(a, n, r, axisOrig, lambOrig, weightOrig, self.envHeight, self.envWidth, self.fov_deg, float(self.F0),
                                           self._cam, pm, bool(need_env), want_tan)
        return (env if need_env else None), d, s


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


def light_heads(xAxis, xLamb, xWeight, need_packed=False):
    """Output activations of the reference's three light decoders (``models.decoderLight`` modes 0 / 1 / 2,
    models.py:336-346) applied to their last convolution's outputs, in one HIP pass each way:
    ``(axisPred [bn,K,3,R,C], lambPred [bn,K,R,C], weightPred [bn,3K,R,C], envmapsPred or None)``.

    ``envmapsPred [bn,7K,R,C]`` (``need_packed=True``) is the packed cascade hand-off tensor of
    wrapperBRDFLight.py:167-168 (what ``outputBRDFLight.py`` stores as ``imenv_*.h5`` and cascade 1 reads)."""
    axis, lamb, weight, packed = _sg.light_heads(xAxis, xLamb, xWeight, bool(need_packed))
    return axis, lamb, weight, (packed if need_packed else None)


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
    if t.dim() != 4 or t.shape[1] != 7 * SGNum:
        raise RuntimeError(f"sgrender: pred must be [bn,{7 * SGNum},envRow,envCol], got {tuple(t.shape)}")
    bn, _, R, C = t.shape
    K = SGNum
    axis = t[:, 0:3 * K].reshape(bn, K, 3, R, C)
    out = _sg.sg_shading(axis, t[:, 3 * K:4 * K], t[:, 4 * K:7 * K], envHeight, envWidth, 1)
    if is_np:
        return out[0].cpu().numpy() if bn == 1 else out.cpu().numpy()
    return out


def light_albedo_scale(diffuseScaled, diffuse, specScaled, spec, albedoPred):
    """``(cLight, cAlbedo)`` of testReal.py:421-432 as 0-d device tensors (no ``.item()`` round trips): the global light /
    albedo scale derived from the ratio of the LSregressDiffSpec-scaled to the unscaled render images, clipped by the
    brightest albedo.  ``envmapsPredImage * cLight`` (testReal.py:431) then stays an asynchronous device multiply."""
    out = _sg.light_albedo_scale(diffuseScaled.detach(), diffuse.detach(), specScaled.detach(), spec.detach(), albedoPred.detach())
    return out[0], out[1]


def light_encoder_input(imBatch, albedoPred, normalPred, roughPred, depthPred, size=(480, 640)):
    """The light encoder's input of wrapperBRDFLight.py:138-156 in two HIP launches: per-image mean-normalisation of
    albedo and depth, bilinear resize of the five maps to ``size`` and their concatenation.
    Returns ``(inputBatch [bn,11,H,W], albedoPredNormalised, depthPredNormalised)`` (the wrapper returns the normalised
    maps, :139-147).  Forward only: the reference feeds ``inputBatch.detach()`` to the encoder (:158-161)."""
    H, W = int(size[0]), int(size[1])
    return tuple(_sg.light_encoder_input(imBatch.detach(), albedoPred.detach(), normalPred.detach(), roughPred.detach(), depthPred.detach(), H, W))


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
