"""Scale-invariant regressions and the render loss, backed by libsgrender.so.

Drop-in free functions (same names / argument meaning as the reference's ``models.py``):

  ``LSregress(pred, gt, origin)``                                     models.py:7-21
  ``LSregressDiffSpec(diff, spec, imOrig, diffOrig, specOrig)``       models.py:23-84

and the fused loss glue of ``wrapperBRDFLight.py:170-171,192,197-207``:

  ``render_loss(diffuse, spec, im, seg, envRow, envCol, group=None)``

which keeps ``pixelNum`` on the device (the reference synchronises with ``.item()``) and, when
the batch is sharded over ranks, all-reduces the ``[numerator, denominator]`` pair so that the
loss is the reference's batch-global ratio.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib
from . import ops as _ops
from .layers import _check_brdf, _check_sg, _dirs, _prepool, _ptr, _require_hip, _stream, _view

__all__ = ["LSregress", "LSregressDiffSpec", "render_loss", "recon_loss", "combine_loss_parts", "light_objective",
           "light_objective_supported"]


def _workspace(bn: int, dev) -> torch.Tensor:
    return torch.empty(_lib.load().sgr_loss_workspace_floats(bn), device=dev, dtype=torch.float32)


def _lsregress_diffspec_live(diff, spec, imOrig, diffOrig, specOrig):
    """models.py:23-84 for call sites that differentiate THROUGH the regression coefficients (the reference does not
    detach ``coefDiffuse / coefSpecular``; ``trainFineTune*_cascade1.py`` pass live ``diffusePred / specularPred`` as the
    first two arguments).  Off the trainLight hot path, so plain torch ops on the device tensors with autograd doing the
    rest: the normal equations of the masked two-column least-squares problem per image (Gram matrix by one batched
    matmul), Cramer's rule with the reference's floors, its one-column fallback where the determinant per element is
    below 1e-2 (a constant indicator), the [0, 1000] clamp, and the second, detached one-unknown rescale of the clamped sum."""
    nb, n = diff.shape[0], diff[0].numel()
    keep = (imOrig < 0.9).to(diff.dtype)
    cols = torch.stack([(diff * keep).reshape(nb, -1), (spec * keep).reshape(nb, -1)], dim=1)      # [nb, 2, n]
    rhs = (imOrig * keep).reshape(nb, -1, 1)                                                      # [nb, n, 1]
    gram = cols @ cols.transpose(1, 2)                                                            # [nb, 2, 2]
    proj = (cols @ rhs).squeeze(-1)                                                               # [nb, 2]
    g_dd, g_ss, g_ds = gram[:, 0, 0], gram[:, 1, 1], gram[:, 0, 1]
    det = g_dd * g_ss - g_ds * g_ds
    floor = torch.clamp(det, min=1e-2)
    kd_two = (proj[:, 0] * g_ss - proj[:, 1] * g_ds) / floor
    ks_two = (g_dd * proj[:, 1] - proj[:, 0] * g_ds) / floor
    kd_one = torch.clamp(proj[:, 0] / torch.clamp(g_dd, min=1e-5), 0.001, 1000.0)
    two = (det.detach() / n) > 1e-2
    kd = torch.clamp(torch.where(two, kd_two, kd_one), 0.0, 1000.0).reshape(nb, 1, 1, 1)
    ks = torch.clamp(torch.where(two, ks_two, torch.zeros_like(ks_two)), 0.0, 1000.0).reshape(nb, 1, 1, 1)
    d_scaled, s_scaled = kd * diffOrig, ks * specOrig
    with torch.no_grad():
        ren = torch.clamp(d_scaled + s_scaled, 0.0, 1.0).reshape(nb, -1)
        again = (ren * imOrig.reshape(nb, -1)).sum(1) / torch.clamp((ren * ren).sum(1), min=1e-5)
        again = torch.clamp(again, 0.001, 1000.0).reshape(nb, 1, 1, 1)
    return again * d_scaled, again * s_scaled


def LSregress(pred, gt, origin):
    """``origin * clamp(<pred,gt> / max(<pred,pred>, 1e-5), 1e-3, 1e3)`` per image (models.py:7-21).

    Like the reference, the coefficient is a constant in backward whatever ``pred`` carries (models.py:13 detaches it),
    so grad-carrying ``pred`` tensors are fine (trainBRDF.py:249-254 passes ``albedoPred * seg``); the gradient flows
    through ``origin`` only."""
    dev = _require_hip(pred, gt, origin)
    nb = pred.shape[0]
    p, g = pred.detach().contiguous(), gt.detach().contiguous()
    coef = torch.empty(nb, device=dev, dtype=torch.float32)
    ws = _workspace(nb, dev)
    with torch.cuda.device(dev):
        _lib.call("sgr_lsregress_coef", _ptr(p), _ptr(g), _ptr(coef), _ptr(ws), nb, p.numel() // nb, _stream(dev))
    return origin * coef.reshape([nb] + [1] * (origin.dim() - 1))


def LSregressDiffSpec(diff, spec, imOrig, diffOrig, specOrig):
    """Two-unknown (diffuse, specular) scale regression, then the one-unknown rescale of the clamped
    sum (models.py:23-84).  Returns ``(diffScaled, specScaled)``.

    With detached ``diff`` / ``spec`` (the trainLight / testReal pattern, wrapperBRDFLight.py:197-201) the coefficients are
    constants and come from the HIP reduction kernels; if either carries a gradient the reference differentiates through
    the coefficients (it does not detach them, models.py:44-63) and so does :func:`_lsregress_diffspec_live`."""
    dev = _require_hip(diff, spec, imOrig, diffOrig, specOrig)
    if diff.shape != spec.shape or diff.shape != imOrig.shape:
        raise RuntimeError("sgrender: LSregressDiffSpec needs diff, spec and imOrig of one shape")
    if torch.is_grad_enabled() and (diff.requires_grad or spec.requires_grad):
        return _lsregress_diffspec_live(diff, spec, imOrig, diffOrig, specOrig)
    nb = diff.shape[0]
    d, s, im = diff.detach().contiguous(), spec.detach().contiguous(), imOrig.detach().contiguous()
    if d.shape != s.shape or d.shape != im.shape:
        raise RuntimeError("sgrender: LSregressDiffSpec needs diff, spec and imOrig of one shape")
    coef = torch.empty((nb, 2), device=dev, dtype=torch.float32)
    ws = _workspace(nb, dev)
    with torch.cuda.device(dev):
        _lib.call("sgr_lsregress_diffspec_coef", _ptr(d), _ptr(s), _ptr(im), _ptr(coef), _ptr(ws), nb, d.numel() // nb,
                  _stream(dev))
    kd = coef[:, 0].reshape(nb, 1, 1, 1)
    ks = coef[:, 1].reshape(nb, 1, 1, 1)
    return kd * diffOrig, ks * specOrig


_CONSTS = {}


def _const_scalar(dev, value: float) -> torch.Tensor:
    """A cached one-element device tensor holding ``value`` (loss weights handed to kernels as device scalars)."""
    key = (str(dev), float(value))
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 256:
            _CONSTS.clear()
        t = _CONSTS[key] = torch.full((1,), float(value), device=dev, dtype=torch.float32)
    return t


def _sharded(group) -> bool:
    return group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


class _RenderLoss(torch.autograd.Function):
    """``(renderErr, rendered)``: sgr_render_loss_fwd -> [all-reduce of the two totals] -> sgr_loss_finalize, and
    sgr_render_loss_bwd_scaled.  Four small launches per step (five + the collective when sharded) and no elementwise torch glue: this sits between the two heavy
    kernels of every training step, where a dozen 5-us launches are worth a tenth of the step."""

    @staticmethod
    def forward(ctx, diffuse, spec, im, seg, R: int, C: int, group):
        dev = _require_hip(diffuse, spec, im, seg)
        d, s, im_c, seg_c = diffuse.contiguous(), spec.contiguous(), im.contiguous(), seg.contiguous()
        bn = d.shape[0]
        if tuple(d.shape) != (bn, 3, R, C) or tuple(s.shape) != (bn, 3, R, C):
            raise RuntimeError(f"sgrender: diffuse/spec must be [bn,3,{R},{C}]")
        imH, imW = im_c.shape[2], im_c.shape[3]
        if tuple(seg_c.shape) != (bn, 1, imH, imW) or im_c.shape[1] != 3:
            raise RuntimeError("sgrender: im must be [bn,3,h,w] and seg [bn,1,h,w]")
        im_s = torch.empty((bn, 3, R, C), device=dev, dtype=torch.float32)
        seg_s = torch.empty((bn, 1, R, C), device=dev, dtype=torch.float32)
        rendered = torch.empty_like(im_s)
        coef = torch.empty((bn, 2), device=dev, dtype=torch.float32)
        parts = torch.empty(2, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)      # returned as it is (0-d, no view); `scale` = d loss / d numerator is kept for backward
        scale = torch.empty(1, device=dev, dtype=torch.float32)      # (separate buffers: no shared version counter)
        ws = _workspace(bn, dev)
        with torch.cuda.device(dev):
            if _sharded(group):
                _lib.call("sgr_render_loss_fwd", _ptr(d), _ptr(s), _ptr(im_c), _ptr(seg_c), _ptr(im_s), _ptr(seg_s),
                          _ptr(rendered), _ptr(coef), _ptr(parts), _ptr(ws), bn, R, C, imH, imW, _stream(dev))
                dist.all_reduce(parts, op=dist.ReduceOp.SUM, group=group)      # [num, den] of the global batch (RCCL over xGMI)
                _lib.call("sgr_loss_finalize", _ptr(parts), _ptr(loss), _ptr(scale), 3.0, _stream(dev))
            else:                                                             # the third pass forms the loss value itself
                _lib.call("sgr_render_loss_fwd_total", _ptr(d), _ptr(s), _ptr(im_c), _ptr(seg_c), _ptr(im_s), _ptr(seg_s),
                          _ptr(rendered), _ptr(coef), _ptr(parts), _ptr(loss), _ptr(scale), 3.0, _ptr(ws), bn, R, C, imH, imW,
                          _stream(dev))
        ctx.save_for_backward(d, s, im_s, seg_s, coef, scale)
        ctx.mark_non_differentiable(rendered)
        ctx.set_materialize_grads(False)      # no zero image for the unused cotangent of `rendered` on every backward
        return loss, rendered

    @staticmethod
    def backward(ctx, g_loss, _g_ren):
        if g_loss is None:
            return (None,) * 7
        d, s, im_s, seg_s, coef, scale = ctx.saved_tensors
        dev = d.device
        bn, _, R, C = d.shape
        g_loss = g_loss.contiguous().reshape(1).to(torch.float32)
        g_d, g_s = torch.empty_like(d), torch.empty_like(s)
        with torch.cuda.device(dev):
            _lib.call("sgr_render_loss_bwd_scaled", _ptr(g_loss), _ptr(scale), _ptr(d), _ptr(s), _ptr(im_s), _ptr(seg_s), _ptr(coef),
                      _ptr(g_d), _ptr(g_s), bn, R, C, _stream(dev))
        return g_d, g_s, None, None, None, None, None


def ddp_loss_scale(group=None) -> float:
    """Factor that makes DistributedDataParallel's gradient AVERAGING reproduce the single-process gradient of the
    batch-global losses of this module (see :func:`combine_loss_parts`): the world size of ``group``; 1.0 when
    torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return float(dist.get_world_size(group))
    return 1.0


def combine_loss_parts(num: torch.Tensor, den_raw: torch.Tensor, group=None, divisor: float = 3.0) -> torch.Tensor:
    """``loss = sum_ranks(num) / max(sum_ranks(den), 1e-5) / divisor`` with the gradient
    ``d loss / d num_local = 1 / (divisor * den_global)``.

    The reference's losses are ratios of two batch-global sums (wrapperBRDFLight.py:192,205-207), not
    means of per-image losses, so under batch sharding the pair is summed across ranks first: one
    all-reduce of two floats (RCCL over xGMI on the GPU path; pure torch + torch.distributed, so the
    same code runs under gloo in the CPU tests).

    Gradient convention under sharding: every rank receives the gradient of the GLOBAL loss with respect to ITS
    shard's inputs -- summed over ranks that is the single-process (reference: nn.DataParallel) gradient.
    ``DistributedDataParallel`` AVERAGES parameter gradients over ranks, which would leave 1/world_size of it:
    multiply the loss by :func:`ddp_loss_scale` (= world size) before ``backward()``, or build DDP with a SUM
    communication hook (tests/test_sharded_loss_gloo.py::test_ddp_gradients_match_single_process)."""
    if _sharded(group):
        pair = torch.stack([num.detach(), den_raw.detach().to(num.dtype)])
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
        num_g, den_g = pair[0], pair[1]
    else:
        num_g, den_g = num.detach(), den_raw.detach()
    den_c = torch.clamp(den_g, min=1e-5)
    return (num + (num_g - num.detach())) / den_c / divisor


def render_loss(diffuse, spec, im, seg, envRow: int, envCol: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(renderErr, renderedImPred)`` of wrapperBRDFLight.py:170-171,192,197-207.

    ``diffuse, spec [bn,3,envRow,envCol]`` (outputs of the render layer), ``im [bn,3,h,w]``,
    ``seg [bn,1,h,w]`` (``segBRDFBatch``).  Image sizes other than 1x / 2x the env grid are
    average-pooled with torch first (same window arithmetic as the reference)."""
    h, w = im.shape[2], im.shape[3]
    if (h, w) != (envRow, envCol) and (h, w) != (2 * envRow, 2 * envCol):
        im = F.adaptive_avg_pool2d(im, (envRow, envCol))
        seg = F.adaptive_avg_pool2d(seg, (envRow, envCol))
    return _RenderLoss.apply(diffuse, spec, im, seg, envRow, envCol, group)


class _ReconLossParts(torch.autograd.Function):
    """sgr_recon_loss_fwd / sgr_recon_loss_bwd: ``(num, den_raw, coef)`` of this rank's shard."""

    @staticmethod
    def forward(ctx, env, env_gt, seg_small, env_ind, offset: float):
        dev = _require_hip(env, env_gt, seg_small, env_ind)
        e, g = env.contiguous(), env_gt.contiguous()
        if e.dim() != 6 or e.shape != g.shape or e.shape[1] != 3:
            raise RuntimeError("sgrender: envmapsPred / envmaps must both be [bn,3,envRow,envCol,envHeight,envWidth]")
        bn, _, R, C, eh, ew = e.shape
        sm = seg_small.contiguous().reshape(bn, R * C)
        ind = env_ind.contiguous().reshape(bn)
        mask = torch.empty((bn, R * C), device=dev, dtype=torch.float32)
        coef = torch.empty(bn, device=dev, dtype=torch.float32)
        parts = torch.empty(2, device=dev, dtype=torch.float32)
        ws = torch.empty(_lib.load().sgr_recon_workspace_floats(bn, R, C), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("sgr_recon_loss_fwd", _ptr(e), _ptr(g), _ptr(sm), _ptr(ind), _ptr(mask), _ptr(coef), _ptr(parts),
                      _ptr(ws), bn, R, C, eh, ew, float(offset), _stream(dev))
        ctx.save_for_backward(e, g, mask, coef)
        ctx.offset = float(offset)
        num, den = parts[0], parts[1]
        ctx.mark_non_differentiable(den, coef)
        ctx.set_materialize_grads(False)
        return num, den, coef

    @staticmethod
    def backward(ctx, g_num, _g_den, _g_coef):
        if g_num is None:
            return (None,) * 5
        e, g, mask, coef = ctx.saved_tensors
        dev = e.device
        bn, _, R, C, eh, ew = e.shape
        g_num = g_num.contiguous().reshape(1).to(torch.float32)
        g_env = torch.empty_like(e)
        with torch.cuda.device(dev):
            _lib.call("sgr_recon_loss_bwd", _ptr(g_num), _ptr(e), _ptr(g), _ptr(mask), _ptr(coef), _ptr(g_env),
                      bn, R, C, eh, ew, ctx.offset, _stream(dev))
        return g_env, None, None, None, None


def recon_loss(envmapsPredImage, envmapsBatch, segBRDFBatch, envmapsIndBatch, envRow: int, envCol: int, offset: float = 1.0,
               group=None, return_scaled: bool = False):
    """``reconstErr`` of wrapperBRDFLight.py:171-188 (log-L2 between the LSregress-scaled predicted env image
    and the ground truth, masked by the pooled object mask, the per-image env indicator and the
    not-dark test), computed by two streaming HIP passes; ``pixelNum`` stays on the device and the
    ``[num, den]`` pair is all-reduced when the batch is sharded.

    With ``return_scaled=True`` also returns ``envmapsPredScaledImage`` (= pred * coef, a torch broadcast
    multiply -- the reference returns it for logging)."""
    seg_s = segBRDFBatch
    if tuple(seg_s.shape[2:]) != (envRow, envCol):
        seg_s = F.adaptive_avg_pool2d(segBRDFBatch, (envRow, envCol))
    num, den, coef = _ReconLossParts.apply(envmapsPredImage, envmapsBatch, seg_s, envmapsIndBatch, offset)
    eh, ew = envmapsPredImage.shape[4], envmapsPredImage.shape[5]
    err = combine_loss_parts(num, den, group, divisor=3.0 * eh * ew)
    if return_scaled:
        return err, envmapsPredImage * coef.reshape(-1, 1, 1, 1, 1, 1)
    return err


# --------------------------------------------------------------------------- #
# the whole trainLight objective, env image never materialised                 #
# --------------------------------------------------------------------------- #
def light_objective_supported(SGNum: int, envRow: int, envCol: int, envHeight: int = 8, envWidth: int = 16) -> bool:
    """Whether :func:`light_objective` has a fused kernel for this configuration (envWidth 16 or 32, SGNum <= 24: the reference's
    8x16 training grid and the 16x32 grid of its ground-truth envmaps, BASELINE config 5)."""
    return bool(_lib.load().sgr_fused_recon_supported(int(SGNum), int(envRow), int(envCol), int(envHeight), int(envWidth)))


def _global_pair(a: torch.Tensor, b: torch.Tensor, group):
    """Sum a pair of device scalars over the ranks of ``group`` (one all-reduce of two floats)."""
    if group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        pair = torch.stack([a, b])
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
        return pair[0], pair[1], True
    return a, b, False


class _LightObjective(torch.autograd.Function):
    """sgr_fused_fwd_recon -> sgr_render_loss_fwd/bwd -> sgr_fused_bwd_recon.

    The loss value of the reconstruction term comes out of the same pass as the SG gradients, so the
    gradients are produced here and handed out (times the incoming cotangent) in ``backward``."""

    @staticmethod
    def forward(ctx, albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind, layer_cfg, ren_w, rec_w, offset, group, heads=False):
        dev = _require_hip(albedo, normal, rough, axis, lamb, weight, im, seg, env_gt, env_ind)
        eh, ew, fov, F0, cam = layer_cfg
        albedo_c, normal_c, rough_c = albedo.contiguous(), normal.contiguous(), rough.contiguous()
        axis_c, lamb_c, weight_c = axis.contiguous(), lamb.contiguous(), weight.contiguous()
        bn, K, R, C = _check_sg(axis_c, lamb_c, weight_c, None)
        bn2, h, w = _check_brdf(albedo_c, normal_c, rough_c)
        gt = env_gt.contiguous()
        if bn2 != bn or tuple(gt.shape) != (bn, 3, R, C, eh, ew):
            raise RuntimeError(f"sgrender: envmapsBatch must be [bn,3,{R},{C},{eh},{ew}] and the batch sizes must agree")
        im_c, seg_c = im.contiguous(), seg.contiguous()
        imH, imW = im_c.shape[2], im_c.shape[3]
        if tuple(seg_c.shape) != (bn, 1, imH, imW) or im_c.shape[1] != 3:
            raise RuntimeError("sgrender: im must be [bn,3,h,w] and seg [bn,1,h,w]")
        ind = env_ind.contiguous().reshape(bn)
        lib = _lib.load()
        f32 = dict(device=dev, dtype=torch.float32)
        diffuse, spec = torch.empty((bn, 3, R, C), **f32), torch.empty((bn, 3, R, C), **f32)
        im_s, seg_s = torch.empty((bn, 3, R, C), **f32), torch.empty((bn, 1, R, C), **f32)
        rendered = torch.empty_like(im_s)
        mask, coef = torch.empty((bn, R * C), **f32), torch.empty(bn, **f32)
        coef_ds = torch.empty((bn, 2), **f32)
        parts_f, parts_r, parts_b = torch.empty(2, **f32), torch.empty(2, **f32), torch.empty(2, **f32)
        ws = torch.empty(lib.sgr_fused_recon_workspace_floats(bn, R, C), **f32)
        ws_r = _workspace(bn, dev)
        g_axis, g_lamb, g_weight = torch.empty_like(axis_c), torch.empty_like(lamb_c), torch.empty_like(weight_c)
        handoff = _ops.tan_handoff() and not heads
        pm = 3 if heads else 1      # 3: axis / lamb / weight are the decoders' last-convolution outputs (heads as the kernels' prologue)
        # post-tan values: written by the forward pass, read by the backward pass (premap mode 2)
        lam_t, w_t = (torch.empty_like(lamb_c), torch.empty_like(weight_c)) if handoff else (lamb_c, weight_c)
        d, v = _dirs(dev, eh, ew), _view(dev, R, C, fov, cam)
        st = _stream(dev)
        sg_args = (_ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(axis_c), _ptr(lamb_c), _ptr(weight_c), _ptr(d), _ptr(v))
        sg_args_tan = (_ptr(albedo_c), _ptr(normal_c), _ptr(rough_c), _ptr(axis_c), _ptr(lam_t), _ptr(w_t), _ptr(d), _ptr(v))
        with torch.cuda.device(dev):
            sharded = _sharded(group)
            # the env mask needs the pooled object mask before the render-loss pass produces it: the kernel pools 2x2 itself
            _lib.call("sgr_fused_fwd_recon_seg", *sg_args, _ptr(gt), _ptr(seg_c), imH, imW, _ptr(ind), _ptr(lam_t) if handoff else None,
                      _ptr(w_t) if handoff else None, _ptr(diffuse),
                      _ptr(spec), _ptr(mask), _ptr(coef), _ptr(parts_f) if sharded else None, _ptr(ws), bn, K, R, C, eh, ew, h, w, float(F0), pm, st)
            render_err, scale_r = torch.empty((), **f32), torch.empty(1, **f32)
            _lib.call("sgr_render_loss_fwd_total", _ptr(diffuse), _ptr(spec), _ptr(im_c), _ptr(seg_c), _ptr(im_s), _ptr(seg_s),
                      _ptr(rendered), _ptr(coef_ds), _ptr(parts_r), None if sharded else _ptr(render_err), None if sharded else _ptr(scale_r),
                      3.0, _ptr(ws_r), bn, R, C, imH, imW, st)
            # everything between the heavy kernels stays on the device: one rank -- the scalar tails ride in the folds of the passes
            # that produce their inputs (ten launches per objective, eight of them small); sharded -- two one-thread launches
            # around the collectives (a dozen one-element torch kernels before round 3: 0.05 ms of a 0.86 ms training step)
            recon_err, objective = torch.empty((), **f32), torch.empty((), **f32)
            den_e_c = None
            if sharded:
                v1 = torch.stack([parts_r[0], parts_r[1], parts_f[1]])      # [num_r, den_r, den_e]: one all-reduce before the backward pass
                dist.all_reduce(v1, op=dist.ReduceOp.SUM, group=group)
                parts_r, den_e_c = v1[:2], v1[2:3]
                _lib.call("sgr_loss_finalize", _ptr(parts_r), _ptr(render_err), _ptr(scale_r), 3.0, st)
            g_d, g_s = torch.empty_like(diffuse), torch.empty_like(spec)
            _lib.call("sgr_render_loss_bwd_scaled", _ptr(_const_scalar(dev, float(ren_w))), _ptr(scale_r), _ptr(diffuse), _ptr(spec), _ptr(im_s),
                      _ptr(seg_s), _ptr(coef_ds), _ptr(g_d), _ptr(g_s), bn, R, C, st)
            # `applied`: two slots, the cotangent the stored gradients are currently scaled by in applied[ctx.parity] (see backward)
            if sharded:
                _lib.call("sgr_fused_bwd_recon", *sg_args_tan, _ptr(gt), _ptr(mask), _ptr(coef), _ptr(den_e_c), _ptr(g_d), _ptr(g_s),
                          _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight), _ptr(parts_b), _ptr(ws),
                          bn, K, R, C, eh, ew, h, w, float(F0), 2 if handoff else pm, float(offset), float(rec_w), st)
                num_e = parts_b[0:1].clone()
                dist.all_reduce(num_e, op=dist.ReduceOp.SUM, group=group)      # the second and last collective: the reconstruction numerator
                parts_b = torch.cat([num_e, den_e_c])
                _lib.call("sgr_objective_finalize", _ptr(render_err), _ptr(parts_b), float(ren_w), float(rec_w), 3.0 * eh * ew,
                          _ptr(objective), _ptr(recon_err), st)
                applied = torch.ones(2, **f32)
            else:
                applied = torch.empty(2, **f32)
                _lib.call("sgr_fused_bwd_recon_total", *sg_args_tan, _ptr(gt), _ptr(mask), _ptr(coef), _ptr(g_d), _ptr(g_s),
                          _ptr(g_axis), _ptr(g_lamb), _ptr(g_weight), _ptr(parts_b), _ptr(ws),
                          bn, K, R, C, eh, ew, h, w, float(F0), 2 if handoff else pm, float(offset), float(rec_w), _ptr(render_err),
                          float(ren_w), _ptr(objective), _ptr(recon_err), _ptr(applied), st)
        ctx.parity = 0
        ctx.save_for_backward(g_axis, g_lamb, g_weight, applied)
        ctx.mark_non_differentiable(render_err, recon_err, rendered, coef)
        ctx.set_materialize_grads(False)      # otherwise every backward zero-fills cotangents for the four reported outputs (one of them an image)
        return objective, render_err, recon_err, rendered, coef

    @staticmethod
    def backward(ctx, g_obj, *_unused):
        if g_obj is None:
            return (None,) * 16
        g_axis, g_lamb, g_weight, applied = ctx.saved_tensors
        if any(ctx.needs_input_grad[:3]) or any(ctx.needs_input_grad[6:10]):
            raise NotImplementedError("sgrender: light_objective differentiates w.r.t. the SG parameters only "
                                      "(trainLight mode, wrapperBRDFLight.py:194 detaches the BRDF maps)")
        # the gradients exist already; scale them by the incoming cotangent on the device (a no-op kernel when it
        # equals what they are scaled by already -- 1 for a plain objective.backward())
        dev = g_axis.device
        gs = (g_axis, g_lamb, g_weight)
        if getattr(ctx, "handed_out", False):      # a second backward through this node (retain_graph): the buffers
            # may be somebody's .grad by now -- leave them alone.  If the first backward came with a zero cotangent the
            # stored gradients were scaled to zero in place and cannot be recovered: say so instead of returning inf/NaN
            # (rare path, so the host sync is acceptable)
            if float(applied[ctx.parity].item()) == 0.0:
                raise RuntimeError("sgrender: light_objective was first back-propagated with a zero cotangent; its stored "
                                   "gradients are gone -- re-evaluate the objective instead of reusing the graph")
            f = g_obj.detach() / applied[ctx.parity]
            return tuple([None] * 3 + [g * f if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(gs)] + [None] * 10)
        ctx.handed_out = True
        ptrs = (ctypes.c_void_p * 3)(*[g.data_ptr() for g in gs])
        lens = (ctypes.c_longlong * 3)(*[g.numel() for g in gs])
        scale = g_obj.detach().to(torch.float32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            _lib.call("sgr_rescale_inplace_flip", ctypes.addressof(ptrs), ctypes.addressof(lens), 3, _ptr(scale), _ptr(applied), ctx.parity,
                      _stream(dev))
        ctx.parity = 1 - ctx.parity
        outs = [None] * 16
        for i, g in ((3, g_axis), (4, g_lamb), (5, g_weight)):
            if ctx.needs_input_grad[i]:
                outs[i] = g
        return tuple(outs)


def light_objective(renderLayer, albedoPred, normalPred, roughPred, axisPred, lambPred, weightPred, imBatch, segBRDFBatch,
                    envmapsBatch, envmapsIndBatch, renderWeight: float = 1.0, reconWeight: float = 10.0, offset: float = 1.0,
                    group=None, decoder_outputs: bool = False):
    """The cascade-0 light objective ``renderWeight * renderErr + reconWeight * reconstErr`` of
    wrapperBRDFLight.py:167-207 / trainLight.py:237 in two heavy kernel passes, without ever writing the
    predicted env image or its gradient.

    ``renderLayer`` is a :class:`renderingLayer` (its fov / F0 / camera / direction grid are used);
    ``axisPred, lambPred, weightPred`` are the raw decoder outputs (pre-tan), exactly what
    ``output2env.output2env`` takes.  Differentiable w.r.t. those three only.

    Configurations without a fused kernel (envWidth other than 16 / 32 or SGNum > 24, see :func:`light_objective_supported`)
    are evaluated by the unfused HIP kernels (forwardSG + render_loss + recon_loss) with the same return values.

    ``decoder_outputs=True``: ``axisPred [bn,3K,R,C] (or [bn,K,3,R,C]), lambPred, weightPred`` are the three light decoders'
    LAST-CONVOLUTION outputs instead -- their output activations (models.py:336-346: ``1.01 tanh`` -> unit axis /
    ``clamp(0.5 (. + 1), 0, 1)``) run as the prologue of the two heavy kernels and their chain rule as the epilogue of the
    backward one (SURVEY.md section 8f rank 2 as written), so the activated SG parameters and their gradients never exist in
    HBM; the gradients come back w.r.t. the raw outputs.  Where the fused kernels do not cover the configuration
    (``sgr_heads_prologue_supported``: SGNum <= 6 or > 24, other direction grids) the same result comes from
    :func:`light_heads` followed by the plain objective.

    Returns ``(objective, renderErr, reconstErr, renderedImPred, envScale)``; the two error terms are reported
    values (no gradient), ``envScale [bn]`` is the LSregress coefficient (``envmapsPredScaledImage =
    envScale * envmapsPredImage`` if the caller materialises the env for logging).  Under batch sharding the
    mask sums are all-reduced before the backward pass and the numerators after it (two collectives of two
    floats each)."""
    impl = getattr(renderLayer, "impl", renderLayer)
    heads = False
    if decoder_outputs:
        if axisPred.dim() == 4 and axisPred.shape[1] % 3 == 0:
            axisPred = axisPred.reshape(axisPred.shape[0], axisPred.shape[1] // 3, 3, axisPred.shape[2], axisPred.shape[3])
        k_, r_, c_ = axisPred.shape[1], axisPred.shape[-2], axisPred.shape[-1]
        if light_objective_supported(k_, r_, c_, impl.envHeight, impl.envWidth) and \
                _lib.load().sgr_heads_prologue_supported(int(k_), int(r_), int(c_), int(impl.envHeight), int(impl.envWidth)):
            heads = True
        else:
            from .layers import light_heads
            axisPred, lambPred, weightPred, _ = light_heads(axisPred.reshape(axisPred.shape[0], -1, r_, c_), lambPred, weightPred)
    bn, K, R, C = _check_sg(axisPred, lambPred, weightPred, None)
    impl._check_grid(R, C)
    if not light_objective_supported(K, R, C, impl.envHeight, impl.envWidth):
        # other direction grids / more than 12 lobes: same objective from the unfused HIP kernels (env image materialised)
        env, diffuse, spec = impl.forwardSG(albedoPred, normalPred, roughPred, axisPred, lambPred, weightPred, need_env=True)
        render_err, rendered = render_loss(diffuse, spec, imBatch, segBRDFBatch, R, C, group)
        seg_s = segBRDFBatch if tuple(segBRDFBatch.shape[2:]) == (R, C) else F.adaptive_avg_pool2d(segBRDFBatch, (R, C))
        num, den, coef = _ReconLossParts.apply(env, envmapsBatch, seg_s, envmapsIndBatch, offset)
        recon_err = combine_loss_parts(num, den, group, divisor=3.0 * impl.envHeight * impl.envWidth)
        objective = float(renderWeight) * render_err + float(reconWeight) * recon_err
        return objective, render_err.detach(), recon_err.detach(), rendered, coef
    a, n, r = _prepool(albedoPred, normalPred, roughPred, R, C)
    im, seg = imBatch, segBRDFBatch
    h, w = im.shape[2], im.shape[3]
    if (h, w) != (R, C) and (h, w) != (2 * R, 2 * C):
        im = F.adaptive_avg_pool2d(im, (R, C))
        seg = F.adaptive_avg_pool2d(seg, (R, C))
    cfg = (impl.envHeight, impl.envWidth, impl.fov_deg, impl.F0, impl._cam)
    return _LightObjective.apply(a, n, r, axisPred, lambPred, weightPred, im, seg, envmapsBatch, envmapsIndBatch, cfg,
                                 float(renderWeight), float(reconWeight), float(offset), group, heads)
