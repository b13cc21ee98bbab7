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

from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib
from . import ops as _ops
from .layers import _prepool, _sg_dims

_sg = torch.ops.sgrender

__all__ = ["LSregress", "LSregressDiffSpec", "render_loss", "recon_loss", "combine_loss_parts", "light_objective",
           "light_objective_supported", "enable_native_allreduce", "disable_native_allreduce", "native_allreduce_enabled"]


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
    nb = pred.shape[0]
    coef = _sg.lsregress_coef(pred.detach(), gt.detach())
    return origin * coef.reshape([nb] + [1] * (origin.dim() - 1))


def LSregressDiffSpec(diff, spec, imOrig, diffOrig, specOrig):
    """Two-unknown (diffuse, specular) scale regression, then the one-unknown rescale of the clamped
    sum (models.py:23-84).  Returns ``(diffScaled, specScaled)``.

    With detached ``diff`` / ``spec`` (the trainLight / testReal pattern, wrapperBRDFLight.py:197-201) the coefficients are
    constants and come from the HIP reduction kernels; if either carries a gradient the reference differentiates through
    the coefficients (it does not detach them, models.py:44-63) and so does :func:`_lsregress_diffspec_live`."""
    if diff.shape != spec.shape or diff.shape != imOrig.shape:
        raise RuntimeError("sgrender: LSregressDiffSpec needs diff, spec and imOrig of one shape")
    if torch.is_grad_enabled() and (diff.requires_grad or spec.requires_grad):
        return _lsregress_diffspec_live(diff, spec, imOrig, diffOrig, specOrig)
    nb = diff.shape[0]
    coef = _sg.lsregress_diffspec_coef(diff.detach(), spec.detach(), imOrig.detach())
    kd = coef[:, 0].reshape(nb, 1, 1, 1)
    ks = coef[:, 1].reshape(nb, 1, 1, 1)
    return kd * diffOrig, ks * specOrig


def _sharded(group) -> bool:
    return group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


# --------------------------------------------------------------------------- #
# the loss collectives: c10d, or the extension's own in-stream RCCL all-reduce   #
# --------------------------------------------------------------------------- #
_NATIVE_COMMS = {}      # process group (None = the default group) -> communicator handle of torch.ops.sgrender


def _group_key(group):
    return None if group is None or group is dist.group.WORLD else group


def enable_native_allreduce(group=None) -> int:
    """Give ``group`` (default: the world) an RCCL communicator owned by the extension, so that the loss collectives of
    :func:`render_loss` / :func:`light_objective` / :func:`combine_loss_parts` on that group are enqueued by
    ``torch.ops.sgrender.allreduce_sum_`` on the CURRENT HIP stream -- SURVEY.md section 8e's "ncclAllReduce on the same HIP
    stream from the extension": no c10d call, no side stream, no event pair inside the step.

    COLLECTIVE: every rank of the group calls it once (``ncclCommInitRank`` synchronises the ranks); rank 0's
    ``ncclGetUniqueId`` travels through the existing c10d group.  RCCL needs one GPU per rank, so this is for the ``nccl``
    backend only; gloo groups (the CPU tests, the rehearsal mode of bench.py) keep the c10d route.  Idempotent; returns the handle."""
    key = _group_key(group)
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key]
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("sgrender: enable_native_allreduce needs an initialised torch.distributed process group")
    if dist.get_backend(group) != "nccl":
        raise RuntimeError("sgrender: the in-stream all-reduce is RCCL's; this group's backend is " + str(dist.get_backend(group)))
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    uid = _sg.comm_unique_id() if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    wire = uid.to(dev)
    dist.broadcast(wire, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    handle = int(_sg.comm_init(wire.cpu(), rank, world, dev.index))
    _NATIVE_COMMS[key] = handle
    return handle


def disable_native_allreduce(group=None) -> None:
    """Destroy the extension's communicator of ``group`` (before ``dist.destroy_process_group()``); the c10d route takes over."""
    handle = _NATIVE_COMMS.pop(_group_key(group), None)
    if handle is not None:
        _sg.comm_destroy(handle)


def _destroy_native_comms() -> None:
    for key in list(_NATIVE_COMMS):
        try:
            _sg.comm_destroy(_NATIVE_COMMS.pop(key))
        except Exception:      # interpreter shutdown: the runtime may already be on its way out
            pass


import atexit  # noqa: E402

atexit.register(_destroy_native_comms)      # a caller that forgets disable_native_allreduce() must not leave RCCL communicators to the teardown order


def native_allreduce_enabled(group=None) -> bool:
    return _group_key(group) in _NATIVE_COMMS


def _allreduce_sum_(t: torch.Tensor, group) -> None:
    """In place, on the current stream when the group has a native communicator; else ``dist.all_reduce``."""
    handle = _NATIVE_COMMS.get(_group_key(group))
    if handle is not None and t.is_cuda:
        _sg.allreduce_sum_(t, handle)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


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
        _allreduce_sum_(pair, group)
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
    if _sharded(group):
        # this shard's [numerator, raw denominator] (three launches), their sum over the ranks (one all-reduce of two floats: RCCL over
        # xGMI), then the loss value and its gradient scale from the GLOBAL totals (one launch; the autograd node sits on this operator:
        # every rank gets the gradient of the global loss w.r.t. its shard).  Five launches + one collective per step, no torch glue.
        with torch.no_grad():
            _, _, parts, rendered, im_s, seg_s, coef = _sg.render_loss(diffuse, spec, im, seg, envRow, envCol, False)
            _allreduce_sum_(parts, group)
        loss, _ = _sg.render_loss_finalize(diffuse, spec, parts, im_s, seg_s, coef)
        return loss, rendered
    # one rank: three launches, the third pass forms the loss value itself; the backward is one more (four small launches per step
    # between the two heavy kernels of a training step, where a dozen 5-us launches would be a tenth of the step)
    loss, _, _, rendered, _, _, _ = _sg.render_loss(diffuse, spec, im, seg, envRow, envCol, True)
    return loss, rendered


def _recon_parts(env, env_gt, seg_small, env_ind, offset: float):
    """``(num, den_raw, coef)`` of this rank's shard (sgr_recon_loss_fwd / sgr_recon_loss_bwd behind one operator)."""
    parts, _, coef = _sg.recon_loss_parts(env, env_gt, seg_small, env_ind, float(offset))
    return parts[0], parts[1], coef


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
    num, den, coef = _recon_parts(envmapsPredImage, envmapsBatch, seg_s, envmapsIndBatch, offset)
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
    bn, K, R, C = _sg_dims(axisPred)
    impl._check_grid(R, C)
    if not light_objective_supported(K, R, C, impl.envHeight, impl.envWidth):
        # other direction grids / more than 12 lobes: same objective from the unfused HIP kernels (env image materialised)
        env, diffuse, spec = impl.forwardSG(albedoPred, normalPred, roughPred, axisPred, lambPred, weightPred, need_env=True)
        render_err, rendered = render_loss(diffuse, spec, imBatch, segBRDFBatch, R, C, group)
        seg_s = segBRDFBatch if tuple(segBRDFBatch.shape[2:]) == (R, C) else F.adaptive_avg_pool2d(segBRDFBatch, (R, C))
        num, den, coef = _recon_parts(env, envmapsBatch, seg_s, envmapsIndBatch, offset)
        recon_err = combine_loss_parts(num, den, group, divisor=3.0 * impl.envHeight * impl.envWidth)
        objective = float(renderWeight) * render_err + float(reconWeight) * recon_err
        return objective, render_err.detach(), recon_err.detach(), rendered, coef
    a, n, r = _prepool(albedoPred, normalPred, roughPred, R, C)
    im, seg = imBatch, segBRDFBatch
    h, w = im.shape[2], im.shape[3]
    if (h, w) != (R, C) and (h, w) != (2 * R, 2 * C):
        im = F.adaptive_avg_pool2d(im, (R, C))
        seg = F.adaptive_avg_pool2d(seg, (R, C))
    eh, ew, fov, F0, cam = impl.envHeight, impl.envWidth, impl.fov_deg, float(impl.F0), impl._cam
    handoff = _ops.tan_handoff() and not heads
    # differentiable w.r.t. the SG parameters only -- on BOTH routes (the one-rank operator raised, the sharded route ran its
    # stage operators under no_grad and silently returned no gradient for a BRDF map that still carried one)
    if torch.is_grad_enabled():
        for name, t in (("albedoPred", albedoPred), ("normalPred", normalPred), ("roughPred", roughPred), ("imBatch", imBatch), ("segBRDFBatch", segBRDFBatch),
                        ("envmapsBatch", envmapsBatch), ("envmapsIndBatch", envmapsIndBatch)):
            if t.requires_grad:
                raise RuntimeError(f"sgrender: light_objective is differentiable with respect to the SG parameters only; {name} requires grad -- "
                                   "detach it (wrapperBRDFLight.py:194 detaches albedoPred; the other maps come from frozen networks), or use "
                                   "forwardSG + render_loss + recon_loss for gradients with respect to the BRDF maps")
    if not _sharded(group):
        # one operator: forward statistics pass -> render loss -> [render-loss backward -> the objective's backward pass, which also
        # yields the reconstruction loss value and the scalar tail]; under torch.no_grad() / without a grad-requiring SG input the
        # bracketed half shrinks to a loss-only pass (no gradient kernels are launched)
        return tuple(_sg.light_objective(a, n, r, axisPred, lambPred, weightPred, im, seg, envmapsBatch, envmapsIndBatch, eh, ew, fov, F0, cam,
                                         float(renderWeight), float(reconWeight), float(offset), heads, handoff))
    # batch sharded over ranks: three stage operators with the two collectives between them
    need = torch.is_grad_enabled() and (axisPred.requires_grad or lambPred.requires_grad or weightPred.requires_grad)
    with torch.no_grad():
        diffuse, spec, mask, coef, im_s, seg_s, rendered, coef_ds, sums, ws, lam_t, w_t = _sg.light_objective_stage1(
            a, n, r, axisPred, lambPred, weightPred, im, seg, envmapsBatch, envmapsIndBatch, eh, ew, fov, F0, cam, heads, handoff and need)
        _allreduce_sum_(sums, group)      # [num_r, den_r, 0, den_e]: one all-reduce before the backward pass
        render_err, g_axis, g_lamb, g_weight, parts_b = _sg.light_objective_stage2(
            a, n, r, axisPred, lambPred, weightPred, envmapsBatch, mask, coef, diffuse, spec, im_s, seg_s, coef_ds, sums, ws, lam_t, w_t,
            eh, ew, fov, F0, cam, float(renderWeight), float(reconWeight), float(offset), heads, need)
        num_e = parts_b[0:1]
        _allreduce_sum_(num_e, group)     # the second and last collective: the reconstruction numerator
        objective, recon_err = _sg.light_objective_stage3(render_err, num_e, sums, float(renderWeight), float(reconWeight), eh, ew)
    if need:      # every rank holds the gradient of the GLOBAL objective w.r.t. ITS shard (see combine_loss_parts)
        applied = torch.ones(2, device=objective.device, dtype=torch.float32)
        objective = _sg.attach_grads(objective, axisPred, lambPred, weightPred, g_axis, g_lamb, g_weight, applied)
    return objective, render_err, recon_err, rendered, coef
