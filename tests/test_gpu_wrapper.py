"""GPU: the reference's cascade-0 step wrapper (wrapperBRDFLight.py:164-207) replayed call for call with the drop-in
layers swapped in, against fixtures captured from the UNMODIFIED ``wrapperBRDFLight.wrapperBRDFLight`` running seeded
random reference networks (``oracle/make_golden_wrapper.py`` -> ``tests/golden/g5_wrapper_*.npz``).

Two routes through the product, same fixtures:
  (A) drop-in:  light_heads -> output2env.output2env -> LSregress -> log-L2 (torch glue, as in the reference) ->
                renderingLayer.forwardEnv -> LSregressDiffSpec -> clamp -> masked L2 (torch glue)
  (B) fused:    light_heads -> light_objective (env image never written)
  (C) fused, decoder heads as the kernels' prologue:  light_objective(decoder_outputs=True) on the networks' last-convolution outputs
Tolerances: BASELINE.md section 3 -- rel-L2 <= 1e-4 against the reference's fp32 values, and error against the fp64
oracle evaluated on the same fp32 inputs no worse than 2 x the reference's own (floor 1e-5 where the reference is exact).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN_DIR, rel_l2

pytestmark = pytest.mark.gpu

CASES = ["g5_wrapper_small", "g5_wrapper_120x160"]
REN_W, REC_W, OFFSET = 1.0, 10.0, 1.0
TOL = 1e-4


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    bn, imH, imW, R, C, K, eh, ew, blk, s = [int(v) for v in z["meta"]]
    cfg = dict(bn=bn, imH=imH, imW=imW, R=R, C=C, K=K, eh=eh, ew=ew, blk=blk, s=s)
    t = {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in
         ("x_axis", "x_lamb", "x_weight", "albedoPred", "normalPred", "roughPred", "im", "segBRDF", "envmapsInd")}
    t["envmaps"] = torch.from_numpy(z["envmaps_blocks"]).repeat_interleave(blk, dim=2).repeat_interleave(blk, dim=3).contiguous()
    return z, cfg, t


def _oracle64(t, cfg):
    """fp64 arbiter: the oracle's restatement of the same call sequence on the same fp32 boundary tensors."""
    from oracle import sg_oracle as O
    R, C = cfg["R"], cfg["C"]
    x = {k: t[k].double().requires_grad_(True) for k in ("x_axis", "x_lamb", "x_weight")}
    a, l, w, packed = O.light_heads(x["x_axis"], x["x_lamb"], x["x_weight"])
    env, _, _, _ = O.output2env(a, l, w, cfg["eh"], cfg["ew"])
    rec, scaled, _, _ = O.recon_loss(env, t["envmaps"].double(), t["segBRDF"].double(), t["envmapsInd"].double(), R, C, OFFSET)
    d, s = O.render_env(t["albedoPred"].double(), t["normalPred"].double(), t["roughPred"].double(), env)
    ren_err, rendered, _, _ = O.render_loss(d, s, t["im"].double(), t["segBRDF"].double(), R, C)
    total = REN_W * ren_err + REC_W * rec
    g = torch.autograd.grad(total, [x["x_axis"], x["x_lamb"], x["x_weight"]])
    return dict(reconstErr=rec.item(), renderErr=ren_err.item(), rendered=rendered.detach(), diffuse=d.detach(), spec=s.detach(),
                envScaled=scaled.detach(), envmapsPred=packed.detach(), gx_axis=g[0], gx_lamb=g[1], gx_weight=g[2])


def _check(name, what, got, ref32, ref64, sub=None):
    """got vs the reference's fp32 value, and vs the fp64 oracle relative to the reference's own error."""
    got = got.detach().double().cpu()
    if sub is not None:
        got, ref64 = got[sub], ref64[sub]
    r32 = torch.as_tensor(ref32).double()
    e_ref = rel_l2(r32, ref64)
    e_hip = rel_l2(got, ref64)
    assert rel_l2(got, r32) < TOL, (name, what, "vs ref32", rel_l2(got, r32))
    assert e_hip <= max(2.0 * e_ref, 1e-5), (name, what, "vs fp64 oracle", e_hip, "reference's own", e_ref)


def _scalar_ok(got, ref32, ref64):
    e_ref = abs(ref32 - ref64)
    return abs(got - ref32) <= max(2.0 * e_ref, TOL * abs(ref32)) and abs(got - ref64) <= max(2.0 * e_ref, 1e-5 * abs(ref64))


@pytest.mark.parametrize("name", CASES)
def test_wrapper_sequence_dropin(sgr, name):
    """Route (A): wrapperBRDFLight.py:164-207, every hot-path call replaced by its drop-in, the glue left in torch."""
    z, cfg, t = _load(name)
    R, C, K, eh, ew, s = cfg["R"], cfg["C"], cfg["K"], cfg["eh"], cfg["ew"], cfg["s"]
    ref64 = _oracle64(t, cfg)
    dev = torch.device("cuda")
    g = {k: v.to(dev) for k, v in t.items()}
    xa, xl, xw = (g[k].clone().requires_grad_(True) for k in ("x_axis", "x_lamb", "x_weight"))
    output2env = sgr.output2env(SGNum=K, envWidth=ew, envHeight=eh)
    renderLayer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    imBatch, segBRDFBatch, envmapsBatch, envmapsIndBatch = g["im"], g["segBRDF"], g["envmaps"], g["envmapsInd"]
    albedoPred, normalPred, roughPred = g["albedoPred"], g["normalPred"], g["roughPred"]

    # :164-168  decoder heads + packed prediction
    axisPred, lambPred, weightPred, envmapsPred = sgr.light_heads(xa, xl, xw, need_packed=True)
    # :170-174
    imBatchSmall = F.adaptive_avg_pool2d(imBatch, (R, C))
    segBatchSmall = F.adaptive_avg_pool2d(segBRDFBatch, (R, C))
    notDarkEnv = (torch.mean(torch.mean(torch.mean(envmapsBatch, 4), 4), 1, True) > 0.001).float()
    segEnvBatch = (segBatchSmall * envmapsIndBatch.expand_as(segBatchSmall)).unsqueeze(-1).unsqueeze(-1)
    segEnvBatch = segEnvBatch * notDarkEnv.unsqueeze(-1).unsqueeze(-1)
    # :177
    envmapsPredImage, axisPred, lambPred, weightPred = output2env.output2env(axisPred, lambPred, weightPred)
    # :179-188
    pixelNum = max((torch.sum(segEnvBatch).cpu().data).item(), 1e-5)
    envmapsPredScaledImage = sgr.LSregress(envmapsPredImage.detach() * segEnvBatch.expand_as(envmapsBatch),
                                           envmapsBatch * segEnvBatch.expand_as(envmapsBatch), envmapsPredImage)
    dlog = torch.log(envmapsPredScaledImage + OFFSET) - torch.log(envmapsBatch + OFFSET)
    reconstErr = torch.sum(dlog * dlog * segEnvBatch.expand_as(envmapsPredImage)) / pixelNum / 3.0 / ew / eh
    # :192-207
    pixelNum = max((torch.sum(segBatchSmall).cpu().data).item(), 1e-5)
    diffusePred, specularPred = renderLayer.forwardEnv(albedoPred.detach(), normalPred, roughPred, envmapsPredImage)
    diffusePredScaled, specularPredScaled = sgr.LSregressDiffSpec(diffusePred.detach(), specularPred.detach(), imBatchSmall,
                                                                  diffusePred, specularPred)
    renderedImPred = torch.clamp(diffusePredScaled + specularPredScaled, 0, 1)
    renderErr = torch.sum((renderedImPred - imBatchSmall) * (renderedImPred - imBatchSmall)
                          * segBatchSmall.expand_as(imBatchSmall)) / pixelNum / 3.0
    # trainLight.py:237
    totalErr = REN_W * renderErr + REC_W * reconstErr
    totalErr.backward()

    assert _scalar_ok(reconstErr.item(), float(z["ref32_reconstErr"]), ref64["reconstErr"]), (name, reconstErr.item(), float(z["ref32_reconstErr"]))
    assert _scalar_ok(renderErr.item(), float(z["ref32_renderErr"]), ref64["renderErr"]), (name, renderErr.item(), float(z["ref32_renderErr"]))
    sub, sub2 = (slice(None), slice(None), slice(None, None, s), slice(None, None, s)), \
        (slice(None), slice(None), slice(None, None, 2 * s), slice(None, None, 2 * s))
    _check(name, "renderedImPred", renderedImPred, z["ref32_rendered"], ref64["rendered"])
    _check(name, "diffusePred", diffusePred, z["ref32_diffuse"], ref64["diffuse"])
    _check(name, "specularPred", specularPred, z["ref32_spec"], ref64["spec"])
    _check(name, "envmapsPred", envmapsPred, z["ref32_envmapsPred"], ref64["envmapsPred"], sub)
    _check(name, "envmapsPredScaledImage", envmapsPredScaledImage, z["ref32_envScaled"], ref64["envScaled"], sub2)
    tot = envmapsPredScaledImage.double().sum().item(), (envmapsPredScaledImage.double() ** 2).sum().item()
    assert abs(tot[0] - z["ref32_envScaled_sum"][0]) < 1e-4 * abs(z["ref32_envScaled_sum"][0])
    assert abs(tot[1] - z["ref32_envScaled_sum"][1]) < 2e-4 * abs(z["ref32_envScaled_sum"][1])
    for k, x in (("gx_axis", xa), ("gx_lamb", xl), ("gx_weight", xw)):
        _check(name, k, x.grad, z["ref32_" + k], ref64[k], sub)
    norms = [x.grad.double().norm().item() for x in (xa, xl, xw)]
    assert np.allclose(norms, z["ref32_gx_norms"], rtol=2e-4), (norms, z["ref32_gx_norms"])


@pytest.mark.parametrize("prologue", [False, True], ids=["heads_pass", "heads_prologue"])
@pytest.mark.parametrize("name", CASES)
def test_wrapper_sequence_fused_objective(sgr, name, prologue):
    """Routes (B) and (C): the same step through sgr.light_heads -> sgr.light_objective (two heavy kernels, no env image), and
    with the decoder heads inside those kernels (``decoder_outputs=True``: the activated SG parameters never exist)."""
    z, cfg, t = _load(name)
    R, C, K, eh, ew, s = cfg["R"], cfg["C"], cfg["K"], cfg["eh"], cfg["ew"], cfg["s"]
    ref64 = _oracle64(t, cfg)
    g = {k: v.cuda() for k, v in t.items()}
    xa, xl, xw = (g[k].clone().requires_grad_(True) for k in ("x_axis", "x_lamb", "x_weight"))
    renderLayer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    if prologue:
        obj, renderErr, reconstErr, renderedImPred, envScale = sgr.light_objective(
            renderLayer, g["albedoPred"], g["normalPred"], g["roughPred"], xa, xl, xw, g["im"], g["segBRDF"],
            g["envmaps"], g["envmapsInd"], REN_W, REC_W, OFFSET, decoder_outputs=True)
    else:
        axisPred, lambPred, weightPred, _ = sgr.light_heads(xa, xl, xw)
        obj, renderErr, reconstErr, renderedImPred, envScale = sgr.light_objective(
            renderLayer, g["albedoPred"], g["normalPred"], g["roughPred"], axisPred, lambPred, weightPred, g["im"], g["segBRDF"],
            g["envmaps"], g["envmapsInd"], REN_W, REC_W, OFFSET)
    obj.backward()
    assert _scalar_ok(reconstErr.item(), float(z["ref32_reconstErr"]), ref64["reconstErr"]), (name, reconstErr.item(), float(z["ref32_reconstErr"]))
    assert _scalar_ok(renderErr.item(), float(z["ref32_renderErr"]), ref64["renderErr"]), (name, renderErr.item(), float(z["ref32_renderErr"]))
    assert _scalar_ok(obj.item(), float(z["ref32_total"]), REN_W * ref64["renderErr"] + REC_W * ref64["reconstErr"])
    _check(name, "renderedImPred", renderedImPred, z["ref32_rendered"], ref64["rendered"])
    sub = (slice(None), slice(None), slice(None, None, s), slice(None, None, s))
    for k, x in (("gx_axis", xa), ("gx_lamb", xl), ("gx_weight", xw)):
        _check(name, k, x.grad, z["ref32_" + k], ref64[k], sub)
