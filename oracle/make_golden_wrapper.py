"""Wrapper-level golden vectors: the UNMODIFIED reference ``wrapperBRDFLight.wrapperBRDFLight`` (cascade 0) run on CPU
with seeded random reference networks, with hooks capturing exactly what crosses the hot-path boundary.
TEST INFRASTRUCTURE ONLY -- authoring container (needs /root/reference):

    python -m oracle.make_golden_wrapper          # writes tests/golden/g5_wrapper_small.npz, g5_wrapper_120x160.npz

What is captured (wrapperBRDFLight.py line numbers):
  inputs of the path   the three light decoders' ``dconvFinal`` outputs (pre-activation, models.py:334-336; made leaves
                       for the gradient), ``albedoPred`` (mean-normalised, :138-142) / ``normalPred`` / ``roughPred``
                       (what ``forwardEnv`` receives, :194), ``imBatch``, ``segBRDFBatch``, ``envmapsBatch``,
                       ``envmapsIndBatch``;
  results              ``envmapsPred`` (packed 7K channels, :167-168), ``envmapsPredScaledImage`` and ``reconstErr``
                       (:179-188), ``diffusePred`` / ``specularPred`` (:194), ``renderedImPred`` / ``renderErr``
                       (:203-207), and d(renW * renderErr + recW * reconstErr)/d(dconvFinal outputs) with
                       renW = 1, recW = 10 (trainLight.py:47-48,237);
  the glue in front    the light encoder's input (``inputBatch`` [bn,11,480,640], :138-156: mean-normalise, bilinear
                       up-sampling, concatenation) and the un-normalised ``albedoPred`` / ``depthPred`` that produce it
                       (sub-sampled: the tensor is 13.5 MB per image).

Large tensors of the 120x160 case are stored sub-sampled (strides recorded in the file) together with full-tensor
sums; the ground-truth env map of that case is block-constant over 8x8 cells so that it stores in 0.5 MB.
"""
from __future__ import annotations

import argparse
import os
import types

import numpy as np
import torch

from oracle import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
K, EH, EW = 12, 8, 16
REN_W, REC_W = 1.0, 10.0

CASES = {
    # name: bn, imH, imW, envRow, envCol, seed, gt block (cells per constant block of the GT env map), output strides
    "g5_wrapper_small": dict(bn=2, imH=32, imW=48, R=16, C=24, seed=31, block=1, stride=1),
    "g5_wrapper_120x160": dict(bn=1, imH=240, imW=320, R=120, C=160, seed=32, block=8, stride=3),
}


def build_nets(M, seed):
    torch.manual_seed(seed)
    nets = dict(encoder=M.encoder0(cascadeLevel=0), albedoDecoder=M.decoder0(mode=0), normalDecoder=M.decoder0(mode=1),
                roughDecoder=M.decoder0(mode=2), depthDecoder=M.decoder0(mode=4),
                lightEncoder=M.encoderLight(cascadeLevel=0, SGNum=K), axisDecoder=M.decoderLight(mode=0, SGNum=K),
                lambDecoder=M.decoderLight(mode=1, SGNum=K), weightDecoder=M.decoderLight(mode=2, SGNum=K))
    with torch.no_grad():      # spread the light heads' pre-activations over the interesting range of tanh / clamp
        for n in ("axisDecoder", "lambDecoder", "weightDecoder"):
            nets[n].dconvFinal.weight.mul_(4.0)
    return nets


def synthetic_batch(cfg):
    """A dataBatch with the keys / shapes of dataLoader.py:188-213 (cascade 0)."""
    g = torch.Generator().manual_seed(cfg["seed"] + 1000)
    bn, h, w, R, C = cfg["bn"], cfg["imH"], cfg["imW"], cfg["R"], cfg["C"]
    nrm = torch.randn(bn, 3, h, w, generator=g)
    nrm[:, 2] = nrm[:, 2].abs() + 0.5
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    blk = cfg["block"]
    gt = torch.rand(bn, 3, R // blk, C // blk, EH, EW, generator=g) * 2.0
    gt[:, :, 0, 0] = 0.0                                            # a dark block: notDarkEnv == 0 there (:172)
    env = gt.repeat_interleave(blk, dim=2).repeat_interleave(blk, dim=3).contiguous()
    ind = torch.ones(bn, 1, 1, 1)
    if bn > 1:
        ind[-1] = 0.0                                               # an image without ground-truth env maps (envmapsInd, :173)
    seg_obj = (torch.rand(bn, 1, h, w, generator=g) < 0.9).float()
    seg_area = ((torch.rand(bn, 1, h, w, generator=g) < 0.05).float() * (1 - seg_obj))
    return dict(albedo=torch.rand(bn, 3, h, w, generator=g), normal=nrm, rough=torch.rand(bn, 1, h, w, generator=g) * 2 - 1,
                depth=torch.rand(bn, 1, h, w, generator=g) * 3 + 1, segArea=seg_area, segEnv=torch.zeros(bn, 1, h, w), segObj=seg_obj,
                im=torch.rand(bn, 3, h, w, generator=g), envmaps=env, envmapsInd=ind), gt


def run_case(name, cfg):
    M = RI.models()
    W = RI.wrapper_brdf_light()
    nets = build_nets(M, cfg["seed"])
    batch, gt_blocks = synthetic_batch(cfg)
    opt = types.SimpleNamespace(cascadeLevel=0, imHeight=cfg["imH"], imWidth=cfg["imW"], envRow=cfg["R"], envCol=cfg["C"],
                                envHeight=EH, envWidth=EW, SGNum=K)
    o2e = M.output2env(isCuda=False, envWidth=EW, envHeight=EH, SGNum=K)
    rl = M.renderingLayer(isCuda=False, imWidth=cfg["C"], imHeight=cfg["R"], envWidth=EW, envHeight=EH)

    cap = {}

    def head_hook(tag):
        def hook(_m, _i, out):
            out.retain_grad()
            cap.setdefault(tag, []).append(out)
        return hook
    hooks = [nets[n].dconvFinal.register_forward_hook(head_hook(t)) for n, t in
             (("axisDecoder", "axis"), ("lambDecoder", "lamb"), ("weightDecoder", "weight"))]
    hooks.append(nets["lightEncoder"].register_forward_pre_hook(lambda _m, args: cap.__setitem__("light_in", args[0].detach().clone())))
    for t, n in (("albedo_raw", "albedoDecoder"), ("depth_raw", "depthDecoder")):
        hooks.append(nets[n].register_forward_hook(lambda _m, _i, out, t=t: cap.__setitem__(t, out.detach().clone())))

    res = W.wrapperBRDFLight(batch, opt, nets["encoder"], nets["albedoDecoder"], nets["normalDecoder"], nets["roughDecoder"],
                             nets["depthDecoder"], nets["lightEncoder"], nets["axisDecoder"], nets["lambDecoder"],
                             nets["weightDecoder"], o2e, rl, offset=1.0, isLightOut=True)
    for h in hooks:
        h.remove()
    albedoPair, normalPair, roughPair, depthPair, envPair, renderPair, lightOut = res
    envScaled, reconstErr = envPair[0], envPair[1]
    rendered, renderErr = renderPair[0], renderPair[1]
    envmapsPred, diffusePred, specularPred = lightOut
    total = REN_W * renderErr + REC_W * reconstErr
    total.backward()
    x = {t: cap[t][-1] for t in ("axis", "lamb", "weight")}     # forward evaluates dconvFinal twice; the second result is the one used
    s = cfg["stride"]
    f = lambda t: t.detach().numpy().astype(np.float32)
    blob = dict(
        meta=np.array([cfg["bn"], cfg["imH"], cfg["imW"], cfg["R"], cfg["C"], K, EH, EW, cfg["block"], s], dtype=np.int64),
        x_axis=f(x["axis"]), x_lamb=f(x["lamb"]), x_weight=f(x["weight"]),
        albedoPred=f(albedoPair[0]), normalPred=f(normalPair[0]), roughPred=f(roughPair[0]),
        im=f(batch["im"]), segBRDF=f(batch["segObj"]), envmaps_blocks=f(gt_blocks), envmapsInd=f(batch["envmapsInd"]),
        ref32_reconstErr=f(reconstErr), ref32_renderErr=f(renderErr), ref32_total=f(total),
        ref32_rendered=f(rendered), ref32_diffuse=f(diffusePred), ref32_spec=f(specularPred),
        ref32_envmapsPred=f(envmapsPred[:, :, ::s, ::s]),
        ref32_envScaled=f(envScaled[:, :, ::2 * s, ::2 * s]),        # [bn,3,R/2s,C/2s,eh,ew]
        ref32_envScaled_sum=np.array([envScaled.double().sum().item(), (envScaled.double() ** 2).sum().item()]),
        ref32_gx_axis=f(x["axis"].grad[:, :, ::s, ::s]), ref32_gx_lamb=f(x["lamb"].grad[:, :, ::s, ::s]),
        ref32_gx_weight=f(x["weight"].grad[:, :, ::s, ::s]),
        ref32_gx_norms=np.array([x[t].grad.double().norm().item() for t in ("axis", "lamb", "weight")]),
        # the glue in front of the path (wrapperBRDFLight.py:138-156)
        albedo_raw=f(0.5 * (cap["albedo_raw"] + 1)), depth_raw=f(0.5 * (cap["depth_raw"] + 1)),
        light_in=f(cap["light_in"][:, :, ::8, ::8]),
        light_in_sum=np.stack([cap["light_in"].double().sum(dim=(0, 2, 3)).numpy(), (cap["light_in"].double() ** 2).sum(dim=(0, 2, 3)).numpy()]),
    )
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB", "reconstErr", reconstErr.item(), "renderErr", renderErr.item())


def main():
    if not RI.available():
        raise SystemExit("reference not mounted; fixtures can only be generated in the authoring container")
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    for name, cfg in CASES.items():
        if args.only and args.only != name:
            continue
        run_case(name, cfg)


if __name__ == "__main__":
    main()
