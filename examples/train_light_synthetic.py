#!/usr/bin/env python
"""Synthetic cascade-0 lighting-training loop (BASELINE config 3) on the MI355X render layer.

The reference's trainLight.py cannot be imported without its dataset / cv2 / h5py, so this rebuilds
its step (trainLight.py:203-244 -> wrapperBRDFLight.py:158-207) around synthetic tensors:

  * "frozen BRDF net outputs": albedo / normal / rough maps (no grad, like trainLight.py:121-144);
  * the light network is replaced by three learnable tensors pushed through decoderLight's output
    activations (models.py:336-346): 1.01*tanh -> unit axes, 0.5*(x+1) clamped to [0,1] for lamb/weight
    -- by default inside the objective's two heavy kernels (light_objective(decoder_outputs=True));
    with --unfused as sgr.light_heads, one HIP pass each way; --torch-heads for the op-by-op torch version;
  * step = zero_grad -> objective -> backward -> Adam, the objective being either
      fused (default): sgr.light_objective -- render loss + 10 x log-L2 env reconstruction loss in two heavy
                       kernel passes, the predicted env image never written, or
      --unfused:       forwardSG (env image + diffuse + specular) -> sgr.render_loss + sgr.recon_loss.

    python examples/train_light_synthetic.py --batch 16 --steps 20 [--unfused]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inverserenderingofindoorscene_amd as sgr  # noqa: E402


def decoder_heads(t_axis, t_lamb, t_weight):
    """Output activations of decoderLight (models.py:336-346) in torch ops (--torch-heads; the default path is the
    fused sgr.light_heads kernel pair)."""
    a = 1.01 * torch.tanh(t_axis)
    a = a / torch.clamp(torch.sqrt((a * a).sum(2, keepdim=True)), min=1e-6)
    lam = torch.clamp(0.5 * (1.01 * torch.tanh(t_lamb) + 1), 0, 1)
    w = torch.clamp(0.5 * (1.01 * torch.tanh(t_weight) + 1), 0, 1)
    return a, lam, w


def recon_loss_torch(env_pred, env_gt, seg, env_ind, R, C, offset=1.0):
    """wrapperBRDFLight.py:171-188 with torch ops + sgr.LSregress (kept for comparison with sgr.recon_loss)."""
    eh, ew = env_pred.shape[4], env_pred.shape[5]
    seg_s = F.adaptive_avg_pool2d(seg, (R, C))
    not_dark = (env_gt.mean(5).mean(4).mean(1, keepdim=True) > 0.001).float()
    m = ((seg_s * env_ind.expand_as(seg_s))[..., None, None] * not_dark[..., None, None]).expand_as(env_gt)
    scaled = sgr.LSregress(env_pred.detach() * m, env_gt * m, env_pred)
    dlog = torch.log(scaled + offset) - torch.log(env_gt + offset)
    den = torch.clamp(m[:, :1, :, :, :1, :1].sum(), min=1e-5)
    return (dlog * dlog * m).sum() / den / 3.0 / ew / eh


def make_batch(bn, imH, imW, R, C, eh, ew, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = torch.randn(bn, 3, imH, imW, generator=g)
    n[:, 2] = n[:, 2].abs() + 0.5
    return dict(albedo=torch.rand(bn, 3, imH, imW, generator=g).to(dev),
                normal=(n / n.norm(dim=1, keepdim=True)).to(dev),
                rough=(torch.rand(bn, 1, imH, imW, generator=g) * 2 - 1).to(dev),
                im=torch.rand(bn, 3, imH, imW, generator=g).to(dev),
                seg=(torch.rand(bn, 1, imH, imW, generator=g) < 0.9).float().to(dev),
                env_gt=(torch.rand(bn, 3, R, C, eh, ew, generator=g) * 2).to(dev),
                env_ind=torch.ones(bn, 1, 1, 1, device=dev))


def train(bn=16, steps=10, imH=240, imW=320, R=120, C=160, K=12, eh=8, ew=16, renW=1.0, recW=10.0, lr=1e-2,
          seed=0, verbose=True, fused=True, hip_heads=True):
    dev = torch.device("cuda")
    batch = make_batch(bn, imH, imW, R, C, eh, ew, dev, seed)
    g = torch.Generator().manual_seed(seed + 1)
    params = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(dev))
              for s in ((bn, K, 3, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999), fused=True)        # trainLight.py:178-181 (fused: one multi-tensor kernel)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    hist = []
    warmup = min(3, max(steps - 1, 0))
    t0 = time.perf_counter()
    for it in range(steps):
        if it == warmup:               # the first steps pay module load / allocator growth
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad()
        prologue = hip_heads and fused and sgr.light_objective_supported(K, R, C, eh, ew)
        if prologue:                   # default: the decoders' output activations run inside the objective's kernels
            total, render_err, recon_err, _, _ = sgr.light_objective(layer, batch["albedo"], batch["normal"], batch["rough"], params[0],
                                                                     params[1], params[2], batch["im"], batch["seg"], batch["env_gt"],
                                                                     batch["env_ind"], renW, recW, decoder_outputs=True)
        elif hip_heads:
            axis, lam, w, _ = sgr.light_heads(params[0].view(bn, 3 * K, R, C), params[1], params[2])
        else:
            axis, lam, w = decoder_heads(*params)
        if prologue:
            pass
        elif fused and sgr.light_objective_supported(K, R, C, eh, ew):
            total, render_err, recon_err, _, _ = sgr.light_objective(layer, batch["albedo"], batch["normal"], batch["rough"], axis, lam, w,
                                                                     batch["im"], batch["seg"], batch["env_gt"], batch["env_ind"],
                                                                     renW, recW)
        else:
            env, diffuse, spec = layer.forwardSG(batch["albedo"], batch["normal"], batch["rough"], axis, lam, w, need_env=True)
            render_err, _ = sgr.render_loss(diffuse, spec, batch["im"], batch["seg"], R, C)
            recon_err = sgr.recon_loss(env, batch["env_gt"], batch["seg"], batch["env_ind"], R, C)
            total = renW * render_err + recW * recon_err               # trainLight.py:237
        total.backward()
        opt.step()
        hist.append((total.detach(), render_err.detach(), recon_err.detach()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(steps - warmup, 1)
    hist = [(a.item(), b.item(), c.item()) for a, b, c in hist]
    if verbose:
        for i, (a, b, c) in enumerate(hist):
            print(f"step {i:3d}  total {a:.5f}  renderErr {b:.5f}  reconstErr {c:.5f}")
        print(f"{dt * 1e3:.2f} ms/step  ({bn * imH * imW / dt / 1e6:.1f} Mpix/s incl. torch glue, recon loss and Adam)")
    return hist, dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--unfused", action="store_true", help="materialise the env image and use the separate loss kernels")
    ap.add_argument("--torch-heads", action="store_true", help="decoder output activations as torch ops instead of sgr.light_heads")
    args = ap.parse_args()
    train(bn=args.batch, steps=args.steps, fused=not args.unfused, hip_heads=not args.torch_heads)
