"""Generate tests/golden/*.npz from the UNMODIFIED reference.  TEST INFRASTRUCTURE ONLY.

Run in the authoring container (needs /root/reference):

    python oracle/make_golden.py

The reference has no golden vectors of its own for this path,
so these fixtures -- outputs of ``models.output2env`` / ``models.renderingLayer`` /
``models.LSregressDiffSpec`` / ``models.LSregress`` and of the loss arithmetic of
``wrapperBRDFLight.py:170-207`` evaluated by the reference code itself, in fp32
and in fp64 -- are what pins both the oracle and the HIP kernels on machines
where the reference is not mounted (the GPU box).

Each fixture holds: the inputs, the reference's fp32 outputs (``ref32_*``), the
reference run in fp64 (``ref64_*``), fixed cotangents (``ct_*``) and the
gradients of ``sum(env*ct_env) + sum(diffuse*ct_d) + sum(spec*ct_s)`` with
respect to all six inputs (``g32_*`` / ``g64_*``), plus the loss scalars and the
gradients of ``renderErr + 10*reconstErr`` (``trainLight.py:47-48,237``).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_import as RI           # noqa: E402
from oracle import sg_oracle as O             # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: bn, imH, imW, R, C, K, eh, ew, fov, F0, seed, flavour
    "g1_q4_k12":   dict(bn=2, imH=16, imW=24, R=8, C=12, K=12, eh=8, ew=16, fov=57.0, F0=0.05, seed=20201, flavour="stress"),
    "g2_q1_k5":    dict(bn=1, imH=10, imW=12, R=10, C=12, K=5, eh=4, ew=8, fov=42.75, F0=0.05, seed=20202, flavour="benign"),
    "g3_edges":    dict(bn=1, imH=16, imW=24, R=8, C=12, K=12, eh=8, ew=16, fov=57.0, F0=0.05, seed=20203, flavour="edges"),
}


def make_inputs(cfg):
    inp = O.synthetic_inputs(cfg["bn"], cfg["imH"], cfg["imW"], cfg["R"], cfg["C"], cfg["K"],
                             cfg["eh"], cfg["ew"], seed=cfg["seed"], benign=(cfg["flavour"] == "benign"))
    if cfg["flavour"] == "edges":
        # exercise every clamp / eps branch of models.py:465-509
        n = inp["normal"]
        n[0, :, 0:2, 0:2] = torch.tensor([0.0, 1.0, 0.0]).view(3, 1, 1)      # N == up  -> camy = 0
        n[0, :, 0:2, 2:4] = torch.tensor([0.0, -1.0, 0.0]).view(3, 1, 1)     # N == -up
        n[0, :, 2, 0] = torch.tensor([0.6, 0.0, 0.8]); n[0, :, 2, 1] = torch.tensor([-0.6, 0.0, -0.8])
        n[0, :, 3, 0] = torch.tensor([0.6, 0.0, 0.8]); n[0, :, 3, 1] = torch.tensor([-0.6, 0.0, -0.8])   # pooled |N|^2 < 1e-6
        n[0, :, 4:6, 0:2] *= 1.7                                             # pooled |N|^2 > 1 (upper clamp)
        n[0, :, 6:8, 4:8] = torch.tensor([0.0, 0.0, -1.0]).view(3, 1, 1)     # facing away: ndv clamps to 0
        inp["rough"][0, 0, :, 12:] = -0.97                                   # tiny alpha: nom lower clamp live
        inp["rough"][0, 0, :, :4] = 1.0
        inp["albedo"][0, :, 8:, :] *= 2.5                                    # mean-normalised albedo can exceed 1
        inp["lamb"][0, 0] = 1.0                                              # lambda = tan(pi/2*0.999) = 636.6
        inp["lamb"][0, 1] = 0.0                                              # lambda = 0
        inp["weight"][0, 0:3] = 1.0
        inp["weight"][0, 3:6] = 0.0
        inp["im"][0, :, :, :6] = 0.95                                        # masked out by im<0.9
        inp["seg"][0, :, :4, :] = 0.0
    return inp


def run_reference(cfg, inp, dtype):
    K, R, C = cfg["K"], cfg["R"], cfg["C"]
    M = RI.models()
    o2e, rl = RI.make_layers(K, R, C, cfg["eh"], cfg["ew"], cfg["fov"], cfg["F0"], dtype=dtype)
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight")
    x = {k: inp[k].to(dtype).clone().requires_grad_(True) for k in names}
    env, _, lam_t, w_t = o2e.output2env(x["axis"], x["lamb"], x["weight"])
    d, s = rl.forwardEnv(x["albedo"], x["normal"], x["rough"], env)
    g = torch.Generator().manual_seed(cfg["seed"] + 7)
    ct_env = torch.randn(env.shape, generator=g).to(dtype)
    ct_d = torch.randn(d.shape, generator=g).to(dtype)
    ct_s = torch.randn(s.shape, generator=g).to(dtype)
    lin = (env * ct_env).sum() + (d * ct_d).sum() + (s * ct_s).sum()
    g_lin = torch.autograd.grad(lin, [x[k] for k in names], retain_graph=True)

    # loss arithmetic of wrapperBRDFLight.py:170-207 evaluated with the reference functions
    im, seg, env_gt = inp["im"].to(dtype), inp["seg"].to(dtype), inp["env_gt"].to(dtype)
    env_ind = torch.ones(cfg["bn"], 1, 1, 1, dtype=dtype)
    im_s = F.adaptive_avg_pool2d(im, (R, C))
    seg_s = F.adaptive_avg_pool2d(seg, (R, C))
    not_dark = (torch.mean(torch.mean(torch.mean(env_gt, 4), 4), 1, True) > 0.001).to(dtype)
    seg_env = (seg_s * env_ind.expand_as(seg_s)).unsqueeze(-1).unsqueeze(-1) * not_dark.unsqueeze(-1).unsqueeze(-1)
    pn_env = max(seg_env.sum().item(), 1e-5)
    env_sc = M.LSregress(env.detach() * seg_env.expand_as(env_gt), env_gt * seg_env.expand_as(env_gt), env)
    dl = torch.log(env_sc + 1.0) - torch.log(env_gt + 1.0)
    recon = torch.sum(dl * dl * seg_env.expand_as(env)) / pn_env / 3.0 / cfg["ew"] / cfg["eh"]
    pn = max(seg_s.sum().item(), 1e-5)
    d_sc, s_sc = M.LSregressDiffSpec(d.detach(), s.detach(), im_s, d, s)
    ren = torch.clamp(d_sc + s_sc, 0, 1)
    rerr = torch.sum((ren - im_s) * (ren - im_s) * seg_s.expand_as(im_s)) / pn / 3.0
    total = 1.0 * rerr + 10.0 * recon
    g_tot = torch.autograd.grad(total, [x[k] for k in names], allow_unused=True)
    g_tot = [torch.zeros_like(x[k]) if gi is None else gi for gi, k in zip(g_tot, names)]

    out = dict(env=env, lamb_tan=lam_t, weight_tan=w_t, diffuse=d, spec=s, diff_scaled=d_sc, spec_scaled=s_sc,
               rendered=ren, env_scaled=env_sc, render_err=rerr.reshape(1), recon_err=recon.reshape(1),
               render_num=(rerr * pn * 3.0).reshape(1), render_den=torch.tensor([pn], dtype=dtype))
    out.update({f"glin_{k}": gi for k, gi in zip(names, g_lin)})
    out.update({f"gtot_{k}": gi for k, gi in zip(names, g_tot)})
    cts = dict(ct_env=ct_env, ct_d=ct_d, ct_s=ct_s)
    return out, cts


def main():
    if not RI.available():
        raise SystemExit("reference not mounted; fixtures can only be generated in the authoring container")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for name, cfg in CASES.items():
        inp = make_inputs(cfg)
        r32, cts = run_reference(cfg, inp, torch.float32)
        r64, _ = run_reference(cfg, inp, torch.float64)
        blob = {f"in_{k}": v.numpy() for k, v in inp.items()}
        blob.update({k: v.detach().numpy().astype(np.float32) for k, v in cts.items()})
        blob.update({f"ref32_{k}": v.detach().numpy() for k, v in r32.items()})
        # fp64 reference kept as float64 only for small tensors; env-sized ones are stored rounded to float32
        blob.update({f"ref64_{k}": (v.detach().numpy() if v.numel() < 50000 else v.detach().numpy().astype(np.float32))
                     for k, v in r64.items()})
        blob["cfg_keys"] = np.array(sorted(k for k in cfg if k != "flavour"))
        blob["cfg_vals"] = np.array([float(cfg[k]) for k in sorted(k for k in cfg if k != "flavour")])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  "
              f"renderErr32={r32['render_err'].item():.6g} reconErr32={r32['recon_err'].item():.6g}")


if __name__ == "__main__":
    main()
