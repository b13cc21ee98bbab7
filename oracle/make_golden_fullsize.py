"""One image of BASELINE config 2 (240x320 BRDF maps -> 120x160 env grid, SGNum 12, 8x16 directions) through the
UNMODIFIED reference, fp32 and fp64, forward and backward.  TEST INFRASTRUCTURE ONLY (authoring container):

    python -m oracle.make_golden_fullsize       # writes tests/golden/g7_cfg2_one_image.npz

    python -m oracle.make_golden_fullsize       # writes tests/golden/g7_cfg2_one_image.npz and g9_ratio1_unit_normals.npz

The inputs and cotangents are the seeded synthetic ones of SURVEY.md section 8d drawn from NumPy's frozen legacy stream
(``oracle.sg_oracle.synthetic_inputs_np`` -- round 5: torch's CPU generator is an implementation detail of the installed torch, and
rounds 3-4 had to SKIP these comparisons when its stream differed; now the tests regenerate the inputs from the seed and FAIL on a
checksum mismatch), so only results are stored -- sub-sampled over
the pixel grid where they are env- or SG-sized (stride recorded), with full-tensor norms.  This is what lets the
full-size GPU tests bound the error against the fp64 oracle by the REFERENCE's own fp32 error instead of a bare constant."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import make_golden as MG
from oracle import ref_import as RI
from oracle import sg_oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CFG = dict(bn=1, imH=240, imW=320, R=120, C=160, K=12, eh=8, ew=16, fov=57.0, F0=0.05, seed=20207, flavour="stress", rng="numpy")
S_ENV, S_SG = 6, 3


def checksums(inp):
    return np.array([inp[k].double().sum().item() for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")])


# g9 (round 5): the |N|^2 == 1 clamp kink at size.  BRDF maps AT the env-grid resolution (ratio 1: no pooling, trainFineTune*_cascade1.py /
# testReal.py call the layer that way) with unit input normals: clamp(sum N^2, 1e-6, 1) (models.py:467-468) passes the gradient on one side
# of 1 and blocks it on the other, and a unit fp32 vector squares to 1 - eps, 1 or 1 + eps -- so the normal / roughness gradients of the
# reference's fp32 and fp64 runs differ by O(1) pixel by pixel, fp64 is no arbiter there, and the reference's fp32 VALUES are the
# semantics to match.  Stored: values and all six gradients (fp32 and fp64 runs), plus `agree`: the pixels where the fp32 and the fp64
# evaluation of the clamp take the same branch -- there fp64 IS an arbiter, and the reference's own fp32 error e_ref is measured there.
CFG9 = dict(bn=1, imH=120, imW=160, R=120, C=160, K=12, eh=8, ew=16, fov=57.0, F0=0.05, seed=20209, flavour="stress", rng="numpy")


def clamp_branch_agreement(normal):
    """[1,1,H,W] bool: torch's fp32 sum of squares of the (unpooled) normal and the fp64 one land on the same side of the two-sided
    clamp's upper kink (<= 1: gradient passes; > 1: blocked)."""
    n32 = normal.float()
    nn32 = torch.sum(n32 * n32, dim=1, keepdim=True)                 # models.py:467: torch.sum(normal * normal, dim=1)
    n64 = normal.double()
    nn64 = torch.sum(n64 * n64, dim=1, keepdim=True)
    return (nn32 <= 1.0) == (nn64 <= 1.0)


def kink_fixture():
    inp = MG.make_inputs(CFG9)
    r32, cts = MG.run_reference(CFG9, inp, torch.float32)
    r64, _ = MG.run_reference(CFG9, inp, torch.float64)
    agree = clamp_branch_agreement(inp["normal"])
    keys = sorted(k for k in CFG9 if k not in ("flavour", "rng"))
    blob = dict(cfg_keys=np.array(keys), cfg_vals=np.array([float(CFG9[k]) for k in keys]), strides=np.array([S_ENV, S_SG]), in_checksums=checksums(inp),
                ct_checksums=np.array([cts[k].double().sum().item() for k in ("ct_env", "ct_d", "ct_s")]), agree=agree.numpy())
    for tag, r in (("ref32", r32), ("ref64", r64)):
        blob[f"{tag}_env"] = r["env"].detach()[:, :, ::S_ENV, ::S_ENV].numpy().astype(np.float32)
        blob[f"{tag}_diffuse"] = r["diffuse"].detach().numpy().astype(np.float32)
        blob[f"{tag}_spec"] = r["spec"].detach().numpy().astype(np.float32)
        for k in ("axis", "lamb", "weight"):
            g = r[f"glin_{k}"].detach()
            blob[f"{tag}_glin_{k}"] = g[..., ::S_SG, ::S_SG].numpy().astype(np.float32)
        for k in ("albedo", "normal", "rough"):                     # image-sized (ratio 1): kept whole
            blob[f"{tag}_glin_{k}"] = r[f"glin_{k}"].detach().numpy().astype(np.float32)
    path = os.path.join(OUT, "g9_ratio1_unit_normals.npz")
    np.savez_compressed(path, **blob)
    frac = agree.float().mean().item()
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB; clamp branches agree on {100 * frac:.1f} % of the pixels")
    for k in ("normal", "rough", "albedo"):
        a, b = torch.from_numpy(blob[f"ref32_glin_{k}"]).double(), torch.from_numpy(blob[f"ref64_glin_{k}"]).double()
        m = agree.expand_as(a)
        print(f"  reference fp32 vs fp64 glin_{k:7s} rel-L2 all pixels {((a - b).norm() / b.norm()).item():.2e}   where the branches agree {((a[m] - b[m]).norm() / b[m].norm()).item():.2e}")


def main():
    if not RI.available():
        raise SystemExit("reference not mounted")
    torch.set_num_threads(8)
    kink_fixture()
    inp = MG.make_inputs(CFG)
    r32, cts = MG.run_reference(CFG, inp, torch.float32)
    r64, _ = MG.run_reference(CFG, inp, torch.float64)
    blob = dict(cfg_keys=np.array(sorted(k for k in CFG if k not in ("flavour", "rng"))),
                cfg_vals=np.array([float(CFG[k]) for k in sorted(k for k in CFG if k not in ("flavour", "rng"))]),
                strides=np.array([S_ENV, S_SG]), in_checksums=checksums(inp),
                ct_checksums=np.array([cts[k].double().sum().item() for k in ("ct_env", "ct_d", "ct_s")]))
    for tag, r in (("ref32", r32), ("ref64", r64)):
        blob[f"{tag}_env"] = r["env"].detach()[:, :, ::S_ENV, ::S_ENV].numpy().astype(np.float32)
        blob[f"{tag}_diffuse"] = r["diffuse"].detach().numpy().astype(np.float32)
        blob[f"{tag}_spec"] = r["spec"].detach().numpy().astype(np.float32)
        for k in ("axis", "lamb", "weight", "albedo", "normal", "rough"):      # the BRDF-map gradients too (round 3): a10 at size
            g = r[f"glin_{k}"].detach()
            blob[f"{tag}_glin_{k}"] = g[..., ::S_SG, ::S_SG].numpy().astype(np.float32)
            blob[f"{tag}_glin_{k}_norm"] = np.array([g.double().norm().item()])
        blob[f"{tag}_env_norm"] = np.array([r["env"].detach().double().norm().item()])
    path = os.path.join(OUT, "g7_cfg2_one_image.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")
    for k in ("env", "diffuse", "spec", "glin_axis", "glin_lamb", "glin_weight", "glin_albedo", "glin_normal", "glin_rough"):
        a, b = torch.from_numpy(blob["ref32_" + k]).double(), torch.from_numpy(blob["ref64_" + k]).double()
        print(f"  reference fp32 vs fp64 {k:12s} rel-L2 {((a - b).norm() / b.norm()).item():.2e}")


if __name__ == "__main__":
    main()
