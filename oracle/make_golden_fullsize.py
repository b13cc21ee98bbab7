"""One image of BASELINE config 2 (240x320 BRDF maps -> 120x160 env grid, SGNum 12, 8x16 directions) through the
UNMODIFIED reference, fp32 and fp64, forward and backward.  TEST INFRASTRUCTURE ONLY (authoring container):

    python -m oracle.make_golden_fullsize       # writes tests/golden/g7_cfg2_one_image.npz

The inputs and cotangents are the seeded synthetic ones of SURVEY.md section 8d (``oracle.sg_oracle.synthetic_inputs``;
the tests regenerate them from the seed and check the stored checksums), so only results are stored -- sub-sampled over
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
CFG = dict(bn=1, imH=240, imW=320, R=120, C=160, K=12, eh=8, ew=16, fov=57.0, F0=0.05, seed=20207, flavour="stress")
S_ENV, S_SG = 6, 3


def checksums(inp):
    return np.array([inp[k].double().sum().item() for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")])


def main():
    if not RI.available():
        raise SystemExit("reference not mounted")
    torch.set_num_threads(8)
    inp = MG.make_inputs(CFG)
    r32, cts = MG.run_reference(CFG, inp, torch.float32)
    r64, _ = MG.run_reference(CFG, inp, torch.float64)
    blob = dict(cfg_keys=np.array(sorted(k for k in CFG if k != "flavour")),
                cfg_vals=np.array([float(CFG[k]) for k in sorted(k for k in CFG if k != "flavour")]),
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
