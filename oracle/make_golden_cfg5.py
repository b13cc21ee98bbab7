"""A small reference-made fixture at BASELINE config 5's parameters (SGNum 24, 16x32 directions, 2x2-pooled BRDF maps) through the
UNMODIFIED reference, fp32 and fp64, forward and backward w.r.t. all six inputs.  TEST INFRASTRUCTURE ONLY (authoring container):

    python -m oracle.make_golden_cfg5           # writes tests/golden/g8_cfg5_small.npz

12 x 16 env cells (24 x 32 BRDF maps): the reference's own fp32-vs-fp64 error at K 24 / 16x32 -- the yardstick
``max(2 e_ref, 1e-4)`` of the config-5 tolerances in tests/test_gpu_fullsize.py -- plus values and gradients to compare
against directly (same layout as g1..g3, oracle/make_golden.py)."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import make_golden as MG
from oracle import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CFG = dict(bn=1, imH=24, imW=32, R=12, C=16, K=24, eh=16, ew=32, fov=57.0, F0=0.05, seed=20208, flavour="stress", rng="numpy")


def main():
    if not RI.available():
        raise SystemExit("reference not mounted")
    torch.set_num_threads(8)
    inp = MG.make_inputs(CFG)
    r32, cts = MG.run_reference(CFG, inp, torch.float32)
    r64, _ = MG.run_reference(CFG, inp, torch.float64)
    # inputs and cotangents are regenerated from the seed by the tests (checksums stored, like g7); the env image is kept
    # at every second cell
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight")
    blob = dict(in_checksums=np.array([inp[k].double().sum().item() for k in names]),
                ct_checksums=np.array([cts[k].double().sum().item() for k in ("ct_env", "ct_d", "ct_s")]), env_stride=np.array([2]))
    keep = ("env", "diffuse", "spec") + tuple(f"glin_{k}" for k in names)
    for tag, r in (("ref32", r32), ("ref64", r64)):
        for k in keep:
            v = r[k].detach()
            if k == "env":
                v = v[:, :, ::2, ::2]
            v = v.numpy()
            blob[f"{tag}_{k}"] = v if (tag == "ref64" and v.size < 50000) else v.astype(np.float32)
    blob["cfg_keys"] = np.array(sorted(k for k in CFG if k not in ("flavour", "rng")))
    blob["cfg_vals"] = np.array([float(CFG[k]) for k in sorted(k for k in CFG if k not in ("flavour", "rng"))])
    path = os.path.join(OUT, "g8_cfg5_small.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")
    for k in ("env", "diffuse", "spec") + tuple(f"glin_{n}" for n in ("axis", "lamb", "weight", "albedo", "normal", "rough")):
        a, b = torch.from_numpy(np.asarray(blob["ref32_" + k])).double(), torch.from_numpy(np.asarray(blob["ref64_" + k])).double()
        print(f"  reference fp32 vs fp64 {k:12s} rel-L2 {((a - b).norm() / b.norm()).item():.2e}")


if __name__ == "__main__":
    main()
