"""Golden vectors for ``utils.predToShading`` (utils.py:156-195) produced by the UNMODIFIED reference function.
TEST INFRASTRUCTURE ONLY -- authoring container (needs /root/reference):

    python -m oracle.make_golden_shading        # writes tests/golden/g6_shading.npz

``utils.py`` imports cv2 / h5py / PIL at module level (none installed here) but ``predToShading`` itself needs only
numpy, so empty stand-in modules are registered for the import; the function body runs as written.  The reference
evaluates it in whatever dtype ``pred`` has: float32 (how testReal.py:638 calls it, on the network's output) and
float64 are both stored.  Also stored: the ``cLight / cAlbedo`` post-scale of testReal.py:421-432 evaluated with the
reference's expressions on the same tensors."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

from oracle import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_utils():
    for name in ("cv2", "h5py"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "PIL" not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            pil = types.ModuleType("PIL"); pil.Image = types.ModuleType("PIL.Image")
            sys.modules["PIL"] = pil; sys.modules["PIL.Image"] = pil.Image
    return RI._import("utils")


def main():
    if not RI.available():
        raise SystemExit("reference not mounted")
    U = reference_utils()
    blob = {}
    for tag, K, R, C, eh, ew, seed in (("a", 12, 9, 14, 16, 32, 21), ("b", 12, 6, 10, 8, 16, 22), ("c", 5, 5, 7, 16, 32, 23)):
        g = torch.Generator().manual_seed(seed)
        a = torch.randn(1, K, 3, R, C, generator=g)
        a = a / a.norm(dim=2, keepdim=True)
        pred = torch.cat([a.reshape(1, 3 * K, R, C), torch.rand(1, K, R, C, generator=g), torch.rand(1, 3 * K, R, C, generator=g)], 1).numpy()
        blob[f"{tag}_cfg"] = np.array([K, R, C, eh, ew])
        blob[f"{tag}_pred"] = pred
        blob[f"{tag}_ref32"] = U.predToShading(pred.copy(), envWidth=ew, envHeight=eh, SGNum=K).astype(np.float32)
        blob[f"{tag}_ref64"] = U.predToShading(pred.astype(np.float64), envWidth=ew, envHeight=eh, SGNum=K)
    # testReal.py:413-432 -- LSregressDiffSpec, then the global light / albedo scale; the reference's expressions verbatim
    M = RI.models()
    for tag, seed, spec_gain, alb_gain in (("s1", 40, 0.3, 1.0), ("s2", 41, 0.0, 1.0), ("s3", 42, 3.0, 0.2), ("s4", 40, 0.3, 5.0)):
        g = torch.Generator().manual_seed(seed)
        bn, R, C = 1, 12, 16
        diffusePred = torch.rand(bn, 3, R, C, generator=g)
        specularPred = torch.rand(bn, 3, R, C, generator=g) * spec_gain + (1e-6 if spec_gain == 0.0 else 0.0)
        imBatchSmall = torch.rand(bn, 3, R, C, generator=g)
        albedoPreds = [torch.rand(bn, 3, 2 * R, 2 * C, generator=g) * alb_gain]
        diffusePredNew, specularPredNew = M.LSregressDiffSpec(diffusePred, specularPred, imBatchSmall, diffusePred, specularPred)
        cDiff, cSpec = (torch.sum(diffusePredNew) / torch.sum(diffusePred)).data.item(), ((torch.sum(specularPredNew)) / (torch.sum(specularPred))).data.item()
        if cSpec < 1e-3:
            cAlbedo = 1 / albedoPreds[-1].max().data.item()
            cLight = cDiff / cAlbedo
        else:
            cLight = cSpec
            cAlbedo = cDiff / cLight
            cAlbedo = np.clip(cAlbedo, 1e-3, 1 / albedoPreds[-1].max().data.item())
            cLight = cDiff / cAlbedo
        blob[f"{tag}_diffuse"], blob[f"{tag}_spec"], blob[f"{tag}_im"] = diffusePred.numpy(), specularPred.numpy(), imBatchSmall.numpy()
        blob[f"{tag}_diffuseNew"], blob[f"{tag}_specNew"] = diffusePredNew.numpy(), specularPredNew.numpy()
        blob[f"{tag}_albedo"] = albedoPreds[-1].numpy()
        blob[f"{tag}_ref"] = np.array([cLight, cAlbedo, cDiff, cSpec], dtype=np.float64)
        print(tag, "cLight, cAlbedo, cDiff, cSpec =", blob[f"{tag}_ref"])
    path = os.path.join(OUT, "g6_shading.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) / 1e3, "KB")


if __name__ == "__main__":
    main()
