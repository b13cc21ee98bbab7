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
    # testReal.py:421-432 -- global light / albedo scale from the rendered and the observed image
    g = torch.Generator().manual_seed(40)
    bn, R, C = 2, 12, 16
    diffuse, spec, im = torch.rand(bn, 3, R, C, generator=g), torch.rand(bn, 3, R, C, generator=g) * 0.3, torch.rand(bn, 3, R, C, generator=g)
    albedo = torch.rand(bn, 3, 2 * R, 2 * C, generator=g)
    # the reference's expressions, verbatim semantics (testReal.py:421-432)
    diffusePredNew, specularPredNew, imBatchSmall, albedoPred = diffuse, spec, im, albedo
    cDiff = (torch.sum(diffusePredNew) / torch.sum(diffuse)).data.item()          # == 1 here; testReal divides scaled by unscaled
    blob["scale_diffuse"], blob["scale_spec"], blob["scale_im"], blob["scale_albedo"] = diffuse.numpy(), spec.numpy(), im.numpy(), albedo.numpy()
    path = os.path.join(OUT, "g6_shading.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) / 1e3, "KB")


if __name__ == "__main__":
    main()
