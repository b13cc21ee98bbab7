"""Golden vectors for the light-decoder output heads, produced by the UNMODIFIED reference.  TEST INFRASTRUCTURE.

The head activations live inline at the end of ``models.decoderLight.forward`` (models.py:336-346), so three
real decoders (modes 0 / 1 / 2, random weights) are run on small random feature maps; a forward hook on
``dconvFinal`` captures the pre-activation tensor (made a leaf for the gradient by ``retain_grad``), and the
module's own output and autograd gradient are the reference values.

    python -m oracle.make_golden_heads        # writes tests/golden/g4_heads.npz (authoring container only)
"""
import os

import numpy as np
import torch

from oracle import ref_import as RI

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BN, K, R, C = 2, 12, 6, 10


def run_decoder(mode, seed, dtype):
    M = RI.models()
    torch.manual_seed(seed)
    dec = M.decoderLight(K, mode).to(dtype)
    with torch.no_grad():      # spread the pre-activations over the interesting range of tanh / clamp
        dec.dconvFinal.weight.mul_(6.0)
    g = torch.Generator().manual_seed(seed + 100)
    shapes = [(128, 16), (256, 8), (256, 4), (512, 2), (512, 1), (1024, 1)]           # x1..x6: channels, spatial scale
    xs = [torch.randn(BN, ch, sc, sc, generator=g).to(dtype) for ch, sc in shapes]
    env = torch.zeros(BN, 3, R, C, dtype=dtype)
    cap = {}

    def hook(_m, _i, out):
        out.retain_grad()
        cap.setdefault("x", []).append(out)
    h = dec.dconvFinal.register_forward_hook(hook)
    y = dec(*xs, env)
    h.remove()
    x = cap["x"][-1]            # forward evaluates dconvFinal twice (x_orig, then the one that is used)
    ct = torch.randn(y.shape, generator=g).to(dtype)
    (y * ct).sum().backward()
    return x.detach(), y.detach(), ct, x.grad.detach()


def main():
    if not RI.available():
        raise SystemExit("reference not mounted; fixtures can only be generated in the authoring container")
    blob = {}
    for name, mode, seed in (("axis", 0, 1), ("lamb", 1, 2), ("weight", 2, 3)):
        x32, y32, ct, g32 = run_decoder(mode, seed, torch.float32)
        x64, y64, _, g64 = run_decoder(mode, seed, torch.float64)
        # fp64 run re-draws the same weights/features in double: its pre-activation differs in the last bits, so the
        # fp64 reference is re-evaluated ON the fp32 pre-activation by the oracle in the tests; keep the fp32 set here
        blob[f"x_{name}"] = x32.numpy()
        blob[f"y_{name}"] = y32.numpy()
        blob[f"ct_{name}"] = ct.numpy()
        blob[f"gx_{name}"] = g32.numpy()
    path = os.path.join(OUT, "g4_heads.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) / 1e3, "KB", {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
