"""Import the *unmodified* reference from /root/reference.  TEST INFRASTRUCTURE ONLY.

Only usable in the authoring container (the GPU box has no /root/reference).
Used to (1) validate ``oracle/sg_oracle.py`` against the real implementation
and (2) generate the committed fixtures under ``tests/golden/``
(``oracle/make_golden.py``).  Nothing here copies reference source: the
reference modules are imported where they lie.
"""
from __future__ import annotations

import importlib
import os
import sys

import torch

REFERENCE_ROOT = os.environ.get("SGR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models.py"))


def _import(name: str):
    if not available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module(name)


def models():
    """The reference's ``models`` module (models.py needs only torch + numpy)."""
    return _import("models")


def wrapper_brdf_light():
    """The reference's ``wrapperBRDFLight`` module with its unconditional ``.cuda()`` calls
    neutralised on a GPU-less machine (wrapperBRDFLight.py:15-37)."""
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # type: ignore[assignment]
    return _import("wrapperBRDFLight")


def _to_dtype(obj, dtype):
    for name, val in list(vars(obj).items()):
        if torch.is_tensor(val) and val.is_floating_point():
            setattr(obj, name, val.to(dtype))
    return obj


def make_layers(K: int, R: int, C: int, eh: int = 8, ew: int = 16, fov: float = 57, F0: float = 0.05,
                dtype=torch.float32, cameraPos=(0, 0, 0)):
    """Reference ``(output2env, renderingLayer)`` on CPU, tables cast to ``dtype``.

    With ``dtype=torch.float64`` this is "the reference code run in fp64" that
    SURVEY.md uses as its noise-floor yardstick.
    """
    m = models()
    o2e = m.output2env(SGNum=K, envWidth=ew, envHeight=eh, isCuda=False)
    rl = m.renderingLayer(imWidth=C, imHeight=R, fov=fov, F0=F0, cameraPos=list(cameraPos), envWidth=ew, envHeight=eh, isCuda=False)
    return _to_dtype(o2e, dtype), _to_dtype(rl, dtype)
