"""The cascade hand-off container: the reference's ``*_0.h5`` files, without h5py.

Between the two cascades the reference goes through the file system: ``outputBRDFLight.py:246-301`` writes, per image, the packed raw
SG parameters ``imenv_*_0.h5 [7*SGNum = 84, envRow, envCol]`` (``wrapperBRDFLight.py:167-168,216-223``), the rendered
``imdiffuse_*_0.h5 / imspecular_*_0.h5 [3, envRow, envCol]`` and the BRDF maps, each with ``utils.writeH5ToFile`` (``utils.py:92-99``:
one float32 dataset ``data``, h5py ``compression='lzf'``); ``dataLoader.py:97-105,277-283`` reads them back with ``loadH5``.  This module
keeps those two function names and their behaviour on top of ``libsgrender_h5.so`` (``include/sgrender_h5.h``: libhdf5 located at run
time, h5py's LZF filter 32000 with this package's own LZF coder).  Files written here open in h5py as ``compression == 'lzf'``; files
written by h5py read here bit for bit (``tests/test_h5_handoff.py``).

Host side by nature: the tensors of the path live in HBM and are copied to the host for the write, as the reference does
(``.data.cpu().numpy()``, ``utils.py:96``).  A missing ``libsgrender_h5.so`` or libhdf5 raises :class:`SgrenderUnavailable`; there is no
fallback format."""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import numpy as np

from ._lib import SgrenderError, SgrenderUnavailable

__all__ = ["writeH5ToFile", "loadH5", "h5_available", "h5_dataset_info", "handoff_names", "write_cascade_handoff", "read_cascade_handoff"]

_H5 = None
_MAX_DIMS = 8
_ULL = ctypes.c_ulonglong


def _load():
    global _H5
    if _H5 is not None:
        return _H5
    path = os.environ.get("SGR_H5_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsgrender_h5.so")
    if not os.path.isfile(path):
        raise SgrenderUnavailable(f"{path} not found: build it with `python __graft_entry__.py` (g++ csrc/sgr_h5.cpp)")
    lib = ctypes.CDLL(path)
    lib.sgr_h5_last_error.restype = ctypes.c_char_p
    lib.sgr_h5_available.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    lib.sgr_h5_write_f32.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_ULL), ctypes.c_int]
    lib.sgr_h5_shape.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_ULL)]
    lib.sgr_h5_read_f32.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, _ULL]
    lib.sgr_h5_dataset_info.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_ULL)]
    for fn in (lib.sgr_lzf_compress, lib.sgr_lzf_decompress):
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    if lib.sgr_h5_abi_version() != 1:
        raise SgrenderUnavailable(f"{path}: ABI version {lib.sgr_h5_abi_version()}, this package expects 1")
    _H5 = lib
    return lib


def _check(lib, rc: int) -> None:
    if rc == 0:
        return
    msg = (lib.sgr_h5_last_error() or b"").decode(errors="replace")
    if rc == -2:
        raise SgrenderUnavailable(msg)
    raise SgrenderError(msg or f"sgrender_h5: error {rc}")


def h5_available() -> bool:
    """Whether ``libsgrender_h5.so`` is built and finds a libhdf5 (>= 1.10) to drive."""
    try:
        return bool(_load().sgr_h5_available(None))
    except SgrenderUnavailable:
        return False


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, "detach"):          # a torch tensor, on any device (utils.py:96: .data.cpu().numpy())
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def _write(path: str, arr: np.ndarray, compression: bool = True, name: str = "data") -> None:
    lib = _load()
    if arr.ndim < 1 or arr.ndim > _MAX_DIMS:
        raise SgrenderError(f"sgrender_h5: rank {arr.ndim} outside 1..{_MAX_DIMS}")
    dims = (_ULL * arr.ndim)(*arr.shape)
    _check(lib, lib.sgr_h5_write_f32(os.fsencode(path), name.encode(), arr.ctypes.data_as(ctypes.c_void_p), arr.ndim, dims, 1 if compression else 0))


def writeH5ToFile(imBatch, nameBatch: Sequence[str]) -> None:
    """``utils.writeH5ToFile`` (utils.py:92-99): image ``n`` of the batch ``[bn, ch, H, W]`` goes to the file ``nameBatch[n]`` as the float32
    dataset ``data`` of shape ``[ch, H, W]``, LZF-compressed; an existing file is truncated (h5py mode ``'w'``)."""
    bn = imBatch.shape[0]
    assert bn == len(nameBatch)
    for n in range(bn):
        _write(nameBatch[n], _to_numpy(imBatch[n]))


def loadH5(imName: str) -> Optional[np.ndarray]:
    """``dataLoader.loadH5`` (dataLoader.py:277-283): the dataset ``data`` of the file as a float32 array -- and, like the reference's
    bare ``except``, ``None`` when the file is missing, is no HDF5, or holds no float32 ``data`` (the reference's loader carries on with
    ``None``).  A missing library is NOT swallowed: :class:`SgrenderUnavailable` propagates."""
    lib = _load()
    nd = ctypes.c_int(0)
    dims = (_ULL * _MAX_DIMS)()
    try:
        _check(lib, lib.sgr_h5_shape(os.fsencode(imName), b"data", ctypes.byref(nd), dims))
        shape = tuple(int(dims[i]) for i in range(nd.value))
        out = np.empty(shape, dtype=np.float32)
        _check(lib, lib.sgr_h5_read_f32(os.fsencode(imName), b"data", out.ctypes.data_as(ctypes.c_void_p), out.size))
        return out
    except SgrenderUnavailable:
        raise
    except SgrenderError:
        return None


def h5_dataset_info(path: str, name: str = "data") -> dict:
    """``{"filter": 32000 | 0, "compression": "lzf" | None | "filter <id>", "chunks": tuple | None}`` of a dataset (what h5py reports as
    ``dset.compression`` / ``dset.chunks``)."""
    lib = _load()
    fid, nd = ctypes.c_int(0), ctypes.c_int(0)
    chunk = (_ULL * _MAX_DIMS)()
    _check(lib, lib.sgr_h5_dataset_info(os.fsencode(path), name.encode(), ctypes.byref(fid), ctypes.byref(nd), chunk))
    comp = "lzf" if fid.value == 32000 else (None if fid.value == 0 else f"filter {fid.value}")
    return {"filter": fid.value, "compression": comp, "chunks": tuple(int(chunk[i]) for i in range(nd.value)) or None}


# ------------------------------------------------------------------------------------------------------------------------------------
# the cascade-0 export of outputBRDFLight.py:246-301 for the three tensors the render path produces
# ------------------------------------------------------------------------------------------------------------------------------------
def handoff_names(imName: str, cascadeLevel: int = 0) -> dict:
    """File names of outputBRDFLight.py:246-251 for one image name ``.../im_<id>.hdr``."""
    tail = "_%d.h5" % cascadeLevel
    return {"env": imName.replace("im_", "imenv_").replace(".hdr", tail),
            "diffuse": imName.replace("im_", "imdiffuse_").replace(".hdr", tail),
            "specular": imName.replace("im_", "imspecular_").replace(".hdr", tail)}


def write_cascade_handoff(envmapsPred, diffusePred, specularPred, imNameBatch: Sequence[str], envmapsInd=None, cascadeLevel: int = 0,
                          overwrite: bool = False) -> List[str]:
    """outputBRDFLight.py:277-301: ``envmapsPred [bn, 7*SGNum, R, C]`` (the packed raw SG parameters: ``light_heads(need_packed=True)`` or
    wrapperBRDFLight's ``isLightOut`` return), ``diffusePred / specularPred [bn, 3, R, C]`` -> ``imenv / imdiffuse / imspecular_*_<level>.h5``
    next to each image.  Like the reference it leaves existing files alone (unless ``overwrite``) and writes the env file only for images
    whose ``envmapsInd`` is 1.  Returns the paths written."""
    written = []
    bn = diffusePred.shape[0]
    assert bn == len(imNameBatch) == specularPred.shape[0] == envmapsPred.shape[0]
    ind = None if envmapsInd is None else _to_numpy(envmapsInd).reshape(bn)
    for n in range(bn):
        names = handoff_names(imNameBatch[n], cascadeLevel)
        for key, batch in (("diffuse", diffusePred), ("specular", specularPred)):
            if overwrite or not os.path.isfile(names[key]):
                writeH5ToFile(batch[n:n + 1], [names[key]])
                written.append(names[key])
        if (ind is None or ind[n] == 1) and (overwrite or not os.path.isfile(names["env"])):
            writeH5ToFile(envmapsPred[n:n + 1], [names["env"]])
            written.append(names["env"])
    return written


def read_cascade_handoff(imName: str, cascadeLevel: int = 0) -> dict:
    """What cascade 1's loader reads back for one image (dataLoader.py:97-105,160): ``{"env", "diffuse", "specular"}`` -> float32 arrays or
    ``None`` for a missing file."""
    return {k: loadH5(p) for k, p in handoff_names(imName, cascadeLevel).items()}
