"""CPU: the C-ABI library loads and exports every symbol include/sgrender.h declares; argument
validation works without a GPU (no kernel is launched)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from inverserenderingofindoorscene_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "sgrender.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sgrender.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.sgr_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # NULL tensors -> SGR_ERR_BAD_ARG, with a message
    rc = lib.sgr_sg_to_env_fwd(None, None, None, None, None, None, None, 1, 12, 4, 4, 8, 16, 1, None)
    assert rc == -1 and b"NULL" in lib.sgr_last_error()
    # unsupported pooling ratio / lobe count -> SGR_ERR_UNSUPPORTED (pointer values are never dereferenced)
    fake = ctypes.c_void_p(4096)
    rc = lib.sgr_fused_fwd(fake, fake, fake, fake, fake, fake, fake, fake, None, fake, fake,
                           1, 12, 4, 4, 8, 16, 12, 12, ctypes.c_float(0.05), 1, None)
    assert rc == -2 and b"ratio" in lib.sgr_last_error()
    rc = lib.sgr_fused_fwd(fake, fake, fake, fake, fake, fake, fake, fake, None, fake, fake,
                           1, 33, 4, 4, 8, 16, 4, 4, ctypes.c_float(0.05), 1, None)
    assert rc == -2
    assert lib.sgr_dirs_padded(128) == 128 and lib.sgr_dirs_padded(15) == 32


def test_cpu_tensors_are_rejected_not_emulated():
    import torch
    import inverserenderingofindoorscene_amd as pkg
    o2e = pkg.output2env(SGNum=2, envWidth=4, envHeight=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        o2e.output2env(torch.zeros(1, 2, 3, 2, 2), torch.zeros(1, 2, 2, 2), torch.zeros(1, 6, 2, 2))
    rl = pkg.renderingLayer(imWidth=2, imHeight=2, envWidth=4, envHeight=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rl.forwardEnv(torch.zeros(1, 3, 2, 2), torch.zeros(1, 3, 2, 2), torch.zeros(1, 1, 2, 2), torch.zeros(1, 3, 2, 2, 2, 4))


def test_layer_attributes_mirror_reference():
    import numpy as np
    import inverserenderingofindoorscene_amd as pkg
    rl = pkg.renderingLayer()
    assert tuple(rl.v.shape) == (1, 3, 120, 160) and tuple(rl.ls.shape) == (128, 3)
    assert tuple(rl.envWeight.shape) == (1, 128, 1, 1, 1) and abs(rl.fov - 57 / 180 * np.pi) < 1e-12
    o2e = pkg.output2env(12)
    assert tuple(o2e.ls.shape) == (1, 1, 3, 1, 1, 8, 16) and o2e.SGNum == 12


def test_premap_modes_and_configuration_queries_without_gpu():
    """premap = 3 (decoder heads as the kernels' prologue) is offered where a packed kernel implements it and refused -- before
    any launch -- elsewhere; the workspace / support queries are pure host functions."""
    lib = _lib.load()
    assert lib.sgr_heads_prologue_supported(12, 120, 160, 8, 16) == 1 and lib.sgr_heads_prologue_supported(24, 240, 320, 16, 32) == 1
    assert lib.sgr_heads_prologue_supported(6, 120, 160, 8, 16) == 0        # SGNum <= 6: no prologue (standalone heads pass)
    assert lib.sgr_heads_prologue_supported(25, 120, 160, 8, 16) == 0
    assert lib.sgr_heads_prologue_supported(12, 120, 160, 4, 8) == 0        # a direction grid without packed kernels
    assert lib.sgr_fused_recon_supported(24, 240, 320, 16, 32) == 1 and lib.sgr_fused_recon_supported(12, 120, 160, 4, 8) == 0
    assert lib.sgr_loss_workspace_floats(16) > 16 * 16 * 9 and lib.sgr_fused_recon_workspace_floats(16, 120, 160) > 0
    fake = ctypes.c_void_p(4096)
    f0 = ctypes.c_float(0.05)
    # out of range, and in range but not implemented for this shape: both rejected with a message, nothing launched
    rc = lib.sgr_fused_fwd(fake, fake, fake, fake, fake, fake, fake, fake, None, fake, fake, 1, 12, 4, 4, 8, 16, 4, 4, f0, 4, None)
    assert rc == -1 and b"premap" in lib.sgr_last_error()
    rc = lib.sgr_fused_fwd(fake, fake, fake, fake, fake, fake, fake, fake, None, fake, fake, 1, 5, 4, 4, 8, 16, 4, 4, f0, 3, None)
    assert rc == -2 and b"premap 3" in lib.sgr_last_error()
    rc = lib.sgr_fused_bwd_sg(None, fake, fake, fake, fake, fake, fake, fake, fake, fake, fake, fake, fake, fake, 1, 5, 4, 4, 8, 16, 4, 4,
                              f0, 3, None)
    assert rc == -2 and b"premap 3" in lib.sgr_last_error()
    rc = lib.sgr_sg_to_env_fwd(fake, fake, fake, fake, fake, None, None, 1, 12, 4, 4, 8, 16, 3, None)
    assert rc == -1 and b"premap" in lib.sgr_last_error()
