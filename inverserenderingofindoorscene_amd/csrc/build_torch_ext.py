"""Builds ``../libsgrender_torch.so`` -- the C++ torch extension (``sgr_torch.cpp``: TORCH_LIBRARY(sgrender) schemas, HIP-device /
Meta / Autograd kernels around the C ABI of libsgrender.so) -- in-tree, with one g++ invocation against the installed PyTorch-ROCm.

    python inverserenderingofindoorscene_amd/csrc/build_torch_ext.py [--force]

Host-only C++ (no device code: the kernels live in libsgrender.so, which this extension resolves with dlopen at first use), so
plain g++ is enough; the two defines are the ones PyTorch-ROCm's own headers expect when they are included from host code
(``torch.utils.cpp_extension`` passes the same).  Called by ``__graft_entry__.build()``; rebuilt only when a source is newer."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libsgrender_torch.so")
SOURCES = [os.path.join(HERE, "sgr_torch.cpp")]
DEPS = SOURCES + [os.path.join(HERE, "..", "..", "include", "sgrender.h"), os.path.abspath(__file__)]


def build(force: bool = False) -> str:
    if not force and os.path.isfile(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include"]
    libdir = ce.library_paths()[0]
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wall", "-Wno-unused-function", "-Wno-sign-compare"]
           + inc + SOURCES + ["-o", OUT, f"-L{libdir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ldl",
                              f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN"])
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
