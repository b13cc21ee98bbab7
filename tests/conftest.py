"""pytest configuration: marker registration and shared helpers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["g1_q4_k12", "g2_q1_k5", "g3_edges"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("bn", "imH", "imW", "R", "C", "K", "eh", "ew", "seed"):
        cfg[k] = int(cfg[k])
    return z, cfg


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def rel_max(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return (request.param,) + load_golden(request.param)
