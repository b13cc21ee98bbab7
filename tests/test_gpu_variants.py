"""GPU: the kernels that are NOT the default dispatch for the fixtures' shapes, selected through the two knobs that survive
round 4 (read once per process, hence the subprocesses), must pass the same golden-fixture parity tests:

  SGR_GENERIC=1        generic kernels (table-driven, any direction grid) on the reference's grid -- the any-grid fallback
  SGR_TAN_HANDOFF=1    the fused forward hands the post-tan SG parameters to its backward (premap mode 2; off by default)

Round 1-3's superseded kernels (scalar one-pixel-per-lane / half-wave / two-wave split forms, the scalar objective and BRDF-adjoint
kernels) and the SGR_FWD_MODE / SGR_BWD_MODE / SGR_F1_MODE / SGR_B1_MODE / SGR_BRDF_MODE / SGR_FWD_TJ knobs that selected them are gone;
their A/B records stay under profiles/ (r01*, r02*, r03c_fwd_mode_sweep.txt).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = "golden or trainlight"


@pytest.mark.parametrize("env", [
    {"SGR_GENERIC": "1"},
    {"SGR_TAN_HANDOFF": "1"},
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.timeout(600)
def test_alternative_kernels_pass_golden_parity(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                        "-k", SUBSET, "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=580)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.timeout(600)
def test_tan_handoff_objective():
    """premap mode 2 through the fused objective (forward writes the post-tan tensors, backward reads them)."""
    e = dict(os.environ)
    e["SGR_TAN_HANDOFF"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_objective.py"), "-q", "-m", "gpu", "-x",
                        "-k", "golden or oracle", "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=580)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
