"""GPU: the kernels that are NOT the default dispatch for the fixtures' shapes, selected through the library's tuning
knobs (read once per process, hence the subprocesses), must pass the same golden-fixture parity tests:

  SGR_BWD_MODE=split              two-wave lobe-split backward (sg_bwd_split_kernel)
  SGR_FWD_MODE/SGR_BWD_MODE=half2 half-wave forward for every forward variant, half-wave backward built for 2 waves/SIMD
  SGR_FWD_MODE/SGR_BWD_MODE=half3 the same built for 3 waves/SIMD (round 1's defaults; the packed-fp32 kernels are round 2's)
  SGR_FWD_MODE=full               one-pixel-per-lane forward also for the SG -> env call
  SGR_FWD_MODE=pk|pkhalf2|pkhalf3 packed fp32 forward: one pixel per lane / half-wave for every forward variant (the default mixes them)
  SGR_FWD_MODE=pkhalf2w|pkhalf3w  half-wave with two table rows per env flush (round 3; 3w is the default whenever the env image is written)
  SGR_BRDF_MODE=scalar|pk         BRDF-map adjoint with the env image given: round 1's scalar kernel / packed, one pixel per lane (default: packed half-wave)
  SGR_TAN_HANDOFF=1               the fused forward hands the post-tan SG parameters to its backward (premap mode 2; off by default)
  SGR_GENERIC=1                   generic kernels (table-driven, any direction grid) on the reference's grid
  SGR_F1_MODE=half                half-wave statistics kernel in the fused objective's forward (objective tests)
  SGR_F1_MODE/SGR_B1_MODE=scalar  round 1's scalar objective kernels (the packed-fp32 ones are the default)
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = "golden or trainlight"


@pytest.mark.parametrize("env", [
    {"SGR_BWD_MODE": "split"},
    {"SGR_FWD_MODE": "half2", "SGR_BWD_MODE": "half2"},
    {"SGR_FWD_MODE": "half3", "SGR_BWD_MODE": "half3"},
    {"SGR_FWD_MODE": "full"},
    {"SGR_FWD_MODE": "pk"},
    {"SGR_FWD_MODE": "pkhalf2"},
    {"SGR_FWD_MODE": "pkhalf3"},
    {"SGR_FWD_MODE": "pkhalf2w"},
    {"SGR_BRDF_MODE": "scalar"},
    {"SGR_BRDF_MODE": "pk", "SGR_TAN_HANDOFF": "1"},
    {"SGR_GENERIC": "1"},
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


@pytest.mark.parametrize("env", [{"SGR_F1_MODE": "half"}, {"SGR_F1_MODE": "scalar", "SGR_B1_MODE": "scalar"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.timeout(600)
def test_alternative_objective_kernels_pass_objective_tests(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_objective.py"), "-q", "-m", "gpu", "-x",
                        "-k", "golden or oracle", "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=580)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
