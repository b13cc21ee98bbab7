"""CPU: the oracle's restatement of the cascade-0 step (decoder heads -> output2env -> LSregress / log-L2 ->
forwardEnv -> LSregressDiffSpec -> masked L2, wrapperBRDFLight.py:164-207) against the fixtures captured from the
UNMODIFIED reference wrapper (oracle/make_golden_wrapper.py).  Pins the oracle at wrapper level."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from test_gpu_wrapper import CASES, REC_W, REN_W, _load, _oracle64


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_wrapper(name):
    z, cfg, t = _load(name)
    s = cfg["s"]
    o = _oracle64(t, cfg)
    # scalars: the reference's fp32 evaluation vs the fp64 restatement on the same inputs
    assert abs(o["reconstErr"] - float(z["ref32_reconstErr"])) < 2e-5 * max(1.0, abs(o["reconstErr"]))
    assert abs(o["renderErr"] - float(z["ref32_renderErr"])) < 2e-5 * max(1.0, abs(o["renderErr"]))
    assert abs(REN_W * o["renderErr"] + REC_W * o["reconstErr"] - float(z["ref32_total"])) < 1e-4
    sub = (slice(None), slice(None), slice(None, None, s), slice(None, None, s))
    sub2 = (slice(None), slice(None), slice(None, None, 2 * s), slice(None, None, 2 * s))
    assert rel_l2(z["ref32_rendered"], o["rendered"]) < 1e-4
    assert rel_l2(z["ref32_diffuse"], o["diffuse"]) < 1e-4
    assert rel_l2(z["ref32_spec"], o["spec"]) < 2e-4
    assert rel_l2(z["ref32_envmapsPred"], o["envmapsPred"][sub]) < 1e-5
    assert rel_l2(z["ref32_envScaled"], o["envScaled"][sub2]) < 1e-4
    for k in ("gx_axis", "gx_lamb", "gx_weight"):
        assert rel_l2(z["ref32_" + k], o[k][sub]) < 5e-4, (k, rel_l2(z["ref32_" + k], o[k][sub]))
