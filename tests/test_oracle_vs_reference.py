"""CPU, authoring container only: oracle against the live, unmodified reference imported
from /root/reference (skipped on machines where it is not mounted, e.g. the GPU box)."""
import pytest
import torch

from conftest import rel_l2
from oracle import ref_import as RI
from oracle import sg_oracle as O

pytestmark = pytest.mark.skipif(not RI.available(), reason="/root/reference not mounted")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 5e-6)])
def test_forward_matches_live_reference(dtype, tol):
    bn, imH, imW, R, C, K = 1, 12, 16, 6, 8, 12
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=99, dtype=dtype)
    o2e, rl = RI.make_layers(K, R, C, dtype=dtype)
    env_r, _, lam_r, w_r = o2e.output2env(inp["axis"], inp["lamb"], inp["weight"])
    d_r, s_r = rl.forwardEnv(inp["albedo"], inp["normal"], inp["rough"], env_r)
    env_o, d_o, s_o = O.render_from_sg(inp["albedo"], inp["normal"], inp["rough"], inp["axis"], inp["lamb"], inp["weight"])
    assert rel_l2(env_o, env_r) < tol and rel_l2(d_o, d_r) < tol and rel_l2(s_o, s_r) < tol
    M = RI.models()
    im_s = torch.nn.functional.adaptive_avg_pool2d(inp["im"], (R, C))
    a, b = M.LSregressDiffSpec(d_r, s_r, im_s, d_r, s_r)
    a2, b2 = O.lsregress_diffspec(d_r, s_r, im_s, d_r, s_r)
    assert rel_l2(a2, a) < tol and rel_l2(b2, b) < tol
    c = M.LSregress(env_r, inp["env_gt"], env_r)
    c2 = O.lsregress(env_r, inp["env_gt"], env_r)
    assert rel_l2(c2, c) < tol


def test_view_vectors_and_tables_bit_equal():
    import numpy as np
    for (R, C, fov) in [(120, 160, 57), (6, 8, 42.75), (90, 160, 57)]:
        _, rl = RI.make_layers(12, R, C, fov=fov)
        v = O.view_vectors(C, R, fov)
        assert np.array_equal(v, rl.v[0].numpy())
        cam = [0.1, -0.05, 0.2]
        _, rl = RI.make_layers(12, R, C, fov=fov, cameraPos=cam)
        assert np.array_equal(O.view_vectors(C, R, fov, cam), rl.v[0].numpy())
    for eh, ew in [(8, 16), (16, 32), (4, 8)]:
        o2e, rl = RI.make_layers(12, 6, 8, eh, ew)
        ls, om = O.direction_table(eh, ew)
        assert np.array_equal(ls, rl.ls.numpy())
        assert np.array_equal(om, rl.envWeight.reshape(-1).numpy())
