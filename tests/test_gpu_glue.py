"""GPU: the glue either side of the render path against fixtures made by the UNMODIFIED
reference: utils.predToShading and the cLight / cAlbedo post-scale of testReal.py:421-432 (oracle/make_golden_shading.py
-> g6_shading.npz), and the light encoder's input of wrapperBRDFLight.py:138-156 (captured by a forward pre-hook in
oracle/make_golden_wrapper.py -> g5_wrapper_*.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, rel_l2, rel_max

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_pred_to_shading_vs_reference_fixture(sgr, tag):
    z = np.load(os.path.join(GOLDEN_DIR, "g6_shading.npz"))
    K, R, C, eh, ew = [int(v) for v in z[tag + "_cfg"]]
    got = sgr.predToShading(z[tag + "_pred"], envWidth=ew, envHeight=eh, SGNum=K)
    assert got.shape == (3, R, C)
    e_ref = rel_l2(z[tag + "_ref32"], z[tag + "_ref64"])          # the reference's own fp32 error
    assert rel_l2(got, z[tag + "_ref32"]) < 1e-4 and rel_max(got, z[tag + "_ref32"]) < 2e-4
    assert rel_l2(got, z[tag + "_ref64"]) <= max(2.0 * e_ref, 1e-5), (rel_l2(got, z[tag + "_ref64"]), e_ref)
    # batched tensor in, tensor out
    t = sgr.predToShading(torch.from_numpy(z[tag + "_pred"]).cuda().repeat(2, 1, 1, 1), envWidth=ew, envHeight=eh, SGNum=K)
    assert tuple(t.shape) == (2, 3, R, C) and rel_l2(t[1].cpu(), got) < 1e-6


@pytest.mark.parametrize("tag", ["s1", "s2", "s3", "s4"])
def test_light_albedo_scale_vs_reference_fixture(sgr, tag):
    """testReal.py:413-432: LSregressDiffSpec (live first arguments are data there: no grad), then cLight / cAlbedo."""
    z = np.load(os.path.join(GOLDEN_DIR, "g6_shading.npz"))
    t = {k: torch.from_numpy(z[f"{tag}_{k}"]).cuda() for k in ("diffuse", "spec", "im", "albedo", "diffuseNew", "specNew")}
    dn, sn = sgr.LSregressDiffSpec(t["diffuse"], t["spec"], t["im"], t["diffuse"], t["spec"])
    assert rel_l2(dn.cpu(), z[f"{tag}_diffuseNew"]) < 1e-5
    cLight, cAlbedo = sgr.light_albedo_scale(dn, t["diffuse"], sn, t["spec"], t["albedo"])
    assert cLight.dim() == 0 and cLight.is_cuda
    ref = z[f"{tag}_ref"]
    assert abs(cLight.item() - ref[0]) <= 2e-5 * abs(ref[0]), (cLight.item(), ref[0])
    assert abs(cAlbedo.item() - ref[1]) <= 2e-5 * abs(ref[1]), (cAlbedo.item(), ref[1])
    # same from the reference's own scaled images
    cL2, cA2 = sgr.light_albedo_scale(t["diffuseNew"], t["diffuse"], t["specNew"], t["spec"], t["albedo"])
    assert abs(cL2.item() - ref[0]) <= 1e-5 * abs(ref[0]) and abs(cA2.item() - ref[1]) <= 1e-5 * abs(ref[1])


@pytest.mark.parametrize("name", ["g5_wrapper_small", "g5_wrapper_120x160"])
def test_light_encoder_input_vs_wrapper_fixture(sgr, name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    t = {k: torch.from_numpy(z[k]).cuda() for k in ("im", "albedo_raw", "depth_raw", "normalPred", "roughPred")}
    out, alb_n, dep_n = sgr.light_encoder_input(t["im"], t["albedo_raw"], t["normalPred"], t["roughPred"], t["depth_raw"])
    bn = t["im"].shape[0]
    assert tuple(out.shape) == (bn, 11, 480, 640)
    assert rel_l2(alb_n.cpu(), z["albedoPred"]) < 1e-6                       # the normalised albedo the wrapper returns (:139-142)
    assert rel_l2(out[:, :, ::8, ::8].cpu(), z["light_in"]) < 1e-6 and rel_max(out[:, :, ::8, ::8].cpu(), z["light_in"]) < 1e-5
    s1 = out.double().sum(dim=(0, 2, 3)).cpu().numpy()
    s2 = (out.double() ** 2).sum(dim=(0, 2, 3)).cpu().numpy()
    assert np.allclose(s1, z["light_in_sum"][0], rtol=1e-5) and np.allclose(s2, z["light_in_sum"][1], rtol=1e-5)
    # other target sizes follow torch's own resize
    out2, _, _ = sgr.light_encoder_input(t["im"], t["albedo_raw"], t["normalPred"], t["roughPred"], t["depth_raw"], size=(100, 150))
    want = torch.nn.functional.interpolate(t["im"], [100, 150], mode="bilinear")
    assert rel_l2(out2[:, :3].cpu(), want.cpu()) < 1e-6


def test_cascade_handoff_files_round_trip_through_the_path(sgr, tmp_path):
    """SURVEY.md 8f rank 4 end to end on the GPU: cascade 0's export (outputBRDFLight.py:246-301) -- the packed raw SG parameters out of
    `light_heads(need_packed=True)` and the rendered diffuse / specular images -- written as the reference's lzf-HDF5 files
    (`sgr.write_cascade_handoff`), read back the way cascade 1's loader does (`sgr.read_cascade_handoff`, dataLoader.py:97-105,277-283), split
    with `unpack_envmaps` and rendered again: the files are lossless, so every tensor and the second render are BIT-identical."""
    import os
    bn, K, R, C = 3, 12, 12, 16
    g = torch.Generator().manual_seed(41)
    xa, xl, xw = [torch.randn(s, generator=g).cuda() for s in ((bn, 3 * K, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
    axis, lamb, weight, packed = sgr.light_heads(xa, xl, xw, need_packed=True)
    assert tuple(packed.shape) == (bn, 7 * K, R, C)
    albedo, rough = torch.rand(bn, 3, 2 * R, 2 * C, generator=g).cuda(), (torch.rand(bn, 1, 2 * R, 2 * C, generator=g) * 2 - 1).cuda()
    n = torch.randn(bn, 3, 2 * R, 2 * C, generator=g)
    n[:, 2] = n[:, 2].abs() + 0.5
    normal = (n / n.norm(dim=1, keepdim=True)).cuda()
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    _, d0, s0 = layer.forwardSG(albedo, normal, rough, axis, lamb, weight, need_env=False)
    ims = [str(tmp_path / f"scene{i:04d}" / f"im_{i + 1}.hdr") for i in range(bn)]
    for p in ims:
        os.makedirs(os.path.dirname(p), exist_ok=True)
    ind = torch.tensor([1.0, 1.0, 0.0]).reshape(bn, 1, 1, 1).cuda()
    written = sgr.write_cascade_handoff(packed, d0, s0, ims, envmapsInd=ind)
    assert len(written) == 3 * bn - 1
    for i in range(bn):
        got = sgr.read_cascade_handoff(ims[i])
        assert torch.equal(torch.from_numpy(got["diffuse"]).cuda(), d0[i]) and torch.equal(torch.from_numpy(got["specular"]).cuda(), s0[i])
        if i == 2:
            assert got["env"] is None                      # envmapsInd == 0: no env file (outputBRDFLight.py:292-301)
            continue
        env = torch.from_numpy(got["env"]).cuda().unsqueeze(0)
        assert torch.equal(env[0], packed[i])
        a1, l1, w1 = sgr.unpack_envmaps(env, K)
        assert torch.equal(a1[0], axis[i]) and torch.equal(l1[0], lamb[i]) and torch.equal(w1[0], weight[i])
        _, d1, s1 = layer.forwardSG(albedo[i:i + 1], normal[i:i + 1], rough[i:i + 1], a1.contiguous(), l1.contiguous(), w1.contiguous(), need_env=False)
        assert torch.equal(d1[0], d0[i]) and torch.equal(s1[0], s0[i])
