"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP kernels, called through the
C ABI by the package's autograd functions, against
  * the golden fixtures produced by the unmodified reference (tests/golden/*.npz), and
  * the fp64 CPU oracle on seeded inputs,
with the tolerances of BASELINE.md section 3: rel-L2 <= 1e-4 and max-abs/max|ref| <= 2e-4
versus the reference's fp32 results (floating-point path: not bit-exact by nature), and an
error versus the fp64 arbiter no worse than ~the reference's own fp32 noise."""
import numpy as np
import pytest
import torch

from conftest import NAMES6, layer_kwargs, oracle_fwd_bwd, oracle_with_noise, rel_l2, rel_max, tol2

pytestmark = pytest.mark.gpu

NAMES = ("albedo", "normal", "rough", "axis", "lamb", "weight")
TOL_L2 = 1e-4
TOL_MAX = 2e-4


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    assert torch.cuda.is_available(), "these tests need a GPU"
    _lib.load()        # fail loudly if libsgrender.so is missing -- there is no fallback
    return pkg


def _dev_inputs(z, grad=()):
    x = {}
    for k in NAMES:
        t = torch.from_numpy(np.ascontiguousarray(z["in_" + k])).cuda()
        if k in grad:
            t.requires_grad_(True)
        x[k] = t
    return x


def _nondegenerate_mask(z, cfg):
    n = torch.from_numpy(z["in_normal"])
    pn = torch.nn.functional.adaptive_avg_pool2d(n, (cfg["R"], cfg["C"]))
    nn = (pn * pn).sum(1, keepdim=True)
    un = pn / nn.clamp(1e-6, 1).sqrt()
    ok = ((un[:, 1:2].abs() < 0.999) & (nn > 1e-4)).float()
    return torch.nn.functional.interpolate(ok, size=(cfg["imH"], cfg["imW"]), mode="nearest").numpy() > 0.5


def _check_fwd(name, z, got):
    for k, v in got.items():
        v = v.detach().cpu().numpy()
        ref32, ref64 = z["ref32_" + k], z["ref64_" + k]
        assert np.isfinite(v).all(), (name, k)
        assert rel_l2(v, ref32) < TOL_L2, (name, k, rel_l2(v, ref32))
        assert rel_max(v, ref32) < TOL_MAX, (name, k, rel_max(v, ref32))
        e_ref = rel_l2(ref32, ref64)
        # no worse than 2 x the reference's own fp32 error; floored at 2e-5, the fp32 noise level of this
        # path (the reference's own error is 2.4e-5 .. 4.4e-5 on decoder-range inputs; on the benign fixture
        # it happens to be 7e-6 and the table-driven generic kernels sit at 1.8e-5)
        assert rel_l2(v, ref64) < max(2 * e_ref, 2e-5), (name, k, rel_l2(v, ref64), e_ref)


def _check_grads(name, z, cfg, grads, which="glin"):
    mask = _nondegenerate_mask(z, cfg)
    for k, g in grads.items():
        g = g.detach().cpu().numpy()
        ref32, ref64 = z[f"ref32_{which}_{k}"], z[f"ref64_{which}_{k}"]
        m = np.broadcast_to(mask, g.shape) if k == "normal" else np.ones(g.shape, bool)
        assert np.isfinite(g).all(), (name, k)
        e_ref = rel_l2(ref32[m], ref64[m])
        # primary: the reference's own fp32 gradients.  Where fp64 is a valid arbiter both are within e_ref / 2 e_ref of it, so
        # 3 e_ref bounds their distance; for ratio-1 unit normals the |N|^2 == 1 clamp kink makes the reference's fp32 and
        # fp64 gradients differ by O(1) (e_ref ~ 1): the kernels follow the fp32 rounding, and only the contract's 1e-4 is left
        assert rel_l2(g[m], ref32[m]) < max(3 * e_ref if e_ref < 1e-3 else 0.0, 1e-4), (name, k, rel_l2(g[m], ref32[m]), e_ref)
        if e_ref < 1e-3:
            e = rel_l2(g[m], ref64[m])
            assert e < max(2 * e_ref, 1e-5), (name, k, e, e_ref)


def test_fused_forward_vs_golden(sgr, golden):
    name, z, cfg = golden
    x = _dev_inputs(z)
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    _check_fwd(name, z, dict(env=env, diffuse=d, spec=s))
    none, d2, s2 = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=False)
    assert none is None
    _check_fwd(name, z, dict(diffuse=d2, spec=s2))


def test_dropin_two_call_forward_vs_golden(sgr, golden):
    """The reference call sequence wrapperBRDFLight.py:177,194."""
    name, z, cfg = golden
    x = _dev_inputs(z)
    o2e = sgr.output2env(SGNum=cfg["K"], envWidth=cfg["ew"], envHeight=cfg["eh"])
    rl = sgr.renderingLayer(**layer_kwargs(cfg))
    env, axis, lam_t, w_t = o2e.output2env(x["axis"], x["lamb"], x["weight"])
    assert axis is x["axis"]
    d, s = rl.forwardEnv(x["albedo"], x["normal"], x["rough"], env)
    _check_fwd(name, z, dict(env=env, diffuse=d, spec=s, lamb_tan=lam_t, weight_tan=w_t))
    # fromSGtoIm on the post-tan values is the same env image
    env2 = o2e.fromSGtoIm(x["axis"], lam_t, w_t)
    assert rel_l2(env2.cpu(), env.cpu()) < 1e-6
    # the nn.Module aliases route to the same kernels
    env3, _, _, _ = sgr.output_radiance(cfg["K"], cfg["ew"], cfg["eh"])(x["axis"], x["lamb"], x["weight"])
    assert torch.equal(env3, env)
    d3, s3 = sgr.renderLayer(cfg["C"], cfg["R"], cfg["fov"], cfg["F0"], cfg["cam"], cfg["ew"], cfg["eh"])(
        x["albedo"], x["normal"], x["rough"], env)
    assert torch.equal(d3, d) and torch.equal(s3, s)


def _cotangents(z):
    return (torch.from_numpy(z["ct_env"]).cuda(), torch.from_numpy(z["ct_d"]).cuda(), torch.from_numpy(z["ct_s"]).cuda())


def test_fused_backward_vs_golden(sgr, golden):
    name, z, cfg = golden
    x = _dev_inputs(z, grad=NAMES)
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    ce, cd, cs = _cotangents(z)
    lin = (env * ce).sum() + (d * cd).sum() + (s * cs).sum()
    grads = torch.autograd.grad(lin, [x[k] for k in NAMES])
    _check_grads(name, z, cfg, dict(zip(NAMES, grads)))


def test_dropin_two_call_backward_vs_golden(sgr, golden):
    name, z, cfg = golden
    x = _dev_inputs(z, grad=NAMES)
    o2e = sgr.output2env(SGNum=cfg["K"], envWidth=cfg["ew"], envHeight=cfg["eh"])
    rl = sgr.renderingLayer(**layer_kwargs(cfg))
    env, _, _, _ = o2e.output2env(x["axis"], x["lamb"], x["weight"])
    d, s = rl.forwardEnv(x["albedo"], x["normal"], x["rough"], env)
    ce, cd, cs = _cotangents(z)
    lin = (env * ce).sum() + (d * cd).sum() + (s * cs).sum()
    grads = torch.autograd.grad(lin, [x[k] for k in NAMES])
    _check_grads(name, z, cfg, dict(zip(NAMES, grads)))


def test_brdf_grads_without_env_image(sgr, golden):
    """need_env=False: the BRDF-map adjoint re-evaluates the radiance from the SG lobes (brdf_bwd_pk_sg_kernel on the 8x16 grid)
    instead of reading the env image -- same gradients as the env-given kernel, and against the fixture like it."""
    name, z, cfg = golden
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    _, cd, cs = _cotangents(z)
    out = []
    for need_env in (False, True):
        x = _dev_inputs(z, grad=NAMES)
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=need_env)
        out.append(torch.autograd.grad((d * cd).sum() + (s * cs).sum(), [x[k] for k in NAMES]))
    mask = torch.from_numpy(_nondegenerate_mask(z, cfg)).cuda()
    for k, a, b in zip(NAMES, out[0], out[1]):
        m = mask.expand_as(a) if k == "normal" else torch.ones_like(a, dtype=torch.bool)
        assert rel_l2(a[m], b[m]) < 2e-5, (name, k, rel_l2(a[m], b[m]))


def test_sg_only_grads_like_trainlight(sgr, golden):
    """trainLight mode: only the SG parameters need gradients, no env cotangent."""
    name, z, cfg = golden
    x = _dev_inputs(z, grad=("axis", "lamb", "weight"))
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    _, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=False)
    _, cd, cs = _cotangents(z)
    g = torch.autograd.grad((d * cd).sum() + (s * cs).sum(), [x["axis"], x["lamb"], x["weight"]])
    # same thing through the oracle in fp64 -- and in fp32, whose distance from the former is the yardstick
    from oracle import sg_oracle as O

    def oracle(dtype):
        xo = {k: torch.from_numpy(z["in_" + k]).to("cuda", dtype) for k in NAMES}
        for k in ("axis", "lamb", "weight"):
            xo[k].requires_grad_(True)
        _, do, so = O.render_from_sg(xo["albedo"], xo["normal"], xo["rough"], xo["axis"], xo["lamb"], xo["weight"],
                                     cfg["eh"], cfg["ew"], cfg["fov"], cfg["F0"], cfg["cam"])
        return torch.autograd.grad((do * cd.to(dtype)).sum() + (so * cs.to(dtype)).sum(), [xo["axis"], xo["lamb"], xo["weight"]])

    go, go32 = oracle(torch.float64), oracle(torch.float32)
    for k, a, b, b32 in zip(("axis", "lamb", "weight"), g, go, go32):
        assert rel_l2(a, b) <= tol2(rel_l2(b32, b)), (name, k, rel_l2(a, b), rel_l2(b32, b))


@pytest.mark.parametrize("shape", [
    dict(bn=2, imH=14, imW=18, R=7, C=9, K=12, eh=8, ew=16),     # RC not a multiple of 64, q=4
    dict(bn=1, imH=9, imW=11, R=9, C=11, K=3, eh=3, ew=5),       # J=15: scalar env path, K padded to 4
    dict(bn=1, imH=12, imW=20, R=6, C=10, K=24, eh=16, ew=32),   # config-5-like lobes / directions
    dict(bn=1, imH=10, imW=13, R=4, C=5, K=7, eh=4, ew=8),       # non-integer pooling ratio -> pre-pool
    dict(bn=1, imH=8, imW=8, R=8, C=8, K=32, eh=2, ew=4),        # maximum lobes, J=8 < one tile
    # dispatch branches of the separable-table fast kernels (envWidth 16 / 32)
    dict(bn=2, imH=9, imW=13, R=9, C=13, K=12, eh=8, ew=16),     # ratio-1 maps, RC = 117 (ragged last wave)
    dict(bn=1, imH=12, imW=16, R=6, C=8, K=5, eh=8, ew=16),      # K <= 6: 6-lobe register groups
    dict(bn=1, imH=12, imW=16, R=6, C=8, K=7, eh=7, ew=16),      # K padded to 12, odd envHeight
    dict(bn=1, imH=12, imW=16, R=6, C=8, K=24, eh=4, ew=16),     # forward 24 lobes in registers, backward 2 groups
    dict(bn=1, imH=10, imW=12, R=5, C=6, K=12, eh=3, ew=32),     # envWidth 32: table rows walked as two virtual rows
    dict(bn=2, imH=12, imW=16, R=6, C=8, K=32, eh=3, ew=16),     # three groups of 12 lobes (one workgroup each in the backward)
    dict(bn=1, imH=14, imW=18, R=7, C=9, K=17, eh=5, ew=32),     # two lobe groups, the second partly empty, envWidth 32, ragged tiles
    # round 4: SGNum <= 6 runs the packed half-wave kernels with the upper half's lobe slots empty (the scalar kernels are gone)
    dict(bn=1, imH=10, imW=12, R=5, C=6, K=4, eh=4, ew=32),      # envWidth 32, four lobes
    dict(bn=2, imH=7, imW=9, R=7, C=9, K=6, eh=8, ew=16),        # exactly one half's worth of lobes
    dict(bn=1, imH=12, imW=16, R=6, C=8, K=1, eh=8, ew=16),      # a single lobe
])
def test_shapes_vs_oracle(sgr, shape):
    from oracle import sg_oracle as O
    inp = O.synthetic_inputs(shape["bn"], shape["imH"], shape["imW"], shape["R"], shape["C"], shape["K"],
                             shape["eh"], shape["ew"], seed=4242)
    x = {k: inp[k].cuda().requires_grad_(True) for k in NAMES}
    layer = sgr.renderingLayer(imWidth=shape["C"], imHeight=shape["R"], envWidth=shape["ew"], envHeight=shape["eh"])
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    g = torch.Generator().manual_seed(7)
    cts = [torch.randn(env.shape, generator=g), torch.randn(d.shape, generator=g), torch.randn(s.shape, generator=g)]
    r64, r32, e32 = oracle_with_noise(O, inp, cts, shape["eh"], shape["ew"], NAMES, "cuda")
    for k, v in (("env", env), ("diffuse", d), ("spec", s)):
        assert rel_l2(v.detach(), r64[k]) <= tol2(e32[k]), (shape, k, rel_l2(v.detach(), r64[k]), e32[k])
    grads = torch.autograd.grad([env, d, s], [x[k] for k in NAMES], grad_outputs=[c.cuda() for c in cts])
    # The normal gradient is discontinuous at |N|^2 == 1 (two-sided clamp, models.py:467-468): with unit input normals and no
    # pooling, fp32 and fp64 evaluations of |N|^2 land on different sides of the kink pixel by pixel (e32 of that gradient is
    # then O(1)).  The reference semantics are the fp32 ones, which the kernels follow (|N|^2 with torch's own rounding), so
    # there the gradient is compared with the oracle evaluated in fp32; likewise roughness, which sees the same pixels.
    # Round 5: no bare constant any more (rounds 3-4: 2e-3 against the fp32 oracle).  `agree` = the pixels where the fp32 and the fp64
    # evaluation of the clamp take the same branch: there fp64 IS an arbiter and e_kink = the fp32 oracle's own error on those pixels is
    # the yardstick -- the gradient must be within 2 e_kink of fp64 there, and within max(3 e_kink, 1e-4) of the fp32 evaluation over ALL
    # pixels (a kernel that took the other branch on a single pixel would be O(1) off on it).  The same comparison against the REFERENCE's
    # own fp32 and fp64 values, at size, is tests/test_gpu_fullsize.py::test_clamp_kink_at_size_vs_reference_fp32 (fixture g9).
    n_in = inp["normal"]
    pooled = n_in if (shape["imH"], shape["imW"]) == (shape["R"], shape["C"]) else None
    for k, a in zip(NAMES, grads):
        assert torch.isfinite(a).all(), k
        e, noise = rel_l2(a, r64["g_" + k]), e32["g_" + k]
        if k in ("normal", "rough") and noise > 1e-3:
            assert pooled is not None, (shape, k, noise)      # only unpooled unit normals sit on the kink
            nn32 = torch.sum(pooled.float() * pooled.float(), dim=1, keepdim=True)
            nn64 = torch.sum(pooled.double() * pooled.double(), dim=1, keepdim=True)
            agree = ((nn32 <= 1.0) == (nn64 <= 1.0)).cuda().expand_as(a)
            e_kink = rel_l2(r32["g_" + k][agree], r64["g_" + k][agree])
            e_f32 = rel_l2(a, r32["g_" + k])
            # floor 2e-4: the cap the full-size tests hold these two gradients to (tests/test_gpu_fullsize.py), not 1e-4 -- a handful of
            # near-singular pixels carry the norm of the normal gradient on a 9 x 13 grid
            assert e_f32 <= max(3.0 * e_kink, 2e-4), (shape, k, "vs the fp32 evaluation, all pixels (clamp kink: fp64 is no arbiter)", e_f32, e_kink)
            # 3 e_kink, not 2: e_kink is the fp32 ORACLE's error on a couple of hundred pixels -- a proxy for the reference's own, good to a
            # factor ~1.5 (conftest.oracle_with_noise); with the reference's own e_ref at size the bound is 2 e_ref (fixture g9)
            e_agree = rel_l2(a[agree], r64["g_" + k][agree])
            assert e_agree <= max(3.0 * e_kink, 2e-4), (shape, k, "vs fp64 where the clamp branches agree", e_agree, e_kink)
        else:
            assert e <= (2.0 * noise if k in ("normal", "rough") and noise > 5e-5 else tol2(noise)), (shape, k, e, noise)


def test_full_size_properties(sgr):
    """BASELINE config 2 (bn=16, 240x320 -> 120x160, K=12, 8x16): size-independent properties plus an
    oracle check of two of the sixteen images."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K = 16, 240, 320, 120, 160, 12
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=20202)
    x = {k: inp[k].cuda() for k in NAMES}
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    torch.cuda.synchronize()
    assert torch.isfinite(env).all() and torch.isfinite(d).all() and torch.isfinite(s).all()
    # determinism: a second run is bit-identical
    env2, d2, s2 = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    assert torch.equal(env, env2) and torch.equal(d, d2) and torch.equal(s, s2)
    # batch-permutation equivariance, bit-exact (images are independent)
    perm = torch.randperm(bn, generator=torch.Generator().manual_seed(3)).cuda()
    envp, dp, sp = layer.forwardSG(x["albedo"][perm].contiguous(), x["normal"][perm].contiguous(), x["rough"][perm].contiguous(),
                                   x["axis"][perm].contiguous(), x["lamb"][perm].contiguous(), x["weight"][perm].contiguous(), need_env=True)
    assert torch.equal(envp, env[perm]) and torch.equal(dp, d[perm]) and torch.equal(sp, s[perm])
    # the env-less variant runs the same arithmetic; the two-call form evaluates the microfacet terms in
    # world space instead of the local frame (two fp32 evaluations of an ill-conditioned spec term)
    _, d3, s3 = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=False)
    assert rel_l2(d3.cpu(), d.cpu()) < 1e-6 and rel_l2(s3.cpu(), s.cpu()) < 1e-6
    d4, s4 = layer.forwardEnv(x["albedo"], x["normal"], x["rough"], env)
    assert rel_l2(d4.cpu(), d.cpu()) < 1e-5 and rel_l2(s4.cpu(), s.cpu()) < 2e-4
    # linearity in the (post-tan) weights, exact for a power-of-two scale
    o2e = sgr.output2env(K)
    lam_t = torch.tan(np.pi / 2 * (0.999 * x["lamb"]))
    w_t = torch.tan(np.pi / 2 * (0.999 * x["weight"]))
    e1 = o2e.fromSGtoIm(x["axis"], lam_t, w_t)
    e2 = o2e.fromSGtoIm(x["axis"], lam_t, 2.0 * w_t)
    assert torch.equal(e2, 2.0 * e1)
    # (all sixteen images, whole, against the fp64 oracle: tests/test_gpu_fullsize.py)


def test_tan_handoff_matches_recompute(sgr, monkeypatch):
    """premap mode 2 of the backward entry points (include/sgrender.h): the forward returns the post-tan sharpness / intensity,
    the backward reads them and applies the pre-map's chain rule from those values alone -- same gradients as the backward
    that re-evaluates the pre-map (mode 1), for the fused layer, the two-call drop-in sequence (where the reference returns the
    post-tan tensors anyway) and the fused light objective."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 2, 20, 28, 10, 14, 12, 8, 16
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=99)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    o2e = sgr.output2env(K, ew, eh)
    g = torch.Generator().manual_seed(5)
    cts = [torch.randn((bn, 3, R, C, eh, ew), generator=g).cuda(), torch.randn((bn, 3, R, C), generator=g).cuda(), torch.randn((bn, 3, R, C), generator=g).cuda()]
    ind = torch.ones(bn, 1, 1, 1).cuda()

    def run(handoff):
        monkeypatch.setenv("SGR_TAN_HANDOFF", "1" if handoff else "0")
        x = {k: v.cuda() for k, v in inp.items()}
        for k in ("axis", "lamb", "weight"):
            x[k].requires_grad_(True)
        sg = [x[k] for k in ("axis", "lamb", "weight")]
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)
        g_fused = torch.autograd.grad([env, d, s], sg, grad_outputs=cts)
        obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], *sg, x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0)
        g_obj = torch.autograd.grad(obj[0], sg)
        return env.detach(), g_fused, g_obj, obj[0].detach()

    e1, f1, o1, v1 = run(True)
    e0, f0, o0, v0 = run(False)
    assert torch.equal(e1, e0) and torch.equal(v1, v0)
    for a, b in zip(f1 + o1, f0 + o0):
        assert rel_l2(a, b) < 2e-6, rel_l2(a, b)
    # the two-call sequence saves the post-tan tensors it returns (always mode 2): against the fused layer's gradients
    x = {k: v.cuda() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    env, _, lam_t, w_t = o2e.output2env(x["axis"], x["lamb"], x["weight"])
    g2 = torch.autograd.grad([env, lam_t, w_t], [x["axis"], x["lamb"], x["weight"]], grad_outputs=[cts[0], torch.ones_like(lam_t), torch.ones_like(w_t)])
    y = {k: v.cuda() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        y[k].requires_grad_(True)
    env_f, _, _ = layer.forwardSG(y["albedo"], y["normal"], y["rough"], y["axis"], y["lamb"], y["weight"], need_env=True)
    lt, wt = torch.tan(np.pi / 2 * (0.999 * y["lamb"])), torch.tan(np.pi / 2 * (0.999 * y["weight"]))
    g3 = torch.autograd.grad([env_f, lt, wt], [y["axis"], y["lamb"], y["weight"]], grad_outputs=[cts[0], torch.ones_like(lt), torch.ones_like(wt)])
    for a, b in zip(g2, g3):
        assert rel_l2(a, b) < 1e-5, rel_l2(a, b)


def test_error_behaviour(sgr):
    o2e = sgr.output2env(SGNum=12)
    rl = sgr.renderingLayer(imWidth=16, imHeight=12)
    a = torch.zeros(1, 12, 3, 12, 16)
    with pytest.raises(RuntimeError, match="HIP device"):
        o2e.output2env(a, torch.zeros(1, 12, 12, 16), torch.zeros(1, 36, 12, 16))
    with pytest.raises(RuntimeError, match="lamb must be"):
        o2e.output2env(a.cuda(), torch.zeros(1, 11, 12, 16).cuda(), torch.zeros(1, 36, 12, 16).cuda())
    with pytest.raises(RuntimeError, match="does not match"):
        rl.forwardEnv(torch.zeros(1, 3, 24, 32).cuda(), torch.zeros(1, 3, 24, 32).cuda(), torch.zeros(1, 1, 24, 32).cuda(),
                      torch.zeros(1, 3, 10, 16, 8, 16).cuda())
    with pytest.raises(RuntimeError, match="fp32"):
        o2e.output2env(a.cuda().double(), torch.zeros(1, 12, 12, 16).cuda().double(), torch.zeros(1, 36, 12, 16).cuda().double())




def test_zero_sharpness_lobes(sgr):
    """lamb == 0 exactly is what the decoder's clamp produces (models.py:338-340): exp(0 * t) = 1 in every direction.  The
    packed backward folds lam into the axes and divides the sharpness gradient by it again, with a 2^-40 floor standing in for
    zero -- values and gradients of such lobes (and of weight == 0, lamb == 1 lobes) against the fp64 oracle, fused layer and
    fused objective."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 2, 16, 24, 8, 12, 12, 8, 16
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=77)
    inp["lamb"][:, 0] = 0.0
    inp["lamb"][:, 7] = 0.0
    inp["lamb"][:, 3, ::2] = 0.0
    inp["lamb"][:, 5] = 1.0
    inp["weight"][:, 6:9] = 0.0
    x = {k: v.cuda() for k, v in inp.items()}
    xo = {k: v.double() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
        xo[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    eo, do, so = O.render_from_sg(xo["albedo"], xo["normal"], xo["rough"], xo["axis"], xo["lamb"], xo["weight"], eh, ew)
    g = torch.Generator().manual_seed(3)
    ct = [torch.randn(t.shape, generator=g) for t in (env, d, s)]
    gr = torch.autograd.grad([env, d, s], [x[k] for k in ("axis", "lamb", "weight")], grad_outputs=[t.cuda() for t in ct], retain_graph=True)
    go = torch.autograd.grad([eo, do, so], [xo[k] for k in ("axis", "lamb", "weight")], grad_outputs=[t.double() for t in ct], retain_graph=True)
    assert rel_l2(env.detach().cpu(), eo.detach()) < TOL_L2 and rel_l2(d.detach().cpu(), do.detach()) < TOL_L2
    _, _, e32 = oracle_with_noise(O, inp, ct, eh, ew, ("axis", "lamb", "weight"), "cuda")
    for k, a, b in zip(("axis", "lamb", "weight"), gr, go):
        assert torch.isfinite(a).all(), k
        assert rel_l2(a.cpu(), b) <= tol2(e32["g_" + k]), (k, rel_l2(a.cpu(), b), e32["g_" + k])
    # the zero-sharpness lobes on their own: axis gradient exactly zero, sharpness gradient that of the oracle
    assert float(gr[0][:, 0].abs().max()) == 0.0 and float(gr[0][:, 7].abs().max()) == 0.0
    assert rel_l2(gr[1][:, [0, 7]].cpu(), go[1][:, [0, 7]]) <= tol2(e32["g_lamb"])
    # the same with cotangents the size a normalised training loss produces (1e-9): the lp floor must keep the folded
    # sharpness sums clear of the denormal range
    tiny = 1e-9
    gr_t = torch.autograd.grad([env, d, s], [x[k] for k in ("axis", "lamb", "weight")], grad_outputs=[(t * tiny).cuda() for t in ct], retain_graph=True)
    for k, a, b in zip(("axis", "lamb", "weight"), gr_t, go):
        assert rel_l2(a.cpu().double() / tiny, b) <= tol2(e32["g_" + k]), (k, "1e-9 cotangents", rel_l2(a.cpu().double() / tiny, b))
    assert rel_l2(gr_t[1][:, [0, 7]].cpu().double() / tiny, go[1][:, [0, 7]]) <= tol2(e32["g_lamb"])
    ind = torch.ones(bn, 1, 1, 1)
    obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"],
                              x["env_gt"], ind.cuda(), 1.0, 10.0)
    g2 = torch.autograd.grad(obj[0], [x[k] for k in ("axis", "lamb", "weight")])
    ro, _, _, _ = O.render_loss(do, so, xo["im"], xo["seg"], R, C)
    co, _, _, _ = O.recon_loss(eo, xo["env_gt"], xo["seg"], ind.double(), R, C)
    g3 = torch.autograd.grad(ro + 10.0 * co, [xo[k] for k in ("axis", "lamb", "weight")])
    x32 = {k: v.clone() for k, v in inp.items()}      # the same objective through the oracle in fp32: the yardstick
    for k in ("axis", "lamb", "weight"):
        x32[k].requires_grad_(True)
    e32_, d32_, s32_ = O.render_from_sg(x32["albedo"], x32["normal"], x32["rough"], x32["axis"], x32["lamb"], x32["weight"], eh, ew)
    g3_32 = torch.autograd.grad(O.render_loss(d32_, s32_, x32["im"], x32["seg"], R, C)[0] + 10.0 * O.recon_loss(e32_, x32["env_gt"], x32["seg"], ind, R, C)[0],
                                [x32[k] for k in ("axis", "lamb", "weight")])
    for k, a, b, b32 in zip(("axis", "lamb", "weight"), g2, g3, g3_32):
        assert torch.isfinite(a).all(), k
        assert rel_l2(a.cpu(), b) <= tol2(rel_l2(b32, b)), (k, rel_l2(a.cpu(), b), rel_l2(b32, b))
    assert float(g2[0][:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("lo", [0.9, 0.99], ids=["lam_6_to_636", "lam_64_to_636"])
def test_sharp_lobes_sharpness_gradient(sgr, lo):
    """Round 6: the backward kernels form dL/dlam as  a . S - w . q  in the epilogue (csrc/sgr_pk.inl: sharpness_grad) instead of summing
    T t per direction -- a difference that cancels by ~1 / mean|a . l - 1|, i.e. worst for the sharpest lobes.  Here EVERY lobe is sharp
    (decoder output in [lo, 1): lam = tan(pi/2 0.999 x) from 6.3 or 64 up to the pre-map's maximum 636), fused layer and fused objective,
    against the fp64 oracle at the usual bound max(2 e_ref, 1e-4) with e_ref = the fp32 oracle's own error on these inputs."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 2, 24, 32, 12, 16, 12, 8, 16
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=606)
    g0 = torch.Generator().manual_seed(17)
    inp["lamb"] = lo + (0.99999 - lo) * torch.rand(inp["lamb"].shape, generator=g0)
    x = {k: v.cuda() for k, v in inp.items()}
    xo = {k: v.double() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
        xo[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    eo, do, so = O.render_from_sg(xo["albedo"], xo["normal"], xo["rough"], xo["axis"], xo["lamb"], xo["weight"], eh, ew)
    g = torch.Generator().manual_seed(3)
    ct = [torch.randn(t.shape, generator=g) for t in (env, d, s)]
    gr = torch.autograd.grad([env, d, s], [x[k] for k in ("axis", "lamb", "weight")], grad_outputs=[t.cuda() for t in ct])
    go = torch.autograd.grad([eo, do, so], [xo[k] for k in ("axis", "lamb", "weight")], grad_outputs=[t.double() for t in ct], retain_graph=True)
    _, _, e32 = oracle_with_noise(O, inp, ct, eh, ew, ("axis", "lamb", "weight"), "cuda")
    for k, a, b in zip(("axis", "lamb", "weight"), gr, go):
        assert torch.isfinite(a).all(), k
        assert rel_l2(a.cpu(), b) <= tol2(e32["g_" + k]), ("layer", k, rel_l2(a.cpu(), b), e32["g_" + k])
    print(f"sharp lobes (x >= {lo}): layer sharpness-gradient error {rel_l2(gr[1].cpu(), go[1]):.2e} (fp32 oracle's own {e32['g_lamb']:.2e})", end=" ")
    ind = torch.ones(bn, 1, 1, 1)
    obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"],
                              x["env_gt"], ind.cuda(), 1.0, 10.0)
    g2 = torch.autograd.grad(obj[0], [x[k] for k in ("axis", "lamb", "weight")])
    ro, _, _, _ = O.render_loss(do, so, xo["im"], xo["seg"], R, C)
    co, _, _, _ = O.recon_loss(eo, xo["env_gt"], xo["seg"], ind.double(), R, C)
    g3 = torch.autograd.grad(ro + 10.0 * co, [xo[k] for k in ("axis", "lamb", "weight")])
    x32 = {k: v.clone() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x32[k].requires_grad_(True)
    e32_, d32_, s32_ = O.render_from_sg(x32["albedo"], x32["normal"], x32["rough"], x32["axis"], x32["lamb"], x32["weight"], eh, ew)
    g3_32 = torch.autograd.grad(O.render_loss(d32_, s32_, x32["im"], x32["seg"], R, C)[0] + 10.0 * O.recon_loss(e32_, x32["env_gt"], x32["seg"], ind, R, C)[0],
                                [x32[k] for k in ("axis", "lamb", "weight")])
    for k, a, b, b32 in zip(("axis", "lamb", "weight"), g2, g3, g3_32):
        assert torch.isfinite(a).all(), k
        assert rel_l2(a.cpu(), b) <= tol2(rel_l2(b32, b)), ("objective", k, rel_l2(a.cpu(), b), rel_l2(b32, b))
    print(f"objective {rel_l2(g2[1].cpu(), g3[1]):.2e} ({rel_l2(g3_32[1], g3[1]):.2e})")
