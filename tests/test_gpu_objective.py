"""GPU: the fused trainLight objective (sgr_fused_fwd_recon / sgr_fused_bwd_recon, env image never written)
against the golden fixtures (reference values and gradients), the fp64 oracle and the unfused HIP path."""
import numpy as np
import pytest
import torch

from conftest import layer_kwargs, rel_l2, scalar_close

pytestmark = pytest.mark.gpu

NAMES = ("albedo", "normal", "rough", "axis", "lamb", "weight")


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _t(z, k):
    return torch.from_numpy(np.ascontiguousarray(z[k])).cuda()


def _oracle_objective(inp, ind, R, C, eh, ew, fov, F0, ren_w, rec_w, offset=1.0, dtype=torch.float64):
    """fp64 (or fp32: the yardstick) restatement of wrapperBRDFLight.py:167-207 from the oracle's pieces; returns values and SG gradients."""
    from oracle import sg_oracle as O
    x = {k: v.to(dtype) for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x[k] = x[k].clone().requires_grad_(True)
    env, d, s = O.render_from_sg(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew, fov, F0)
    rerr, _, _, _ = O.render_loss(d, s, x["im"], x["seg"], R, C)
    cerr, _, _, _ = O.recon_loss(env, x["env_gt"], x["seg"], ind.to(dtype), R, C, offset)
    tot = ren_w * rerr + rec_w * cerr
    grads = torch.autograd.grad(tot, [x["axis"], x["lamb"], x["weight"]])
    return tot.item(), rerr.item(), cerr.item(), grads


def test_light_objective_vs_golden(sgr, golden):
    """renderErr + 10 reconstErr (trainLight.py:47-48,237) and its SG gradients against the reference's."""
    name, z, cfg = golden
    R, C = cfg["R"], cfg["C"]
    x = {k: _t(z, "in_" + k) for k in NAMES}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    ind = torch.ones(cfg["bn"], 1, 1, 1, device="cuda")
    args = (layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], _t(z, "in_im"), _t(z, "in_seg"),
            _t(z, "in_env_gt"), ind)
    # g2 (4x8 directions) has no fused kernel: light_objective evaluates it with the unfused HIP kernels
    assert sgr.light_objective_supported(cfg["K"], R, C, cfg["eh"], cfg["ew"]) == (cfg["ew"] in (16, 32) and cfg["K"] <= 24)
    obj, rerr, cerr, ren, coef = sgr.light_objective(*args, 1.0, 10.0)
    # reported values: against the reference's fp64 evaluation, within twice the reference's own fp32 error on that value
    # (floored at 1e-5 RELATIVE -- measured: ~1e-6)
    r_ref, c_ref = float(z["ref32_render_err"][0]), float(z["ref32_recon_err"][0])
    r64, c64 = float(z["ref64_render_err"][0]), float(z["ref64_recon_err"][0])
    assert scalar_close(rerr.item(), r64, r_ref - r64), (name, rerr.item(), r64, r_ref)
    assert scalar_close(cerr.item(), c64, c_ref - c64), (name, cerr.item(), c64, c_ref)
    assert scalar_close(obj.item(), r64 + 10.0 * c64, abs(r_ref - r64) + 10.0 * abs(c_ref - c64)), (name, obj.item())
    assert rel_l2(ren.cpu(), z["ref32_rendered"]) < 1e-4, name
    grads = torch.autograd.grad(obj, [x["axis"], x["lamb"], x["weight"]])
    for k, g in zip(("axis", "lamb", "weight"), grads):
        ref32, ref64 = z["ref32_gtot_" + k], z["ref64_gtot_" + k]
        e_ref = rel_l2(ref32, ref64)
        assert rel_l2(g.cpu(), ref64) < max(2 * e_ref, 1e-4), (name, k, rel_l2(g.cpu(), ref64), e_ref)


@pytest.mark.parametrize("bn,imH,imW,R,C,K,benign,eh,ew", [
    (2, 12, 20, 12, 20, 12, True, 8, 16),      # q = 1, one partial 64-pixel tile per image
    (3, 18, 26, 9, 13, 12, False, 8, 16),      # q = 4, odd grid, decoder-range (stress) lobes
    (2, 10, 14, 10, 14, 5, True, 8, 16),       # SGNum <= 6: the second wave of a workgroup owns no lobes
    (1, 16, 40, 8, 20, 9, True, 8, 16),        # SGNum between 6 and 12
    # round 3: the 16x32 grid (table rows as two virtual rows) and up to 24 lobes (four lane groups per pixel in the backward)
    (2, 12, 20, 6, 10, 12, True, 16, 32),      # config-5 grid, 12 lobes: half-wave statistics kernel, NG = 2 backward on virtual rows
    (2, 18, 26, 9, 13, 24, False, 16, 32),     # config 5's parameters: 24 lobes, 16x32, ragged 16-pixel tiles, stress lobes
    (1, 14, 22, 7, 11, 24, True, 8, 16),       # 24 lobes on the 8x16 grid
    (2, 10, 14, 10, 14, 17, True, 5, 32),      # a partly empty fourth lobe group, odd envHeight, q = 1
    (1, 12, 16, 6, 8, 7, True, 3, 32),         # 16x32-style grid with few lobes
])
def test_light_objective_vs_oracle(sgr, bn, imH, imW, R, C, K, benign, eh, ew):
    """Values and gradients against the fp64 oracle, with env_ind == 0 images, dark ground-truth cells and
    unequal loss weights; the unfused HIP path (forwardSG + render_loss + recon_loss) must agree too."""
    from oracle import sg_oracle as O
    fov, F0 = 57.0, 0.05
    assert sgr.light_objective_supported(K, R, C, eh, ew)
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=77 + K, benign=benign)
    inp["env_gt"][0, :, 1:3, 2:6] = 0.0                      # dark cells drop out of the env mask
    ind = torch.ones(bn, 1, 1, 1)
    if bn > 1:
        ind[1] = 0.0
    ren_w, rec_w, offset = 0.7, 3.0, 1.0
    tot_o, rerr_o, cerr_o, g_o = _oracle_objective(inp, ind, R, C, eh, ew, fov, F0, ren_w, rec_w, offset)
    _, rerr_32, cerr_32, _ = _oracle_objective(inp, ind, R, C, eh, ew, fov, F0, ren_w, rec_w, offset, torch.float32)
    n32 = (abs(rerr_32 - rerr_o), abs(cerr_32 - cerr_o))      # the fp32 oracle's own error on the two reported values

    dev = {k: v.cuda() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        dev[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, fov=fov, F0=F0, envWidth=ew, envHeight=eh)
    obj, rerr, cerr, ren, coef = sgr.light_objective(layer, dev["albedo"], dev["normal"], dev["rough"], dev["axis"], dev["lamb"],
                                                      dev["weight"], dev["im"], dev["seg"], dev["env_gt"], ind.cuda(), ren_w, rec_w,
                                                      offset)
    assert scalar_close(rerr.item(), rerr_o, n32[0]), (rerr.item(), rerr_o, n32)
    assert scalar_close(cerr.item(), cerr_o, n32[1]), (cerr.item(), cerr_o, n32)
    assert scalar_close(obj.item(), tot_o, ren_w * n32[0] + rec_w * n32[1]), (obj.item(), tot_o)
    grads = torch.autograd.grad(2.0 * obj, [dev["axis"], dev["lamb"], dev["weight"]])     # cotangent != 1
    for k, g, go in zip(("axis", "lamb", "weight"), grads, g_o):
        assert rel_l2(g.cpu(), 2.0 * go) < 2e-4, (k, rel_l2(g.cpu(), 2.0 * go))

    # unfused HIP path on the same inputs
    env, d, s = layer.forwardSG(dev["albedo"], dev["normal"], dev["rough"], dev["axis"], dev["lamb"], dev["weight"], need_env=True)
    r2, _ = sgr.render_loss(d, s, dev["im"], dev["seg"], R, C)
    c2 = sgr.recon_loss(env, dev["env_gt"], dev["seg"], ind.cuda(), R, C, offset)
    g2 = torch.autograd.grad(2.0 * (ren_w * r2 + rec_w * c2), [dev["axis"], dev["lamb"], dev["weight"]])
    assert scalar_close(r2.item(), rerr.item(), 0.0, 2e-5) and scalar_close(c2.item(), cerr.item(), 0.0, 2e-5)      # fused vs unfused HIP: both fp32
    for k, ga, gb in zip(("axis", "lamb", "weight"), grads, g2):
        assert rel_l2(ga, gb) < 1e-4, (k, rel_l2(ga, gb))


def test_light_objective_full_size_matches_unfused(sgr):
    """BASELINE config 2 shapes: the fused objective equals the unfused HIP pipeline (value and gradients) and is
    bit-reproducible (no atomics)."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 16, 240, 320, 120, 160, 12, 8, 16
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=20202)
    dev = {k: v.cuda() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        dev[k].requires_grad_(True)
    ind = torch.ones(bn, 1, 1, 1, device="cuda")
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, fov=57, F0=0.05, envWidth=ew, envHeight=eh)

    def fused():
        out = sgr.light_objective(layer, dev["albedo"], dev["normal"], dev["rough"], dev["axis"], dev["lamb"], dev["weight"],
                                  dev["im"], dev["seg"], dev["env_gt"], ind, 1.0, 10.0)
        return out, torch.autograd.grad(out[0], [dev["axis"], dev["lamb"], dev["weight"]])
    (o1, g1), (o2, g2) = fused(), fused()
    assert torch.equal(o1[0], o2[0]) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    env, d, s = layer.forwardSG(dev["albedo"], dev["normal"], dev["rough"], dev["axis"], dev["lamb"], dev["weight"], need_env=True)
    r2, _ = sgr.render_loss(d, s, dev["im"], dev["seg"], R, C)
    c2 = sgr.recon_loss(env, dev["env_gt"], dev["seg"], ind, R, C)
    g3 = torch.autograd.grad(r2 + 10.0 * c2, [dev["axis"], dev["lamb"], dev["weight"]])
    assert scalar_close(o1[1].item(), r2.item(), 0.0, 2e-5)
    assert scalar_close(o1[2].item(), c2.item(), 0.0, 2e-5)
    for k, ga, gb in zip(("axis", "lamb", "weight"), g1, g3):
        assert rel_l2(ga, gb) < 1e-4, (k, rel_l2(ga, gb))
    for g in g1:
        assert torch.isfinite(g).all()


def test_light_objective_cotangent_scaling_and_second_backward(sgr):
    """The gradients exist before backward is called: a cotangent != 1 rescales them on the device, and a second
    backward through the same node (retain_graph) returns fresh tensors."""
    from oracle import sg_oracle as O
    inp = {k: v.cuda() for k, v in O.synthetic_inputs(2, 16, 32, 8, 16, 12, 8, 16, seed=11, benign=True).items()}
    for k in ("axis", "lamb", "weight"):
        inp[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=16, imHeight=8, envWidth=16, envHeight=8)
    ind = torch.ones(2, 1, 1, 1, device="cuda")

    def objective():
        return sgr.light_objective(layer, inp["albedo"], inp["normal"], inp["rough"], inp["axis"], inp["lamb"], inp["weight"],
                                   inp["im"], inp["seg"], inp["env_gt"], ind)[0]
    leaves = [inp["axis"], inp["lamb"], inp["weight"]]
    g1 = torch.autograd.grad(objective(), leaves)
    obj = objective()
    g3 = torch.autograd.grad(3.0 * obj, leaves, retain_graph=True)
    g3 = [g.clone() for g in g3]
    g5 = torch.autograd.grad(5.0 * obj, leaves)
    for a, b, c in zip(g1, g3, g5):
        assert torch.allclose(3.0 * a, b, rtol=1e-6, atol=0) and torch.allclose(5.0 * a, c, rtol=1e-6, atol=0)


def test_light_objective_rejects_brdf_gradients(sgr):
    from oracle import sg_oracle as O
    inp = {k: v.cuda() for k, v in O.synthetic_inputs(1, 8, 16, 8, 16, 12, 8, 16, seed=3, benign=True).items()}
    inp["rough"].requires_grad_(True)
    inp["axis"].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=16, imHeight=8, envWidth=16, envHeight=8)
    # refused at the call (round 4; the Python node of rounds 2-3 only noticed in backward): a grad-requiring BRDF map would
    # silently get no gradient otherwise
    with pytest.raises(RuntimeError, match="SG parameters only"):
        sgr.light_objective(layer, inp["albedo"], inp["normal"], inp["rough"], inp["axis"], inp["lamb"], inp["weight"],
                            inp["im"], inp["seg"], inp["env_gt"], torch.ones(1, 1, 1, 1, device="cuda"))
    with torch.no_grad():      # forward-only evaluation does not care
        sgr.light_objective(layer, inp["albedo"], inp["normal"], inp["rough"], inp["axis"], inp["lamb"], inp["weight"],
                            inp["im"], inp["seg"], inp["env_gt"], torch.ones(1, 1, 1, 1, device="cuda"))


# --------------------------------------------------------------------------- #
# decoder heads as the kernels' prologue (SURVEY.md section 8f rank 2 as written; premap = 3)
# --------------------------------------------------------------------------- #
def _decoder_outputs(bn, K, R, C, seed):
    """Last-convolution outputs of the three light decoders: N(0, 1.2) with saturated entries either side (the 1.01 tanh
    then leaves [-1, 1] and both ends of the [0, 1] clamp are live)."""
    g = torch.Generator().manual_seed(seed)
    xa = 1.2 * torch.randn(bn, 3 * K, R, C, generator=g)
    xl = 1.2 * torch.randn(bn, K, R, C, generator=g)
    xw = 1.2 * torch.randn(bn, 3 * K, R, C, generator=g)
    for x in (xa, xl, xw):
        flat = x.view(-1)
        idx = torch.randperm(flat.numel(), generator=g)[: max(4, flat.numel() // 50)]
        flat[idx] = torch.where(torch.rand(idx.numel(), generator=g) < 0.5, 6.0, -6.0) + 0.3 * torch.randn(idx.numel(), generator=g)
    return xa, xl, xw


@pytest.mark.parametrize("bn,imH,imW,R,C,K,eh,ew", [
    (2, 12, 20, 12, 20, 12, 8, 16),            # config 2's kernels, q = 1, partial tile
    (2, 18, 26, 9, 13, 12, 8, 16),             # q = 4
    (1, 16, 40, 8, 20, 9, 8, 16),              # lobe slots past K in the second half-wave
    (2, 18, 26, 9, 13, 24, 16, 32),            # config 5's kernels (12 lobes per half-wave forward, four lane groups backward)
    (1, 12, 16, 6, 8, 7, 3, 32),
    (2, 10, 14, 10, 14, 5, 8, 16),             # no prologue for SGNum <= 6: light_heads + the plain objective, same answers
])
def test_light_objective_from_decoder_outputs(sgr, bn, imH, imW, R, C, K, eh, ew):
    """``light_objective(decoder_outputs=True)``: values and gradients w.r.t. the decoders' last-convolution outputs against
    the fp64 oracle (heads of models.py:336-346 + wrapperBRDFLight.py:167-207) within twice the fp32 oracle's own noise, and
    against the two-step HIP route (standalone heads pass, then the objective)."""
    from conftest import tol2
    from oracle import sg_oracle as O
    fov, F0, ren_w, rec_w = 57.0, 0.05, 1.0, 10.0
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=300 + K, benign=True)
    xs = _decoder_outputs(bn, K, R, C, seed=900 + K)
    ind = torch.ones(bn, 1, 1, 1)

    def oracle(dtype):
        x = [t.to(dtype).clone().requires_grad_(True) for t in xs]
        a, l, w, _ = O.light_heads(*x)
        o = {k: v.to(dtype) for k, v in inp.items()}
        env, d, s = O.render_from_sg(o["albedo"], o["normal"], o["rough"], a, l, w, eh, ew, fov, F0)
        rerr = O.render_loss(d, s, o["im"], o["seg"], R, C)[0]
        cerr = O.recon_loss(env, o["env_gt"], o["seg"], ind.to(dtype), R, C, 1.0)[0]
        tot = ren_w * rerr + rec_w * cerr
        return tot.item(), rerr.item(), cerr.item(), torch.autograd.grad(tot, x)
    tot64, r64, c64, g64 = oracle(torch.float64)
    tot32, r32, c32, g32 = oracle(torch.float32)

    dev = {k: v.cuda() for k, v in inp.items()}
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, fov=fov, F0=F0, envWidth=ew, envHeight=eh)
    supported = bool(sgr._lib.load().sgr_heads_prologue_supported(K, R, C, eh, ew))
    assert supported == (K > 6)

    def run(prologue):
        x = [t.cuda().requires_grad_(True) for t in xs]
        if prologue:
            out = sgr.light_objective(layer, dev["albedo"], dev["normal"], dev["rough"], x[0], x[1], x[2], dev["im"], dev["seg"],
                                      dev["env_gt"], ind.cuda(), ren_w, rec_w, decoder_outputs=True)
        else:
            a, l, w, _ = sgr.light_heads(*x)
            out = sgr.light_objective(layer, dev["albedo"], dev["normal"], dev["rough"], a, l, w, dev["im"], dev["seg"], dev["env_gt"],
                                      ind.cuda(), ren_w, rec_w)
        return out, torch.autograd.grad(out[0], x)
    (o_p, g_p), (o_s, g_s) = run(True), run(False)
    assert scalar_close(o_p[1].item(), r64, r32 - r64) and scalar_close(o_p[2].item(), c64, c32 - c64), (o_p[1].item(), r64, r32, o_p[2].item(), c64, c32)
    assert scalar_close(o_p[0].item(), tot64, tot32 - tot64), (o_p[0].item(), tot64, tot32)
    for name, gp, gs, a64, a32 in zip(("x_axis", "x_lamb", "x_weight"), g_p, g_s, g64, g32):
        e_ref = rel_l2(a32, a64)
        assert rel_l2(gp.cpu(), a64) < tol2(e_ref), (name, rel_l2(gp.cpu(), a64), e_ref)
        # the two HIP routes differ by the rounding of tanh alone, which the tan pre-map amplifies near w = 1 like any fp32 noise
        assert rel_l2(gp, gs) < tol2(e_ref), (name, rel_l2(gp, gs), e_ref)
        assert torch.isfinite(gp).all()


def test_light_objective_from_decoder_outputs_full_size(sgr):
    """BASELINE config 2 shapes (4 images): the prologue route equals the standalone-heads route, and is bit-reproducible."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 4, 240, 320, 120, 160, 12, 8, 16
    inp = {k: v.cuda() for k, v in O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=20203).items()}
    xs = [t.cuda() for t in _decoder_outputs(bn, K, R, C, seed=41)]
    ind = torch.ones(bn, 1, 1, 1, device="cuda")
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, fov=57, F0=0.05, envWidth=ew, envHeight=eh)

    def run(prologue):
        x = [t.clone().requires_grad_(True) for t in xs]
        if prologue:
            out = sgr.light_objective(layer, inp["albedo"], inp["normal"], inp["rough"], x[0], x[1], x[2], inp["im"], inp["seg"],
                                      inp["env_gt"], ind, 1.0, 10.0, decoder_outputs=True)
        else:
            a, l, w, _ = sgr.light_heads(*x)
            out = sgr.light_objective(layer, inp["albedo"], inp["normal"], inp["rough"], a, l, w, inp["im"], inp["seg"], inp["env_gt"], ind,
                                      1.0, 10.0)
        return out, torch.autograd.grad(out[0], x)
    (o1, g1), (o2, g2), (o3, g3) = run(True), run(True), run(False)
    assert torch.equal(o1[0], o2[0]) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert scalar_close(o1[1].item(), o3[1].item(), 0.0, 2e-5) and scalar_close(o1[2].item(), o3[2].item(), 0.0, 2e-5)
    for name, ga, gb in zip(("x_axis", "x_lamb", "x_weight"), g1, g3):
        assert rel_l2(ga, gb) < 1e-4, (name, rel_l2(ga, gb))


def test_forward_only_objective_launches_no_gradient_kernel(sgr):
    """``light_objective`` under ``torch.no_grad()`` (the evaluation loops: testLight.py drives wrapperBRDFLight.py:167-207 without a
    backward) and with no grad-requiring SG input: the same five values as the grad-mode call, bit for bit for the render terms and to
    fp32 summation noise for the reconstruction term, from a pass that launches no gradient kernel (the kernel names of the two calls are read back
    from the profiler; since ABI 5 the grad-mode call has no separate render-loss backward launch either)."""
    from oracle import sg_oracle as O
    from torch.profiler import ProfilerActivity, profile
    bn, imH, imW, R, C, K = 2, 24, 32, 12, 16, 12
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=91)
    dev = {k: v.cuda() for k, v in inp.items()}
    ind = torch.ones(bn, 1, 1, 1, device="cuda")
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)

    def call(grad):
        x = [dev[k].clone().requires_grad_(grad) for k in ("axis", "lamb", "weight")]
        return sgr.light_objective(layer, dev["albedo"], dev["normal"], dev["rough"], x[0], x[1], x[2], dev["im"], dev["seg"], dev["env_gt"], ind, 1.0, 10.0)

    def kernels(fn):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            out = fn()
            torch.cuda.synchronize()
        return out, [e.key for e in prof.key_averages() if "sgr::" in e.key]

    ref = call(True)
    assert ref[0].requires_grad
    call(False)      # warm-up of the loss-only kernel
    (o_ng, names_ng) = kernels(lambda: call(False))
    with torch.no_grad():
        (o_ng2, names_ng2) = kernels(lambda: call(True))
    (o_g, names_g) = kernels(lambda: call(True))
    for o in (o_ng, o_ng2):
        assert not o[0].requires_grad and o[0].grad_fn is None
        assert torch.equal(o[1], ref[1]) and torch.equal(o[3], ref[3]) and torch.equal(o[4], ref[4])      # render terms: the same kernels
        assert scalar_close(o[2].item(), ref[2].item(), 0.0, 2e-6) and scalar_close(o[0].item(), ref[0].item(), 0.0, 2e-6)
    if names_g:      # the profiler reports device kernels on this box
        grad_kernel = [n for n in names_g if "sg_bwd_recon_pk_kernel" in n]
        assert grad_kernel, names_g
        assert not any("loss_bwd" in n for n in names_g), names_g      # ABI 5: the render-loss gradient comes out of the loss's third pass
        for names in (names_ng, names_ng2):
            assert names, "no device kernels recorded for the forward-only call"
            assert not any("loss_bwd" in n for n in names), names
            lossonly = [n for n in names if "sg_bwd_recon_pk_kernel" in n]
            assert lossonly and all(n not in grad_kernel for n in lossonly), (lossonly, grad_kernel)      # the GRADS = false instantiation


@pytest.mark.parametrize("bn,imH,imW,R,C,K,eh,ew,premap", [
    (16, 240, 320, 120, 160, 12, 8, 16, 1),      # config 2: one pixel per lane statistics kernel, 600 (64-pixel) partials per image
    (2, 24, 32, 12, 16, 12, 8, 16, 3),           # decoder heads as the prologue: half-wave statistics kernel
    (3, 18, 26, 9, 13, 9, 8, 16, 1),             # ragged tiles
    (2, 12, 20, 6, 10, 24, 16, 32, 1),           # config-5 grid, 24 lobes
])
def test_objective_forward_half_in_four_launches_equals_the_separate_calls(sgr, bn, imH, imW, R, C, K, eh, ew, premap):
    """ABI 5: sgr_light_objective_fwd (statistics kernel + the render loss's three passes, the first folding the env statistics per image
    as an extra workgroup, the third writing the render-loss gradient) against the calls it replaces -- sgr_fused_fwd_recon_seg (own fold
    launch) + sgr_render_loss_fwd_total_grads: every output bit-identical, incl. the per-image mask sums left in the workspace."""
    from oracle import sg_oracle as O
    from inverserenderingofindoorscene_amd import _lib
    from inverserenderingofindoorscene_amd.ops import _dirs, _ptr, _stream, _view
    lib = _lib.load()
    dev = torch.device("cuda")
    x = {k: v.to(dev) for k, v in O.synthetic_inputs(bn, imH, imW, R, C, K, eh=eh, ew=ew, seed=77 + bn).items()}
    ind = torch.ones(bn, device=dev)
    dirs, view = _dirs(dev, eh, ew), _view(dev, R, C, 57.0)
    st = _stream(dev)
    ren_w = 0.8

    def buffers():
        e = lambda *sh: torch.empty(*sh, device=dev)
        return dict(diffuse=e(bn, 3, R, C), spec=e(bn, 3, R, C), mask=e(bn, R * C), coef_env=e(bn), im_s=e(bn, 3, R, C), seg_s=e(bn, 1, R, C),
                    rendered=e(bn, 3, R, C), coef_ds=e(bn, 2), parts=e(2), loss=e(1), scale=e(1), gd=e(bn, 3, R, C), gs=e(bn, 3, R, C),
                    ws=torch.zeros(lib.sgr_fused_recon_workspace_floats(bn, R, C), device=dev), wsl=e(lib.sgr_loss_workspace_floats(bn)))

    a, b = buffers(), buffers()
    sg = [_ptr(x[k]) for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")]
    rc = lib.sgr_fused_fwd_recon_seg(*sg, _ptr(dirs), _ptr(view), _ptr(x["env_gt"]), _ptr(x["seg"]), imH, imW, _ptr(ind), None, None, _ptr(a["diffuse"]),
                                     _ptr(a["spec"]), _ptr(a["mask"]), _ptr(a["coef_env"]), None, _ptr(a["ws"]), bn, K, R, C, eh, ew, imH, imW, 0.05, premap, st)
    assert rc == 0, lib.sgr_last_error()
    rc = lib.sgr_render_loss_fwd_total_grads(_ptr(a["diffuse"]), _ptr(a["spec"]), _ptr(x["im"]), _ptr(x["seg"]), _ptr(a["im_s"]), _ptr(a["seg_s"]), _ptr(a["rendered"]),
                                             _ptr(a["coef_ds"]), _ptr(a["parts"]), _ptr(a["loss"]), _ptr(a["scale"]), 3.0, ren_w, _ptr(a["gd"]), _ptr(a["gs"]),
                                             _ptr(a["wsl"]), bn, R, C, imH, imW, st)
    assert rc == 0, lib.sgr_last_error()
    rc = lib.sgr_light_objective_fwd(*sg, _ptr(dirs), _ptr(view), _ptr(x["env_gt"]), _ptr(x["im"]), _ptr(x["seg"]), _ptr(ind), None, None, _ptr(b["diffuse"]),
                                     _ptr(b["spec"]), _ptr(b["mask"]), _ptr(b["coef_env"]), _ptr(b["im_s"]), _ptr(b["seg_s"]), _ptr(b["rendered"]), _ptr(b["coef_ds"]),
                                     _ptr(b["parts"]), _ptr(b["loss"]), _ptr(b["scale"]), ren_w, _ptr(b["gd"]), _ptr(b["gs"]), _ptr(b["ws"]), _ptr(b["wsl"]),
                                     bn, K, R, C, eh, ew, imH, imW, imH, imW, 0.05, premap, st)
    assert rc == 0, lib.sgr_last_error()
    torch.cuda.synchronize()
    for k in ("diffuse", "spec", "mask", "coef_env", "im_s", "seg_s", "rendered", "coef_ds", "parts", "loss", "scale", "gd", "gs"):
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
    assert torch.equal(a["ws"][:bn], b["ws"][:bn])      # the per-image mask sums the backward entry point reads
    assert torch.isfinite(b["coef_env"]).all() and b["ws"][:bn].min().item() > 0
