"""GPU: scale-invariant regressions, render loss and the trainLight-style objective through the
C ABI, against the golden fixtures (reference outputs) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import layer_kwargs, rel_l2, rel_max, scalar_close

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


def test_lsregress_functions_vs_golden(sgr, golden):
    name, z, cfg = golden
    R, C = cfg["R"], cfg["C"]
    d, s = _t(z, "ref32_diffuse"), _t(z, "ref32_spec")
    im_s = torch.nn.functional.adaptive_avg_pool2d(_t(z, "in_im"), (R, C))
    ds, ss = sgr.LSregressDiffSpec(d, s, im_s, d, s)
    assert rel_l2(ds.cpu(), z["ref32_diff_scaled"]) < 1e-5, name
    assert rel_l2(ss.cpu(), z["ref32_spec_scaled"]) < 1e-5 or float(np.abs(z["ref32_spec_scaled"]).max()) == 0.0, name
    # one-unknown regression on env-sized tensors (the call at wrapperBRDFLight.py:180-181)
    env, env_gt = _t(z, "ref32_env"), _t(z, "in_env_gt")
    seg_s = torch.nn.functional.adaptive_avg_pool2d(_t(z, "in_seg"), (R, C))
    m = seg_s[..., None, None].expand_as(env_gt)
    sc = sgr.LSregress(env * m, env_gt * m, env)
    assert rel_l2(sc.cpu(), z["ref32_env_scaled"]) < 1e-5, name
    # the reference detaches the coefficient itself (models.py:13): a grad-carrying `pred` is a valid input
    # (trainBRDF.py:249-254 passes albedoPred * seg) and the gradient flows through `origin` only
    pred_live = (env * m).clone().requires_grad_(True)
    origin_live = env.clone().requires_grad_(True)
    sc2 = sgr.LSregress(pred_live, env_gt * m, origin_live)
    assert torch.equal(sc2.detach(), sc)
    ct = torch.randn(sc2.shape, generator=torch.Generator().manual_seed(3)).cuda()
    g_origin, g_pred = torch.autograd.grad((sc2 * ct).sum(), [origin_live, pred_live], allow_unused=True)
    assert g_pred is None or float(g_pred.abs().max()) == 0.0
    nb = env.shape[0]
    pm, gm = (env * m).reshape(nb, -1).double(), (env_gt * m).reshape(nb, -1).double()
    coef = torch.clamp((pm * gm).sum(1) / torch.clamp((pm * pm).sum(1), min=1e-5), 0.001, 1000.0).reshape(nb, 1, 1, 1, 1, 1)
    assert rel_l2(g_origin.cpu(), (ct.double() * coef).cpu()) < 1e-5, name
    # LSregressDiffSpec with live first arguments (trainFineTune*_cascade1.py): the reference differentiates through the
    # coefficients; same values as the kernel path, and the gradient against the oracle's fp64 restatement
    from oracle import sg_oracle as O
    d_live, s_live = d.clone().requires_grad_(True), s.clone().requires_grad_(True)
    ds2, ss2 = sgr.LSregressDiffSpec(d_live, s_live, im_s, d_live, s_live)
    assert rel_l2(ds2.detach().cpu(), z["ref32_diff_scaled"]) < 1e-5, name
    ct2 = torch.randn(ds2.shape, generator=torch.Generator().manual_seed(4)).cuda()
    gl = torch.autograd.grad((ds2 * ct2).sum() + (ss2 * ct2).sum(), [d_live, s_live])
    do_, so_ = d.double().cpu().requires_grad_(True), s.double().cpu().requires_grad_(True)
    dso, sso = O.lsregress_diffspec(do_, so_, im_s.double().cpu(), do_, so_)
    go = torch.autograd.grad((dso * ct2.double().cpu()).sum() + (sso * ct2.double().cpu()).sum(), [do_, so_])
    assert rel_l2(gl[0].cpu(), go[0]) < 1e-4 and rel_l2(gl[1].cpu(), go[1]) < 1e-4, name


def test_render_loss_vs_golden(sgr, golden):
    name, z, cfg = golden
    R, C = cfg["R"], cfg["C"]
    d = _t(z, "ref32_diffuse").requires_grad_(True)
    s = _t(z, "ref32_spec").requires_grad_(True)
    err, ren = sgr.render_loss(d, s, _t(z, "in_im"), _t(z, "in_seg"), R, C)
    assert scalar_close(err.item(), float(z["ref32_render_err"][0]), 0.0, 2e-6), (name, err.item())      # same fp32 inputs, same arithmetic: summation order only
    assert rel_max(ren.cpu(), z["ref32_rendered"]) < 1e-5, name
    gd, gs = torch.autograd.grad(err, [d, s])
    # oracle gradient in fp64 on the same (reference fp32) render outputs
    from oracle import sg_oracle as O
    do = torch.from_numpy(z["ref32_diffuse"]).double().requires_grad_(True)
    so = torch.from_numpy(z["ref32_spec"]).double().requires_grad_(True)
    eo, _, _, _ = O.render_loss(do, so, torch.from_numpy(z["in_im"]).double(), torch.from_numpy(z["in_seg"]).double(), R, C)
    gdo, gso = torch.autograd.grad(eo, [do, so])
    assert rel_l2(gd.cpu(), gdo) < 1e-4, (name, rel_l2(gd.cpu(), gdo))
    assert rel_l2(gs.cpu(), gso) < 1e-4 or float(gso.abs().max()) == 0.0, name


def test_trainlight_objective_grads_vs_golden(sgr, golden):
    """renderErr + 10*reconstErr (trainLight.py:47-48,237): render path and render loss are the
    product; the reconstruction loss (a 'next' row) is evaluated with the oracle's
    torch code on the GPU tensors so the total can be compared with the reference's gradients."""
    from oracle import sg_oracle as O
    name, z, cfg = golden
    R, C = cfg["R"], cfg["C"]
    x = {k: _t(z, "in_" + k) for k in NAMES}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    layer = sgr.renderingLayer(**layer_kwargs(cfg))
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    rerr, ren = sgr.render_loss(d, s, _t(z, "in_im"), _t(z, "in_seg"), R, C)
    ind = torch.ones(cfg["bn"], 1, 1, 1, device="cuda")
    cerr, _, _, _ = O.recon_loss(env, _t(z, "in_env_gt"), _t(z, "in_seg"), ind, R, C)
    r32, c32, r64, c64 = (float(z[k][0]) for k in ("ref32_render_err", "ref32_recon_err", "ref64_render_err", "ref64_recon_err"))
    assert scalar_close(rerr.item(), r64, r32 - r64), (rerr.item(), r64, r32)
    assert scalar_close(cerr.item(), c64, c32 - c64), (cerr.item(), c64, c32)
    grads = torch.autograd.grad(rerr + 10.0 * cerr, [x["axis"], x["lamb"], x["weight"]])
    for k, g in zip(("axis", "lamb", "weight"), grads):
        ref32, ref64 = z["ref32_gtot_" + k], z["ref64_gtot_" + k]
        e_ref = rel_l2(ref32, ref64)
        assert rel_l2(g.cpu(), ref64) < max(3 * e_ref, 1e-4), (name, k, rel_l2(g.cpu(), ref64), e_ref)


def test_render_loss_full_size_and_determinism(sgr):
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K = 16, 240, 320, 120, 160, 12
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=5)
    g = torch.Generator().manual_seed(6)
    d = (torch.rand(bn, 3, R, C, generator=g) * 0.8)
    s = (torch.rand(bn, 3, R, C, generator=g) * 0.3)
    e1, r1 = sgr.render_loss(d.cuda(), s.cuda(), inp["im"].cuda(), inp["seg"].cuda(), R, C)
    e2, r2 = sgr.render_loss(d.cuda(), s.cuda(), inp["im"].cuda(), inp["seg"].cuda(), R, C)
    assert torch.equal(e1, e2) and torch.equal(r1, r2)            # no atomics: bit-reproducible
    eo, ro, _, _ = O.render_loss(d.double(), s.double(), inp["im"].double(), inp["seg"].double(), R, C)
    assert abs(e1.item() - eo.item()) < 1e-5 * eo.item()
    assert rel_max(r1.cpu(), ro) < 1e-5


def test_recon_loss_vs_golden(sgr, golden):
    """sgr.recon_loss (wrapperBRDFLight.py:171-188) against the reference's value and gradient."""
    from oracle import sg_oracle as O
    name, z, cfg = golden
    R, C = cfg["R"], cfg["C"]
    env = _t(z, "ref32_env").requires_grad_(True)
    ind = torch.ones(cfg["bn"], 1, 1, 1, device="cuda")
    err, scaled = sgr.recon_loss(env, _t(z, "in_env_gt"), _t(z, "in_seg"), ind, R, C, return_scaled=True)
    ref = float(z["ref32_recon_err"][0])
    assert scalar_close(err.item(), ref, 0.0, 2e-5), (name, err.item(), ref)
    assert rel_l2(scaled.detach().cpu(), z["ref32_env_scaled"]) < 1e-5, name
    (g,) = torch.autograd.grad(err, [env])
    eo = torch.from_numpy(z["ref32_env"]).double().requires_grad_(True)
    co, _, _, _ = O.recon_loss(eo, torch.from_numpy(z["in_env_gt"]).double(), torch.from_numpy(z["in_seg"]).double(),
                               torch.ones(cfg["bn"], 1, 1, 1, dtype=torch.float64), R, C)
    (go,) = torch.autograd.grad(co, [eo])
    assert rel_l2(g.cpu(), go) < 1e-4, (name, rel_l2(g.cpu(), go))


def test_recon_loss_masks_and_dark_envs(sgr):
    """env_ind == 0 images and all-dark ground-truth cells drop out of both sums; odd J takes the scalar path."""
    from oracle import sg_oracle as O
    bn, R, C, eh, ew = 3, 5, 7, 3, 5
    g = torch.Generator().manual_seed(3)
    env = torch.rand(bn, 3, R, C, eh, ew, generator=g) * 2
    gt = torch.rand(bn, 3, R, C, eh, ew, generator=g) * 2
    gt[0, :, 1:3, 2:5] = 0.0                       # dark cells
    seg = (torch.rand(bn, 1, R, C, generator=g) < 0.7).float()
    ind = torch.tensor([1.0, 0.0, 1.0]).reshape(bn, 1, 1, 1)
    err = sgr.recon_loss(env.cuda(), gt.cuda(), seg.cuda(), ind.cuda(), R, C)
    eo, _, _, _ = O.recon_loss(env.double(), gt.double(), seg.double(), ind.double(), R, C)
    assert scalar_close(err.item(), eo.item(), 0.0, 1e-5), (err.item(), eo.item())


def test_render_loss_is_deterministic_and_its_result_may_be_modified_in_place(sgr):
    """The loss value and the gradient scale the backward pass needs live in separate buffers (they used to be two
    elements of one, and ``err *= w`` on the returned scalar broke backward); repeated evaluations are bit-identical."""
    from oracle import sg_oracle as O
    for bn, imH, imW, R, C in ((16, 240, 320, 120, 160), (3, 18, 26, 9, 13)):
        inp = O.synthetic_inputs(bn, imH, imW, R, C, 12, seed=5)
        g = torch.Generator().manual_seed(2)
        d0, s0 = torch.rand(bn, 3, R, C, generator=g).cuda(), torch.rand(bn, 3, R, C, generator=g).cuda() * 0.3
        im, seg = inp["im"].cuda(), inp["seg"].cuda()

        def run(scale):
            d, s = d0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
            err, ren = sgr.render_loss(d, s, im, seg, R, C)
            val = err.detach().clone()
            err *= scale                                      # in place on the returned scalar
            gd, gs = torch.autograd.grad(err, [d, s])
            return val, ren, gd, gs

        a, b, c = run(1.0), run(1.0), run(2.0)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert torch.equal(c[0], a[0]) and torch.equal(c[2], 2.0 * a[2]) and torch.equal(c[3], 2.0 * a[3])


def test_last_arriving_workgroup_fold_over_many_launches(sgr):
    """The batch totals of the render loss are folded by whichever workgroup of the third pass arrives last (release on its ticket,
    acquire by the last arrival -- round 3 relied on gfx950's write-through stores being counted in vmcnt).  A lost or
    stale partial would show up as a wrong total: 400 launches on fresh inputs, each checked against the totals recomputed in fp64 from
    the pass's own outputs (rendered, pooled image, pooled mask) and against the value of the two-step route (totals, then a separate
    sgr_loss_finalize launch)."""
    ops = torch.ops.sgrender
    bn, imH, imW, R, C = 16, 240, 320, 120, 160
    g = torch.Generator(device="cuda").manual_seed(11)
    worst = 0.0
    for it in range(400):
        d = torch.rand(bn, 3, R, C, device="cuda", generator=g)
        s = torch.rand(bn, 3, R, C, device="cuda", generator=g) * 0.3
        im = torch.rand(bn, 3, imH, imW, device="cuda", generator=g)
        seg = (torch.rand(bn, 1, imH, imW, device="cuda", generator=g) < 0.9).float()
        loss, scale, parts, rendered, im_s, seg_s, coef = ops.render_loss(d, s, im, seg, R, C, True)
        num = (((rendered.double() - im_s.double()) ** 2) * seg_s.double()).sum()
        den = seg_s.double().sum()
        got = parts.double()
        e = max(abs(got[0] - num).item() / num.item(), abs(got[1] - den).item() / den.item())
        worst = max(worst, e)
        assert e < 2e-6, (it, got.tolist(), num.item(), den.item())
        assert abs(loss.double() - num / den.clamp_min(1e-5) / 3.0).item() < 2e-6 * loss.item(), it
        if it % 40 == 0:      # the sharded route's totals come out of the same fold; its loss out of a separate launch
            _, _, parts2, _, _, _, _ = ops.render_loss(d, s, im, seg, R, C, False)
            assert torch.equal(parts2, parts), it
    print(f"last-arriving fold: worst relative deviation of the totals over 400 launches {worst:.2e}")


@pytest.mark.parametrize("bn,imH,imW,R,C", [(16, 240, 320, 120, 160), (3, 18, 26, 9, 13), (2, 12, 20, 12, 20)])
def test_render_loss_third_pass_gradients_equal_the_separate_backward(sgr, bn, imH, imW, R, C):
    """ABI 5: sgr_render_loss_fwd_total_grads writes weight * d loss / d{diffuse, spec} from its third pass (the fused light objective
    has no loss_bwd launch between its two heavy kernels).  Bit-identical to the three passes followed by sgr_render_loss_bwd_scaled,
    and the loss-side outputs are untouched by the extra stores."""
    from inverserenderingofindoorscene_amd import _lib
    from inverserenderingofindoorscene_amd.ops import _ptr, _stream
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(5 + bn)
    d = torch.rand(bn, 3, R, C, device=dev, generator=g) * 1.5
    s = torch.rand(bn, 3, R, C, device=dev, generator=g) * 0.6
    im = torch.rand(bn, 3, imH, imW, device=dev, generator=g)
    seg = (torch.rand(bn, 1, imH, imW, device=dev, generator=g) < 0.9).float()
    weight = 0.37

    def outputs():
        e = lambda *sh: torch.empty(*sh, device=dev)
        return dict(im_s=e(bn, 3, R, C), seg_s=e(bn, 1, R, C), rendered=e(bn, 3, R, C), coef=e(bn, 2), parts=e(2), loss=e(1), scale=e(1),
                    gd=e(bn, 3, R, C), gs=e(bn, 3, R, C), ws=e(lib.sgr_loss_workspace_floats(bn)))

    a, b = outputs(), outputs()
    st = _stream(dev)
    rc = lib.sgr_render_loss_fwd_total(_ptr(d), _ptr(s), _ptr(im), _ptr(seg), _ptr(a["im_s"]), _ptr(a["seg_s"]), _ptr(a["rendered"]), _ptr(a["coef"]),
                                       _ptr(a["parts"]), _ptr(a["loss"]), _ptr(a["scale"]), 3.0, _ptr(a["ws"]), bn, R, C, imH, imW, st)
    assert rc == 0
    rc = lib.sgr_render_loss_bwd_scaled(None, weight, _ptr(a["scale"]), _ptr(d), _ptr(s), _ptr(a["im_s"]), _ptr(a["seg_s"]), _ptr(a["coef"]), _ptr(a["gd"]),
                                        _ptr(a["gs"]), bn, R, C, st)
    assert rc == 0
    rc = lib.sgr_render_loss_fwd_total_grads(_ptr(d), _ptr(s), _ptr(im), _ptr(seg), _ptr(b["im_s"]), _ptr(b["seg_s"]), _ptr(b["rendered"]), _ptr(b["coef"]),
                                             _ptr(b["parts"]), _ptr(b["loss"]), _ptr(b["scale"]), 3.0, weight, _ptr(b["gd"]), _ptr(b["gs"]), _ptr(b["ws"]),
                                             bn, R, C, imH, imW, st)
    assert rc == 0
    torch.cuda.synchronize()
    for k in ("im_s", "seg_s", "rendered", "coef", "parts", "loss", "scale", "gd", "gs"):
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
    assert a["gd"].abs().max().item() > 0 and torch.isfinite(a["gd"]).all()
    # the one-rank pair is required: NULL loss / scale is refused
    rc = lib.sgr_render_loss_fwd_total_grads(_ptr(d), _ptr(s), _ptr(im), _ptr(seg), _ptr(b["im_s"]), _ptr(b["seg_s"]), _ptr(b["rendered"]), _ptr(b["coef"]),
                                             _ptr(b["parts"]), None, None, 3.0, weight, _ptr(b["gd"]), _ptr(b["gs"]), _ptr(b["ws"]), bn, R, C, imH, imW, st)
    assert rc != 0
