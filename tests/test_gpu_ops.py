"""GPU: the registered operators (torch.ops.sgrender.*, the C++ extension csrc/sgr_torch.cpp) behave as a torch
extension: ``torch.library.opcheck`` (schema, fake-tensor agreement with the real kernels, autograd registration, AOT
dispatch) and a ``torch.compile`` smoke test -- the layer + a loss captured as ONE graph (fullgraph=True; backend
aot_eager: dynamo + AOTAutograd + fake tensors, no code generation -- there is nothing to generate around a hand-written
kernel), values and gradients equal to eager mode bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
bn, K, R, C, eh, ew, imH, imW = 2, 12, 12, 16, 8, 16, 24, 32


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _inputs(grad=True):
    from oracle import sg_oracle as O      # input generator only
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=31)
    x = {k: v.cuda() for k, v in inp.items()}
    for k in ("axis", "lamb", "weight", "albedo", "normal", "rough"):
        x[k].requires_grad_(grad)
    return x


def test_opcheck(sgr):
    x = _inputs()
    cam = [0.0, 0.0, 0.0]
    ops = torch.ops.sgrender
    torch.library.opcheck(ops.fused_render, (x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew, 57.0, 0.05, cam, True, True, True))
    torch.library.opcheck(ops.fused_render, (x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew, 57.0, 0.05, cam, True, False, False))
    torch.library.opcheck(ops.fused_render, (x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew, 57.0, 0.05, cam, False, False, False))
    torch.library.opcheck(ops.sg_to_env, (x["axis"], x["lamb"], x["weight"], eh, ew, True, True))
    env = ops.sg_to_env(x["axis"], x["lamb"], x["weight"], eh, ew, True, False)[0].detach().requires_grad_(True)
    torch.library.opcheck(ops.render_env, (x["albedo"], x["normal"], x["rough"], env, 57.0, 0.05, cam))
    g = torch.randn(bn, 3, R, C, device="cuda")
    torch.library.opcheck(ops.fused_render_bwd_sg, (None, g, g, x["albedo"].detach(), x["normal"].detach(), x["rough"].detach(), x["axis"].detach(),
                                                    x["lamb"].detach(), x["weight"].detach(), eh, ew, 57.0, 0.05, cam, 1))
    # round 4: the loss / heads / objective operators live in the same C++ extension
    d = torch.rand(bn, 3, R, C, device="cuda", requires_grad=True)
    s = torch.rand(bn, 3, R, C, device="cuda", requires_grad=True)
    torch.library.opcheck(ops.render_loss, (d, s, x["im"], x["seg"], R, C, True))
    torch.library.opcheck(ops.render_loss, (d, s, x["im"], x["seg"], R, C, False))
    with torch.no_grad():
        _, _, parts, _, im_s, seg_s2, coef = ops.render_loss(d, s, x["im"], x["seg"], R, C, False)
    torch.library.opcheck(ops.render_loss_finalize, (d, s, parts, im_s, seg_s2, coef))      # the sharded route's autograd node
    xa = torch.randn(bn, 3 * K, R, C, device="cuda", requires_grad=True)
    xl = torch.randn(bn, K, R, C, device="cuda", requires_grad=True)
    xw = torch.randn(bn, 3 * K, R, C, device="cuda", requires_grad=True)
    torch.library.opcheck(ops.light_heads, (xa, xl, xw, True))
    env = ops.sg_to_env(x["axis"], x["lamb"], x["weight"], eh, ew, True, False)[0].detach().requires_grad_(True)
    seg_s = torch.nn.functional.avg_pool2d(x["seg"], 2)
    ind = torch.ones(bn, 1, 1, 1, device="cuda")
    torch.library.opcheck(ops.recon_loss_parts, (env, x["env_gt"], seg_s, ind, 1.0))
    # the objective produces its gradients in forward and hands them out through a node that rescales them in place: schema and
    # fake-tensor agreement are checked; the generic autograd / AOT checks do not apply to that design
    obj_args = (x["albedo"].detach(), x["normal"].detach(), x["rough"].detach(), x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"], ind,
                eh, ew, 57.0, 0.05, cam, 1.0, 10.0, 1.0, False, False)
    torch.library.opcheck(ops.light_objective, obj_args, test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(ops.light_objective_fwdbwd, tuple(t.detach() if torch.is_tensor(t) else t for t in obj_args) + (True,),
                          test_utils=("test_schema", "test_faketensor"))


def test_torch_compile_captures_the_layer(sgr):
    x = _inputs()
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    ct = torch.randn(bn, 3, R, C, eh, ew, device="cuda") * 1e-2

    def objective(albedo, normal, rough, axis, lamb, weight):
        env, d, s = layer.forwardSG(albedo, normal, rough, axis, lamb, weight, need_env=True)
        return (torch.clamp(d + s, 0, 1) ** 2).mean() + (env * ct).sum()

    args = [x[k] for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")]
    eager = objective(*args)
    g_eager = torch.autograd.grad(eager, args)
    compiled = torch.compile(objective, fullgraph=True, backend="aot_eager")
    out = compiled(*args)
    g_comp = torch.autograd.grad(out, args)
    assert torch.equal(out, eager)
    for a, b in zip(g_comp, g_eager):
        assert torch.equal(a, b)
    # the un-fused pair under FakeTensorMode: shapes only, no kernel launched
    from torch._subclasses.fake_tensor import FakeTensorMode
    detached = [t.detach() for t in args]
    with FakeTensorMode(allow_non_fake_inputs=False) as mode:
        fa = [mode.from_tensor(t) for t in detached]
        o2e = sgr.output2env(K)
        env, _, lam_t, w_t = o2e.output2env(fa[3], fa[4], fa[5])
        d, s = layer.forwardEnv(fa[0], fa[1], fa[2], env)
        assert tuple(env.shape) == (bn, 3, R, C, eh, ew) and tuple(d.shape) == (bn, 3, R, C) and lam_t.shape == fa[4].shape


def test_torch_compile_captures_layer_and_losses(sgr):
    """Round 4: the loss / heads operators are C++ autograd nodes too -- decoder heads -> fused layer -> render loss -> unfused
    reconstruction loss captured as ONE graph (fullgraph=True, aot_eager), values and gradients bit-identical to eager."""
    x = _inputs(grad=False)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    g = torch.Generator().manual_seed(3)
    raw = [(torch.randn(s, generator=g) * 0.7).cuda().requires_grad_(True) for s in ((bn, 3 * K, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
    ind = torch.ones(bn, 1, 1, 1, device="cuda")

    def objective(xa, xl, xw):
        axis, lamb, weight, _ = sgr.light_heads(xa, xl, xw)
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], axis, lamb, weight, need_env=True)
        err, _ = sgr.render_loss(d, s, x["im"], x["seg"], R, C)
        rec = sgr.recon_loss(env, x["env_gt"], x["seg"], ind, R, C)
        return err + 10.0 * rec

    eager = objective(*raw)
    g_eager = torch.autograd.grad(eager, raw)
    compiled = torch.compile(objective, fullgraph=True, backend="aot_eager")
    out = compiled(*raw)
    g_comp = torch.autograd.grad(out, raw)
    assert torch.equal(out, eager)
    for a, b in zip(g_comp, g_eager):
        assert torch.equal(a, b)
