"""CPU: the render path is registered with the PyTorch dispatcher (torch.ops.sgrender.*): operator schemas exist, the
fake-tensor shape functions agree with the documented shapes (meta tensors, forward and through autograd), the package's
classes route through the operators, and CPU tensors are rejected (no CPU kernels)."""
import pytest
import torch

import inverserenderingofindoorscene_amd as sgr  # noqa: F401  (registers the operators)

OPS = ("sg_to_env", "sg_to_env_bwd", "render_env", "render_env_bwd_env", "render_bwd_brdf", "fused_render", "fused_render_bwd_sg")
bn, K, R, C, eh, ew, imH, imW = 2, 12, 6, 8, 8, 16, 12, 16


def m(*shape, grad=False):
    return torch.empty(*shape, device="meta", requires_grad=grad)


def test_operators_have_schemas():
    for name in OPS:
        op = getattr(torch.ops.sgrender, name)
        schema = str(op.default._schema)
        assert schema.startswith(f"sgrender::{name}("), schema
    assert "Tensor? g_env" in str(torch.ops.sgrender.fused_render_bwd_sg.default._schema)


def test_fake_shapes_forward_and_backward():
    ops = torch.ops.sgrender
    axis, lamb, weight = m(bn, K, 3, R, C, grad=True), m(bn, K, R, C, grad=True), m(bn, 3 * K, R, C, grad=True)
    alb, nrm, rgh = m(bn, 3, imH, imW, grad=True), m(bn, 3, imH, imW, grad=True), m(bn, 1, imH, imW, grad=True)
    env, lam_t, w_t = ops.sg_to_env(axis, lamb, weight, eh, ew, True, True)
    assert tuple(env.shape) == (bn, 3, R, C, eh, ew) and lam_t.shape == lamb.shape and w_t.shape == weight.shape
    env0, lam0, _ = ops.sg_to_env(axis, lamb, weight, eh, ew, False, False)
    assert lam0.numel() == 0 and tuple(env0.shape) == (bn, 3, R, C, eh, ew)
    d, s = ops.render_env(alb, nrm, rgh, env, 57.0, 0.05, [0.0, 0.0, 0.0])
    assert tuple(d.shape) == tuple(s.shape) == (bn, 3, R, C)
    g = torch.autograd.grad([d.sum() + s.sum()], [alb, nrm, rgh, axis, lamb, weight])
    assert [tuple(t.shape) for t in g] == [tuple(t.shape) for t in (alb, nrm, rgh, axis, lamb, weight)]
    env2, d2, s2, lt2, wt2 = ops.fused_render(alb, nrm, rgh, axis, lamb, weight, eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], True, True, True)
    assert tuple(env2.shape) == (bn, 3, R, C, eh, ew) and tuple(d2.shape) == (bn, 3, R, C)
    assert lt2.shape == lamb.shape and wt2.shape == weight.shape      # the post-tan hand-off to the backward (premap mode 2)
    g2 = torch.autograd.grad([env2.sum() + d2.sum()], [axis, lamb, weight, alb])
    assert [tuple(t.shape) for t in g2] == [tuple(axis.shape), tuple(lamb.shape), tuple(weight.shape), tuple(alb.shape)]
    env3, d3, _, lt3, _ = ops.fused_render(alb, nrm, rgh, axis, lamb, weight, eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], True, False, False)
    assert env3.numel() == 0 and lt3.numel() == 0 and tuple(d3.shape) == (bn, 3, R, C)
    g3 = torch.autograd.grad([d3.sum()], [axis, lamb, weight])
    assert [tuple(t.shape) for t in g3] == [tuple(axis.shape), tuple(lamb.shape), tuple(weight.shape)]


def test_operators_reject_cpu_tensors():
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        torch.ops.sgrender.fused_render(z(1, 3, 2, 2), z(1, 3, 2, 2), z(1, 1, 2, 2), z(1, 2, 3, 2, 2), z(1, 2, 2, 2), z(1, 6, 2, 2),
                                        2, 4, 57.0, 0.05, [0.0, 0.0, 0.0], True, True, False)


def test_zero_sized_inputs_are_refused_for_fake_and_device_tensors_alike():
    """the reference's broadcasts would return zero-sized results; the kernels have no launch for them: refused with a message, and
    by the shape function too, so that a traced graph cannot pass tracing and then fail on the device"""
    e = lambda *s: torch.empty(*s, device="meta")
    with pytest.raises(RuntimeError, match="zero-sized SG tensors"):
        torch.ops.sgrender.sg_to_env(e(0, 12, 3, 4, 4), e(0, 12, 4, 4), e(0, 36, 4, 4), 8, 16, True, False)
    with pytest.raises(RuntimeError, match="zero-sized BRDF maps"):
        torch.ops.sgrender.render_env(e(0, 3, 8, 8), e(0, 3, 8, 8), e(0, 1, 8, 8), e(0, 3, 4, 4, 8, 16), 57.0, 0.05, [0.0, 0.0, 0.0])


# --------------------------------------------------------------------------- #
# round 4: the loss / heads / objective operators of the C++ extension          #
# --------------------------------------------------------------------------- #
ALL_OPS = OPS + ("lsregress_coef", "lsregress_diffspec_coef", "render_loss", "render_loss_finalize", "render_loss_bwd", "recon_loss_parts", "recon_loss_bwd",
                 "light_heads", "light_heads_bwd", "sg_shading", "light_albedo_scale", "light_encoder_input", "rescale_grads_", "attach_grads",
                 "light_objective_fwdbwd", "light_objective", "light_objective_stage1", "light_objective_stage2", "light_objective_stage3")


def test_every_operator_is_registered_by_the_cpp_extension():
    import os
    from inverserenderingofindoorscene_amd import ops as host
    assert os.path.basename(host.EXT_PATH) == "libsgrender_torch.so" and os.path.isfile(host.EXT_PATH)
    for name in ALL_OPS:
        op = getattr(torch.ops.sgrender, name).default
        assert str(op._schema).startswith(f"sgrender::{name}("), name
        # registered from C++ (TORCH_LIBRARY), not by a Python torch.library.custom_op
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"sgrender::{name}", "Meta"), name
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"sgrender::{name}", "CUDA"), name
    assert "bool want_tan=False" in str(torch.ops.sgrender.fused_render.default._schema)      # defaulted, old call sites keep working


def _objective_args():
    alb, nrm, rgh = m(bn, 3, imH, imW), m(bn, 3, imH, imW), m(bn, 1, imH, imW)
    axis, lamb, weight = m(bn, K, 3, R, C, grad=True), m(bn, K, R, C, grad=True), m(bn, 3 * K, R, C, grad=True)
    im, seg, gt, ind = m(bn, 3, imH, imW), m(bn, 1, imH, imW), m(bn, 3, R, C, eh, ew), m(bn, 1, 1, 1)
    return alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind


def test_loss_and_heads_operators_fake_shapes_and_autograd():
    ops = torch.ops.sgrender
    d, s = m(bn, 3, R, C, grad=True), m(bn, 3, R, C, grad=True)
    im, seg = m(bn, 3, imH, imW), m(bn, 1, imH, imW)
    loss, scale, parts, rendered, im_s, seg_s, coef = ops.render_loss(d, s, im, seg, R, C, True)
    assert loss.dim() == 0 and tuple(parts.shape) == (2,) and tuple(rendered.shape) == (bn, 3, R, C) and tuple(coef.shape) == (bn, 2)
    assert loss.requires_grad and not rendered.requires_grad and not parts.requires_grad
    gd, gs = torch.autograd.grad(loss, [d, s])
    assert gd.shape == d.shape and gs.shape == s.shape
    # the sharded route: totals without an autograd node, the node on the finalize operator
    with torch.no_grad():
        _, _, parts2, _, im_s2, seg_s2, coef2 = ops.render_loss(d, s, im, seg, R, C, False)
    loss2, scale2 = ops.render_loss_finalize(d, s, parts2, im_s2, seg_s2, coef2)
    assert loss2.dim() == 0 and loss2.requires_grad and not scale2.requires_grad
    assert [t.shape for t in torch.autograd.grad(loss2, [d, s])] == [d.shape, s.shape]
    xa, xl, xw = m(bn, 3 * K, R, C, grad=True), m(bn, K, R, C, grad=True), m(bn, 3 * K, R, C, grad=True)
    axis, lamb, weight, packed = ops.light_heads(xa, xl, xw, True)
    assert tuple(axis.shape) == (bn, K, 3, R, C) and tuple(packed.shape) == (bn, 7 * K, R, C)
    g = torch.autograd.grad(axis.sum() + packed.sum(), [xa, xl, xw])
    assert [t.shape for t in g] == [xa.shape, xl.shape, xw.shape]
    env = m(bn, 3, R, C, eh, ew, grad=True)
    parts3, mask, rcoef = ops.recon_loss_parts(env, m(bn, 3, R, C, eh, ew), m(bn, 1, R, C), m(bn, 1, 1, 1), 1.0)
    assert tuple(mask.shape) == (bn, R * C) and tuple(rcoef.shape) == (bn,) and parts3.requires_grad and not mask.requires_grad
    assert torch.autograd.grad(parts3[0], [env])[0].shape == env.shape
    assert tuple(ops.lsregress_coef(m(bn, 3, R, C), m(bn, 3, R, C)).shape) == (bn,)
    assert tuple(ops.lsregress_diffspec_coef(m(bn, 3, R, C), m(bn, 3, R, C), m(bn, 3, R, C)).shape) == (bn, 2)
    assert tuple(ops.sg_shading(m(bn, K, 3, R, C), m(bn, K, R, C), m(bn, 3 * K, R, C), 16, 32, 1).shape) == (bn, 3, R, C)
    out, alb_n, dep_n = ops.light_encoder_input(m(bn, 3, imH, imW), m(bn, 3, imH, imW), m(bn, 3, imH, imW), m(bn, 1, imH, imW), m(bn, 1, imH, imW), 48, 64)
    assert tuple(out.shape) == (bn, 11, 48, 64) and tuple(dep_n.shape) == (bn, 1, imH, imW)


def test_light_objective_operator_graph_and_forward_only_mode():
    ops = torch.ops.sgrender
    alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind = _objective_args()
    cfg = (eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], 1.0, 10.0, 1.0, False, False)
    obj, rerr, cerr, rendered, coef = ops.light_objective(alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind, *cfg)
    assert obj.dim() == 0 and obj.requires_grad and not rerr.requires_grad and not cerr.requires_grad
    assert tuple(rendered.shape) == (bn, 3, R, C) and tuple(coef.shape) == (bn,)
    g = torch.autograd.grad(obj, [axis, lamb, weight])
    assert [t.shape for t in g] == [axis.shape, lamb.shape, weight.shape]
    # forward-only: no autograd node, and the gradient half of the operator is not even allocated
    with torch.no_grad():
        o2 = ops.light_objective(alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind, *cfg)
    assert not o2[0].requires_grad and o2[0].grad_fn is None
    o3 = ops.light_objective(alb, nrm, rgh, axis.detach(), lamb.detach(), weight.detach(), im, seg, gt, ind, *cfg)
    assert not o3[0].requires_grad
    full = ops.light_objective_fwdbwd(alb, nrm, rgh, axis.detach(), lamb.detach(), weight.detach(), im, seg, gt, ind, *cfg, False)
    assert full[5].numel() == 0 and full[6].numel() == 0 and full[7].numel() == 0 and full[8].numel() == 0
    full = ops.light_objective_fwdbwd(alb, nrm, rgh, axis.detach(), lamb.detach(), weight.detach(), im, seg, gt, ind, *cfg, True)
    assert full[5].shape == axis.shape and full[6].shape == lamb.shape and full[7].shape == weight.shape and tuple(full[8].shape) == (2,)
    # a grad-requiring BRDF map is refused at the call (trainLight mode only)
    with pytest.raises(RuntimeError, match="SG parameters only"):
        ops.light_objective(alb, nrm, rgh.clone().requires_grad_(True), axis, lamb, weight, im, seg, gt, ind, *cfg)
    # the sharded route's stage operators and the node that attaches their gradients
    st1 = ops.light_objective_stage1(alb, nrm, rgh, axis.detach(), lamb.detach(), weight.detach(), im, seg, gt, ind, eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], False, False)
    assert len(st1) == 12 and tuple(st1[8].shape) == (4,)
    diffuse, spec, mask, coef1, im_s, seg_s, rendered1, coef_ds, sums, ws, lam_t, w_t = st1
    st2 = ops.light_objective_stage2(alb, nrm, rgh, axis.detach(), lamb.detach(), weight.detach(), gt, mask, coef1, diffuse, spec, im_s, seg_s, coef_ds, sums, ws,
                                     lam_t, w_t, eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], 1.0, 10.0, 1.0, False, True)
    assert st2[1].shape == axis.shape and tuple(st2[4].shape) == (2,)
    objective, recon = ops.light_objective_stage3(st2[0], st2[4][0:1], sums, 1.0, 10.0, eh, ew)
    out = ops.attach_grads(objective, axis, lamb, weight, st2[1], st2[2], st2[3], m(2))
    assert out.requires_grad and [t.shape for t in torch.autograd.grad(out, [axis, lamb, weight])] == [axis.shape, lamb.shape, weight.shape]


def test_stage_operators_report_real_sizes_and_check_their_inputs():
    """stage 1's Meta kernel reports the workspace at its real size (the C ABI's sgr_fused_recon_workspace_floats, restated as
    host arithmetic in the extension), stage 2 declares that it writes it (`Tensor(a!) ws`) and refuses stage-1 tensors of another shape --
    for meta tensors exactly as for device tensors."""
    from inverserenderingofindoorscene_amd import _lib
    ops = torch.ops.sgrender
    L = _lib.load()
    for b_, r_, c_ in ((1, 1, 1), (2, 6, 8), (16, 120, 160), (4, 240, 320), (3, 7, 5)):
        assert ops.recon_workspace_floats(b_, r_, c_) == L.sgr_fused_recon_workspace_floats(b_, r_, c_), (b_, r_, c_)
    alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind = _objective_args()
    a_, l_, w_ = axis.detach(), lamb.detach(), weight.detach()
    st1 = ops.light_objective_stage1(alb, nrm, rgh, a_, l_, w_, im, seg, gt, ind, eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], False, False)
    diffuse, spec, mask, coef1, im_s, seg_s, rendered1, coef_ds, sums, ws, lam_t, w_t = st1
    assert ws.numel() == L.sgr_fused_recon_workspace_floats(bn, R, C)
    assert "Tensor(a!) ws" in str(ops.light_objective_stage2.default._schema)
    tail = (eh, ew, 57.0, 0.05, [0.0, 0.0, 0.0], 1.0, 10.0, 1.0, False, True)
    ops.light_objective_stage2(alb, nrm, rgh, a_, l_, w_, gt, mask, coef1, diffuse, spec, im_s, seg_s, coef_ds, sums, ws, lam_t, w_t, *tail)
    bad = dict(mask=m(bn, R * C + 1), coef=m(bn + 1), diffuse=m(bn, 3, R, C + 1), im_s=m(bn + 1, 3, R, C), seg_s=m(bn, 1, R + 1, C), coef_ds=m(bn, 3), sums=m(5),
               ws=m(ws.numel() - 1))
    for name, t in bad.items():
        args = dict(mask=mask, coef=coef1, diffuse=diffuse, spec=spec, im_s=im_s, seg_s=seg_s, coef_ds=coef_ds, sums=sums, ws=ws)
        args[name] = t
        with pytest.raises(RuntimeError, match="light_objective_stage2"):
            ops.light_objective_stage2(alb, nrm, rgh, a_, l_, w_, gt, args["mask"], args["coef"], args["diffuse"], args["spec"], args["im_s"], args["seg_s"],
                                       args["coef_ds"], args["sums"], args["ws"], lam_t, w_t, *tail)


def test_light_objective_refuses_grad_requiring_maps_on_both_routes():
    """the one-rank operator raised for a BRDF map / image that requires grad, the sharded route (stage operators under no_grad)
    silently returned no gradient for it.  The check now sits in front of the branch."""
    import torch.distributed as dist
    from inverserenderingofindoorscene_amd import losses
    alb, nrm, rgh, axis, lamb, weight, im, seg, gt, ind = _objective_args()
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, isCuda=False)

    class _Group:      # any explicit group selects the sharded route (losses._sharded); the check must fire before any collective
        pass
    for group in (None, _Group()):
        for k in range(3):
            maps = [alb, nrm, rgh]
            maps[k] = maps[k].clone().requires_grad_(True)
            with pytest.raises(RuntimeError, match="SG parameters only"):
                losses.light_objective(layer, maps[0], maps[1], maps[2], axis, lamb, weight, im, seg, gt, ind, group=group)
        with pytest.raises(RuntimeError, match="SG parameters only"):
            losses.light_objective(layer, alb, nrm, rgh, axis, lamb, weight, im.clone().requires_grad_(True), seg, gt, ind, group=group)
    assert not dist.is_initialized()


def test_every_operator_rejects_cpu_tensors():
    z = torch.zeros
    ops = torch.ops.sgrender
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.render_loss(z(1, 3, 2, 2), z(1, 3, 2, 2), z(1, 3, 2, 2), z(1, 1, 2, 2), 2, 2, True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.light_heads(z(1, 6, 2, 2), z(1, 2, 2, 2), z(1, 6, 2, 2), False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.light_objective(z(1, 3, 2, 2), z(1, 3, 2, 2), z(1, 1, 2, 2), z(1, 2, 3, 2, 2), z(1, 2, 2, 2), z(1, 6, 2, 2), z(1, 3, 2, 2), z(1, 1, 2, 2),
                            z(1, 3, 2, 2, 8, 16), z(1, 1, 1, 1), 8, 16, 57.0, 0.05, [0.0, 0.0, 0.0], 1.0, 10.0, 1.0, False, False)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.lsregress_coef(z(1, 3, 2, 2), z(1, 3, 2, 2))
