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
