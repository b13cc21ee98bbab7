"""GPU: the steps captured in a HIP graph (torch.cuda.CUDAGraph) replay bit-identically to eager launches -- the layer's
forward + backward, the with-render-loss step and the fused trainLight objective (wrapperBRDFLight.py:167-207) -- also after
the inputs are overwritten in place between replays.  What this pins: every workspace / table the operators use is either
allocated from torch's (capture-aware) allocator or built before the capture; the render loss's last-arrival ticket is re-armed
inside the captured launches, not by the host; nothing in the path synchronises or reads a device value on the host."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SG = ("axis", "lamb", "weight")


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _inputs(bn, imH, imW, R, C, K, seed):
    from oracle import sg_oracle as O
    return {k: v.cuda() for k, v in O.synthetic_inputs(bn, imH, imW, R, C, K, seed=seed).items()}


def _capture(step, static):
    """Warm up on a side stream (torch's capture recipe), capture one step, return (graph, outputs-of-the-captured-step)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(static)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step(static)
    return g, outs


def _check_replays(step, bn, imH, imW, R, C, K):
    static = _inputs(bn, imH, imW, R, C, K, seed=3)
    for k in SG:
        static[k].requires_grad_(True)
    graph, outs = _capture(step, static)
    for seed in (3, 11, 12):
        fresh = _inputs(bn, imH, imW, R, C, K, seed=seed)
        with torch.no_grad():
            for k, v in fresh.items():
                static[k].copy_(v)
        graph.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        eager_in = {k: v.clone() for k, v in fresh.items()}
        for k in SG:
            eager_in[k].requires_grad_(True)
        want = step(eager_in)
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (seed, i, (a - b).abs().max().item())
            assert torch.isfinite(a).all()


def test_layer_step_replays(sgr):
    bn, imH, imW, R, C, K = 2, 48, 64, 24, 32, 12
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    gen = torch.Generator(device="cuda").manual_seed(5)
    ct_env = torch.randn((bn, 3, R, C, 8, 16), device="cuda", generator=gen) * 1e-3
    ct = torch.randn((bn, 3, R, C), device="cuda", generator=gen)

    def step(x):
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
        g = torch.autograd.grad([env, d, s], [x[k] for k in SG], grad_outputs=[ct_env, ct, ct])
        return [env, d, s, *g]

    _check_replays(step, bn, imH, imW, R, C, K)


def test_render_loss_step_replays(sgr):
    """three loss launches with the last-arrival fold + the loss backward: the ticket is re-armed by the captured launches themselves"""
    bn, imH, imW, R, C, K = 3, 36, 52, 18, 26, 12
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)

    def step(x):
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=False)
        err, ren = sgr.render_loss(d, s, x["im"], x["seg"], R, C)
        g = torch.autograd.grad(err, [x[k] for k in SG])
        return [err, ren, *g]

    _check_replays(step, bn, imH, imW, R, C, K)


@pytest.mark.parametrize("decoder_outputs", [False, True])
def test_light_objective_step_replays(sgr, decoder_outputs):
    bn, imH, imW, R, C, K = 2, 48, 64, 24, 32, 12
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)
    ind = torch.ones(bn, 1, 1, 1, device="cuda")

    def step(x):
        obj, rerr, cerr, ren, coef = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"],
                                                         x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0, decoder_outputs=decoder_outputs)
        g = torch.autograd.grad(obj, [x[k] for k in SG])
        return [obj, rerr, cerr, ren, coef, *g]

    _check_replays(step, bn, imH, imW, R, C, K)


def test_capture_step_helper_at_the_reference_batch(sgr):
    """sgr.capture_step (the packaged recipe): layer + render loss + backward at the reference's default batch of 5
    (trainLight.py:28), replayed after the static inputs are overwritten -- bit-identical to the eager step on the same values."""
    bn, imH, imW, R, C, K = 5, 48, 64, 24, 32, 12
    static = _inputs(bn, imH, imW, R, C, K, seed=21)
    for k in SG:
        static[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R)

    def step(x):
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
        err, rendered = sgr.render_loss(d, s, x["im"], x["seg"], R, C)
        g = torch.autograd.grad([err, env], [x[k] for k in SG], grad_outputs=[None, torch.full_like(env, 1e-3)])
        return (err, rendered) + tuple(g)

    captured = sgr.capture_step(lambda: step(static))
    assert isinstance(captured, sgr.CapturedStep)
    for seed in (21, 22, 23):
        fresh = _inputs(bn, imH, imW, R, C, K, seed=seed)
        with torch.no_grad():
            for k, v in fresh.items():
                static[k].copy_(v)
        got = [o.clone() for o in captured()]
        torch.cuda.synchronize()
        eager_in = {k: v.clone() for k, v in fresh.items()}
        for k in SG:
            eager_in[k].requires_grad_(True)
        want = step(eager_in)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    assert captured.replays == 3
