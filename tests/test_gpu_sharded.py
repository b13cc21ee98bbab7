"""GPU, world_size 2 on ONE device: the product's HIP loss kernels under batch sharding.

Two ranks (torch.multiprocessing spawn, gloo backend -- it moves CUDA tensors through the host, so one GPU is enough)
each run the render layer, ``sgr.render_loss(..., group)`` and ``sgr.light_objective(..., group)`` -- also with
``decoder_outputs=True`` and at config-5 shapes (24 lobes, 16x32 directions, ragged 16-pixel tiles) -- on their half of the batch; the losses and the SG gradients must equal those of the single-process full batch
(wrapperBRDFLight.py:192,205-207: the normaliser is the batch-global mask sum).  The 8-GPU RCCL run itself belongs to the
driver (bench.py --gpus N)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
# the three cases: round 2's (12 lobes, 8x16), round 3's entry points -- the decoder heads as the objective kernels' prologue
# (decoder_outputs=True, premap 3), and config-5 shapes (24 lobes on the 16x32 grid, 35 env cells: ragged 16-pixel tiles)
CASES = {
    "k12_8x16": dict(bn=4, imH=24, imW=32, R=12, C=16, K=12, eh=8, ew=16, heads=False),
    "k12_8x16_decoder_outputs": dict(bn=4, imH=24, imW=32, R=12, C=16, K=12, eh=8, ew=16, heads=True),
    "k24_16x32_ragged": dict(bn=4, imH=10, imW=14, R=5, C=7, K=24, eh=16, ew=32, heads=False),
    "k24_16x32_decoder_outputs": dict(bn=4, imH=10, imW=14, R=5, C=7, K=24, eh=16, ew=32, heads=True),
}
NAMES = ("albedo", "normal", "rough", "axis", "lamb", "weight")
SG = ("axis", "lamb", "weight")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(c):
    from oracle import sg_oracle as O      # checker-side input generator only
    inp = O.synthetic_inputs(c["bn"], c["imH"], c["imW"], c["R"], c["C"], c["K"], c["eh"], c["ew"], seed=4242)
    inp["seg"][1] = 0.0                    # uneven denominators across the shards
    inp["ind"] = torch.tensor([1.0, 1.0, 0.0, 1.0]).reshape(c["bn"], 1, 1, 1)
    if c["heads"]:                         # the three light decoders' last-convolution outputs (models.py:336-346 come after them)
        g = torch.Generator().manual_seed(99)
        bn, K, R, C = c["bn"], c["K"], c["R"], c["C"]
        inp["axis"] = torch.randn(bn, 3 * K, R, C, generator=g)
        inp["lamb"] = torch.randn(bn, K, R, C, generator=g)
        inp["weight"] = torch.randn(bn, 3 * K, R, C, generator=g)
    return inp


def _run(case, sl, group):
    import inverserenderingofindoorscene_amd as sgr
    c = CASES[case]
    R, C = c["R"], c["C"]
    inp = _inputs(c)
    x = {k: v[sl].cuda().contiguous() for k, v in inp.items()}
    for k in SG:
        x[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=c["ew"], envHeight=c["eh"])
    if c["heads"]:
        axis, lamb, weight, _ = sgr.light_heads(x["axis"], x["lamb"], x["weight"])      # the unfused route starts from the activated tensors
    else:
        axis, lamb, weight = x["axis"], x["lamb"], x["weight"]
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], axis, lamb, weight, need_env=True)
    err, _ = sgr.render_loss(d, s, x["im"], x["seg"], R, C, group=group)
    rec = sgr.recon_loss(env, x["env_gt"], x["seg"], x["ind"], R, C, group=group)
    g1 = torch.autograd.grad(err + 10.0 * rec, [x[k] for k in SG])
    obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"],
                              x["env_gt"], x["ind"], 1.0, 10.0, group=group, decoder_outputs=c["heads"])
    g2 = torch.autograd.grad(obj[0], [x[k] for k in SG])
    torch.cuda.synchronize()
    return dict(err=err.item(), rec=rec.item(), obj=obj[0].item(), obj_err=obj[1].item(), obj_rec=obj[2].item(),
                g1=[t.cpu() for t in g1], g2=[t.cpu() for t in g2])


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = CASES[case]["bn"] // world
    out[rank] = _run(case, slice(rank * per, (rank + 1) * per), dist.group.WORLD)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_on_one_gpu_match_the_full_batch(case):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    full = _run(case, slice(0, CASES[case]["bn"]), None)
    per = CASES[case]["bn"] // world
    for r in range(world):
        o = out[r]
        for k in ("err", "rec", "obj", "obj_err", "obj_rec"):
            assert abs(o[k] - full[k]) <= 2e-6 * abs(full[k]), (case, r, k, o[k], full[k])
        for tag in ("g1", "g2"):
            for name, a, b in zip(SG, o[tag], full[tag]):
                b = b[r * per:(r + 1) * per]
                assert torch.isfinite(a).all()
                rel = ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
                assert rel < 5e-6, (case, r, tag, name, rel)
