"""CPU, world_size 2 over gloo: the N>1 path of the render loss.

The product's collective logic (`combine_loss_parts`: one all-reduce of [numerator, denominator],
gradient scale from the GLOBAL denominator) is pure torch + torch.distributed, so it runs here on
CPU tensors; the per-shard numerator/denominator come from the oracle (a checker, allowed in
tests).  The sharded loss and its gradient must equal the single-process full-batch values."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sg_oracle as O

BN, IMH, IMW, R, C, K = 4, 12, 16, 6, 8, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _full_batch():
    inp = O.synthetic_inputs(BN, IMH, IMW, R, C, K, seed=77, dtype=torch.float64)
    inp["seg"][1] = 0.0        # uneven denominators across shards
    env, d, s = O.render_from_sg(inp["albedo"], inp["normal"], inp["rough"], inp["axis"], inp["lamb"], inp["weight"])
    return inp, d.detach(), s.detach()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inverserenderingofindoorscene_amd.losses import combine_loss_parts
    inp, d, s = _full_batch()
    per = BN // world
    sl = slice(rank * per, (rank + 1) * per)
    d_l = d[sl].clone().requires_grad_(True)
    s_l = s[sl].clone().requires_grad_(True)
    _, _, num, den = O.render_loss(d_l, s_l, inp["im"][sl], inp["seg"][sl], R, C)
    loss = combine_loss_parts(num, den, group=None)
    loss.backward()
    # the fused light objective's two collectives as losses.light_objective issues them: the stage-1 vector [num_r, den_r, 0, den_e]
    # all-reduced in place before the backward pass, then the reconstruction numerator -- a one-element VIEW of stage 2's pair --
    # in place after it (the pair's second element, this shard's own mask sum, must stay untouched)
    from inverserenderingofindoorscene_amd.losses import _sharded
    sums = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 0.0, 5.0])
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=None)
    parts_b = torch.tensor([rank + 1.0, 99.0])
    num_e = parts_b[0:1]
    dist.all_reduce(num_e, op=dist.ReduceOp.SUM, group=None)
    out[rank] = (loss.item(), d_l.grad.clone(), s_l.grad.clone(), (sums.tolist(), parts_b.tolist(), _sharded(None)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_render_loss_matches_full_batch():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    inp, d, s = _full_batch()
    d_f = d.clone().requires_grad_(True)
    s_f = s.clone().requires_grad_(True)
    err, _, _, _ = O.render_loss(d_f, s_f, inp["im"], inp["seg"], R, C)
    err.backward()
    per = BN // world
    for r in range(world):
        loss_r, gd_r, gs_r, pair = out[r]
        assert pair == ([3.0, 30.0, 0.0, 10.0], [3.0, 99.0], True)
        assert abs(loss_r - err.item()) < 1e-12 * max(1.0, abs(err.item()))
        assert torch.allclose(gd_r, d_f.grad[r * per:(r + 1) * per], rtol=1e-10, atol=1e-14)
        assert torch.allclose(gs_r, s_f.grad[r * per:(r + 1) * per], rtol=1e-10, atol=1e-14)


def _tiny_net():
    torch.manual_seed(5)
    return torch.nn.Conv2d(3, 3, 3, padding=1, bias=True).double()


def _ddp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inverserenderingofindoorscene_amd.losses import combine_loss_parts, ddp_loss_scale
    inp, d, s = _full_batch()
    per = BN // world
    sl = slice(rank * per, (rank + 1) * per)
    net = torch.nn.parallel.DistributedDataParallel(_tiny_net())
    d_l, s_l = net(d[sl]), net(s[sl])              # a trainable stage in front of the loss (stands in for the light decoders)
    _, _, num, den = O.render_loss(d_l, s_l, inp["im"][sl], inp["seg"][sl], R, C)
    loss = combine_loss_parts(num, den, group=None)
    (loss * ddp_loss_scale()).backward()           # DDP averages the parameter gradients over the ranks
    out[rank] = (loss.item(), [p.grad.clone() for p in net.module.parameters()])
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_ddp_gradients_match_single_process():
    """The light decoders under DistributedDataParallel: with the loss scaled by ddp_loss_scale() the averaged
    parameter gradients equal the single-process full-batch gradients (what nn.DataParallel gives the reference)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    inp, d, s = _full_batch()
    net = _tiny_net()
    err, _, _, _ = O.render_loss(net(d), net(s), inp["im"], inp["seg"], R, C)
    err.backward()
    for r in range(world):
        loss_r, grads_r = out[r]
        assert abs(loss_r - err.item()) < 1e-12 * max(1.0, abs(err.item()))
        for g, p in zip(grads_r, net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-9, atol=1e-13)


def test_single_process_is_not_sharded():
    from inverserenderingofindoorscene_amd.losses import _sharded
    assert _sharded(None) is False      # no process group: light_objective / render_loss take their one-operator routes


def test_combine_single_process_is_plain_ratio():
    from inverserenderingofindoorscene_amd.losses import combine_loss_parts
    num = torch.tensor(3.0, requires_grad=True)
    den = torch.tensor(0.0)
    loss = combine_loss_parts(num, den)
    assert abs(loss.item() / (3.0 / 1e-5 / 3.0) - 1) < 1e-5      # max(den, 1e-5), wrapperBRDFLight.py:192
    loss.backward()
    assert abs(num.grad.item() / (1.0 / 1e-5 / 3.0) - 1) < 1e-5


def test_lsregress_diffspec_live_coefficients_match_the_oracle():
    """LSregressDiffSpec with grad-carrying first arguments (trainFineTune*_cascade1.py): the host layer's differentiable
    restatement against the oracle's (itself pinned to the reference), values and gradients, fp64, including an image whose
    determinant falls under the two-unknown threshold and one with bright (masked) pixels."""
    from inverserenderingofindoorscene_amd.losses import _lsregress_diffspec_live
    g = torch.Generator().manual_seed(9)
    nb, R_, C_ = 4, 6, 8
    d = torch.rand(nb, 3, R_, C_, generator=g, dtype=torch.float64)
    s = torch.rand(nb, 3, R_, C_, generator=g, dtype=torch.float64) * 0.5
    im = torch.rand(nb, 3, R_, C_, generator=g, dtype=torch.float64) * 1.1        # some pixels >= 0.9: masked out
    s[1] = d[1] * 0.3                                                             # collinear columns: determinant 0 -> one-unknown fallback
    d[2] *= 1e-2; s[2] *= 1e-2                                                    # tiny Gram matrix: floors active
    args_a = [t.clone().requires_grad_(True) for t in (d, s)]
    args_b = [t.clone().requires_grad_(True) for t in (d, s)]
    orig_a = [t.clone().requires_grad_(True) for t in (d, s)]
    orig_b = [t.clone().requires_grad_(True) for t in (d, s)]
    oa = _lsregress_diffspec_live(args_a[0], args_a[1], im, orig_a[0], orig_a[1])
    ob = O.lsregress_diffspec(args_b[0], args_b[1], im, orig_b[0], orig_b[1])
    w = [torch.randn(nb, 3, R_, C_, generator=g, dtype=torch.float64) for _ in range(2)]
    for x, y in zip(oa, ob):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12)
    ga = torch.autograd.grad((oa[0] * w[0]).sum() + (oa[1] * w[1]).sum(), args_a + orig_a)
    gb = torch.autograd.grad((ob[0] * w[0]).sum() + (ob[1] * w[1]).sum(), args_b + orig_b)
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, rtol=1e-8, atol=1e-10)
    assert ga[0].abs().max() > 0 and ga[1].abs().max() > 0        # the coefficients do carry gradient
