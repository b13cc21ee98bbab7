"""GPU: the RCCL path itself, on the one GPU a test box has.

``losses._sharded(group)`` is true for an explicit group whatever its size, so a world of ONE rank over the ``nccl`` backend
(= RCCL on ROCm) drives everything the 8-GPU run does except the wire: ``init_process_group("nccl")``, the all-reduce of the
device-resident ``[num, den]`` pair between the render loss's passes and its finalize operator (wrapperBRDFLight.py:192,205-207:
the normaliser is batch-global), the two all-reduces between the light objective's three stage operators (with and without the decoder
heads as the kernels' prologue), on the stream the kernels run on.  A sum over one rank is the identity, so the render loss, every image
output and EVERY GRADIENT must be BIT-identical to the unsharded call's -- and are.  The objective's two reported scalars (the reconstruction
term and the weighted sum) may differ by one rounding: the one-rank operator folds the batch totals and the scalar tail in one kernel, the
staged route in a stage operator's fold plus the finalize operator (1.1e-7 relative on two images of config 2, 0 on the small cases): held to
2e-7.  Runs in a subprocess: a process group must not outlive the test in the pytest process.
Round 6: the same again with the extension's OWN communicator (``sgr.enable_native_allreduce``: ``ncclGetUniqueId`` / ``ncclCommInitRank`` from
libsgrender_torch.so, ``ncclAllReduce`` enqueued on the current HIP stream by ``torch.ops.sgrender.allreduce_sum_``; SURVEY.md 8e's native
variant) -- ``dist.all_reduce`` is replaced by a function that raises while that route runs, and the results are held to the same bits.
tests/test_gpu_sharded.py covers world size 2 (gloo, host copies); N > 1 over RCCL has never been run -- it is the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SGR_ROOT"])
import inverserenderingofindoorscene_amd as sgr
from oracle import sg_oracle as O          # checker-side input generator only
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
out = {}
for case, (bn, imH, imW, R, C, K, eh, ew, heads) in {"k12_8x16": (3, 24, 32, 12, 16, 12, 8, 16, False), "k12_8x16_decoder_outputs": (3, 24, 32, 12, 16, 12, 8, 16, True),
                                                   "k24_16x32_ragged": (2, 10, 14, 5, 7, 24, 16, 32, False), "config2_image": (2, 240, 320, 120, 160, 12, 8, 16, True)}.items():
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=777)
    inp["seg"][0, :, : imH // 2] = 0.0
    ind = torch.ones(bn, 1, 1, 1); ind[-1] = 0.0
    if heads:
        g = torch.Generator().manual_seed(5)
        inp["axis"], inp["lamb"], inp["weight"] = (torch.randn(bn, 3 * K, R, C, generator=g), torch.randn(bn, K, R, C, generator=g), torch.randn(bn, 3 * K, R, C, generator=g))
    x = {k: v.to(dev) for k, v in inp.items()}
    ind = ind.to(dev)
    sg = [x[k].requires_grad_(True) for k in ("axis", "lamb", "weight")]
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    res = {}
    for tag, group in (("plain", None), ("rccl", dist.group.WORLD), ("native", dist.group.WORLD)):
        if tag == "native":      # round 6: the extension's own communicator -- ncclAllReduce enqueued on the current stream by
            # torch.ops.sgrender.allreduce_sum_; c10d's all_reduce must not be reached any more
            handle = sgr.enable_native_allreduce(dist.group.WORLD)
            assert sgr.native_allreduce_enabled(dist.group.WORLD) and sgr.enable_native_allreduce(None) == handle
            assert torch.ops.sgrender.comm_world_size(handle) == 1
            probe = torch.tensor([1.5, -2.0, 3.25], device=dev); probe64 = probe.double()
            torch.ops.sgrender.allreduce_sum_(probe, handle); torch.ops.sgrender.allreduce_sum_(probe64, handle)
            assert probe.tolist() == [1.5, -2.0, 3.25] and probe64.tolist() == [1.5, -2.0, 3.25]
            c10d_all_reduce = dist.all_reduce
            def _refuse(*a, **k):
                raise AssertionError("dist.all_reduce reached although the group has a native communicator")
            dist.all_reduce = _refuse
        if heads:
            a, l, w, _ = sgr.light_heads(*sg)
        else:
            a, l, w = sg
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], a, l, w, need_env=True)
        err, rendered = sgr.render_loss(d, s, x["im"], x["seg"], R, C, group=group)
        g_loss = torch.autograd.grad(err, sg)
        obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0,
                                  group=group, decoder_outputs=heads)
        g_obj = torch.autograd.grad(obj[0], sg)
        with torch.no_grad():      # the forward-only route through the collectives as well
            obj_ng = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0,
                                         group=group, decoder_outputs=heads)
        torch.cuda.synchronize()
        res[tag] = dict(err=err, rendered=rendered, g_loss=g_loss, obj=obj, g_obj=g_obj, obj_ng=obj_ng)
        if tag == "native":
            dist.all_reduce = c10d_all_reduce
            sgr.disable_native_allreduce(dist.group.WORLD)
            assert not sgr.native_allreduce_enabled(None)
    for route in ("rccl", "native"):
        p, r = res["plain"], res[route]
        rec = {}
        rec["render_err_equal"] = bool(torch.equal(p["err"], r["err"]))
        rec["rendered_equal"] = bool(torch.equal(p["rendered"], r["rendered"]))
        rec["render_grads_equal"] = all(bool(torch.equal(a_, b_)) for a_, b_ in zip(p["g_loss"], r["g_loss"]))
        rec["obj_render_err_equal"] = bool(torch.equal(p["obj"][1], r["obj"][1]))
        rec["obj_rendered_equal"] = bool(torch.equal(p["obj"][3], r["obj"][3])) and bool(torch.equal(p["obj"][4], r["obj"][4]))
        rec["obj_grads_equal"] = all(bool(torch.equal(a_, b_)) for a_, b_ in zip(p["g_obj"], r["g_obj"]))
        relt = lambda a_, b_: float((a_.double() - b_.double()).norm() / b_.double().norm().clamp_min(1e-300))
        rec["render_grads_rel"] = max(relt(a_, b_) for a_, b_ in zip(r["g_loss"], p["g_loss"]))
        rec["obj_grads_rel"] = max(relt(a_, b_) for a_, b_ in zip(r["g_obj"], p["g_obj"]))
        rec["rendered_rel"] = max(relt(r["rendered"], p["rendered"]), relt(r["obj"][3], p["obj"][3]), relt(r["obj"][4], p["obj"][4]))
        rec["render_err_rel"] = max(abs(float(r["err"]) - float(p["err"])) / abs(float(p["err"])), abs(float(r["obj"][1]) - float(p["obj"][1])) / abs(float(p["obj"][1])))
        rec["obj_grads_finite_nonzero"] = all(bool(torch.isfinite(t).all()) and float(t.abs().max()) > 0 for t in r["g_obj"])
        rel = lambda a_, b_: abs(float(a_) - float(b_)) / max(abs(float(b_)), 1e-30)
        rec["obj_rel"] = rel(r["obj"][0], p["obj"][0]); rec["recon_rel"] = rel(r["obj"][2], p["obj"][2])
        rec["obj_ng_rel"] = rel(r["obj_ng"][0], p["obj_ng"][0]); rec["obj_ng_vs_grad_rel"] = rel(r["obj_ng"][0], r["obj"][0])
        out[case + "/" + route] = rec
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_render_loss_and_light_objective_through_rccl_world_of_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), SGR_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    out = json.loads(line[-1][7:])
    assert len(out) == 8 and sum(k.endswith("/native") for k in out) == 4
    print(json.dumps(out))
    for case, rec in out.items():
        assert rec["obj_grads_finite_nonzero"] is True, (case, rec)
        for k in ("render_err_equal", "rendered_equal", "render_grads_equal", "obj_render_err_equal", "obj_rendered_equal", "obj_grads_equal"):
            assert rec[k] is True, (case, k, rec)
        for k in ("render_err_rel", "render_grads_rel", "obj_grads_rel", "rendered_rel"):
            assert rec[k] == 0.0, (case, k, rec)
        for k in ("obj_rel", "recon_rel", "obj_ng_rel"):
            assert rec[k] <= 2e-7, (case, k, rec)
        assert rec["obj_ng_vs_grad_rel"] <= 2e-6, (case, rec)
