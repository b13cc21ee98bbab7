"""How much of a step is host work?  Enqueue time (python + dispatcher + ctypes, no synchronisation) against GPU time per step for
the layer-only step, the with-render-loss step and the fused objective (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import inverserenderingofindoorscene_amd as pkg
from oracle import sg_oracle as O

bn, imH, imW, R, C, K = 16, 240, 320, 120, 160, 12
dev = torch.device("cuda")
inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=1)
x = {k: v.to(dev) for k, v in inp.items()}
for k in ("axis", "lamb", "weight"):
    x[k].requires_grad_(True)
layer = pkg.renderingLayer(imWidth=C, imHeight=R)
ct_env = torch.randn((bn, 3, R, C, 8, 16), device=dev) * 1e-3
ct = torch.randn((bn, 3, R, C), device=dev)
ind = torch.ones(bn, 1, 1, 1, device=dev)
sg = [x["axis"], x["lamb"], x["weight"]]

def layer_step():
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)
    torch.autograd.grad([env, d, s], sg, grad_outputs=[ct_env, ct, ct])

def loss_step():
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)
    err, _ = pkg.render_loss(d, s, x["im"], x["seg"], R, C)
    torch.autograd.grad([err, env], sg, grad_outputs=[None, ct_env])

def obj_step():
    obj = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], *sg, x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0)[0]
    torch.autograd.grad(obj, sg)

for name, fn in (("layer", layer_step), ("layer + render loss", loss_step), ("fused objective", obj_step)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:22s} enqueue {1e3 * (t1 - t0) / n:.3f} ms/step   total {1e3 * (t2 - t0) / n:.3f} ms/step")

# where the enqueue time goes (round 4): the raw C-ABI launch through ctypes, the registered operator without and with an autograd
# node, the backward through torch.autograd.grad -- each enqueue-only, per call
from inverserenderingofindoorscene_amd import _lib
from inverserenderingofindoorscene_amd.ops import _dirs, _ptr, _stream, _view
env_b = torch.empty((bn, 3, R, C, 8, 16), device=dev); dif_b = torch.empty((bn, 3, R, C), device=dev); spc_b = torch.empty_like(dif_b)
d_tab, v_tab = _dirs(dev, 8, 16), _view(dev, R, C, 57.0)
raw_args = [_ptr(x[k]) for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")] + [_ptr(d_tab), _ptr(v_tab), _ptr(env_b), None, None, _ptr(dif_b), _ptr(spc_b),
                                                                                             bn, K, R, C, 8, 16, imH, imW, 0.05, 1, _stream(dev)]
lib = _lib.load()

def raw_launch():
    lib.sgr_fused_fwd_tan(*raw_args)

def op_nograd():
    with torch.no_grad():
        torch.ops.sgrender.fused_render(x["albedo"], x["normal"], x["rough"], *sg, 8, 16, 57.0, 0.05, (0.0, 0.0, 0.0), 1, True, False)

def op_grad():
    torch.ops.sgrender.fused_render(x["albedo"], x["normal"], x["rough"], *sg, 8, 16, 57.0, 0.05, (0.0, 0.0, 0.0), 1, True, False)

def fwd_wrapper():
    layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)

def empty5():
    for _ in range(5):
        torch.empty((bn, 3, R, C), device=dev)

# PyTorch's own floor for "one forward node, one backward through torch.autograd.grad with three outputs / cotangents": the same
# call pattern as layer_step, with aten kernels on 16-element tensors -- what the engine (graph task, hand-off to the device thread)
# and torch.autograd.grad's Python cost whatever the node does
tiny = [torch.randn(16, device=dev, requires_grad=True) for _ in range(3)]
tiny_ct = [torch.randn(16, device=dev) for _ in range(3)]

def torch_floor():
    outs = [t * 2.0 for t in tiny]
    torch.autograd.grad(outs, tiny, grad_outputs=tiny_ct)

def bwd_only():
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)
    return env, d, s

for name, fn in (("C ABI via ctypes (1 launch)", raw_launch), ("5 x torch.empty", empty5), ("operator, no_grad", op_nograd), ("operator, autograd node", op_grad),
                 ("layer.forwardSG", fwd_wrapper), ("torch floor: 3 aten mul + autograd.grad", torch_floor)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"   {name:30s} enqueue {1e6 * (t1 - t0) / n:7.1f} us/call")

if os.environ.get("SGR_HOST_PROFILE"):
    import cProfile, pstats
    for name, fn in (("layer", layer_step), ("layer + render loss", loss_step)):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        print("=====", name)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
