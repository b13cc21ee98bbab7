"""Does the forward's last partial round leave room for the render loss's first pass?  (round-3 review, item 8; development tool)

The default forward (`fwd_pk_half_kernel`, 9600 work units over 3072 wave slots) ends with ~1/6 of its span at 0.27-0.9 of the
SIMDs busy (`profiles/r03k_wavetrace_report.txt`).  `loss_stage_a` reads the forward's diffuse / specular images, so it cannot move
ahead of it as it is; what CAN is its first half -- pooling the full-resolution image and mask to the env grid (19.7 MB in,
4.9 MB out at config 2), the part that is cold in a training loop.  This script measures the UPPER BOUND of splitting that half off
onto a second stream before committing to the split: the forward alone, the forward followed by an equivalent pooling pass on the
same stream, and the two on two streams (fork before the forward, join after it), in a loop that evicts nothing in between (the
most favourable case for the overlap).  All three on the same box, interleaved.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import inverserenderingofindoorscene_amd as pkg
from oracle import sg_oracle as O

bn, imH, imW, R, C, K = 16, 240, 320, 120, 160, 12
dev = torch.device("cuda")
inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=1)
x = {k: v.to(dev) for k, v in inp.items()}
layer = pkg.renderingLayer(imWidth=C, imHeight=R)
sg = [x["axis"], x["lamb"], x["weight"]]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def fwd():
    with torch.no_grad():
        return layer.forwardSG(x["albedo"], x["normal"], x["rough"], *sg, need_env=True)


def pool():
    return F.avg_pool2d(x["im"], 2), F.avg_pool2d(x["seg"], 2)


def serial():
    r = fwd()
    return r, pool()


def forked():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        p = pool()
    r = fwd()
    main.wait_stream(side)
    return r, p


def loss_serial():
    with torch.no_grad():
        env, d, s = fwd()
        return pkg.render_loss(d, s, x["im"], x["seg"], R, C)


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


res = {}
for rep in range(3):
    for name, fn in (("forward", fwd), ("pool", pool), ("forward; pool (one stream)", serial), ("forward || pool (two streams)", forked),
                     ("forward; render loss (product)", loss_serial)):
        res.setdefault(name, []).append(timed(fn))
for name, v in res.items():
    print(f"{name:34s} " + "  ".join(f"{t:7.1f}" for t in v) + "   us per iteration")
a, b, c = (min(res[k]) for k in ("forward", "forward; pool (one stream)", "forward || pool (two streams)"))
print(f"upper bound of the split: serial - forked = {b - c:.1f} us of {b:.1f} (forward alone {a:.1f})")
