"""How fast is the CPU port (oracle/sg_oracle.py; its broadcast formulation ``render_from_sg_broadcast`` is what bench.py's ``cpu_baseline`` times on the GPU box) against the
UNMODIFIED reference on the same cores?  TEST INFRASTRUCTURE ONLY, authoring container (the reference is not on the GPU box):

    python -m oracle.calibrate_port_vs_reference      # writes profiles/cpu_calibration.json

One image of BASELINE config 2 (240x320 -> 120x160, SGNum 12, 8x16) in the three modes BASELINE.md section 3 names -- forward only,
forward + backward w.r.t. the SG parameters (the metric's mode), forward + backward w.r.t. all six inputs -- with the bench's cotangents,
fp32, all cores of this container, best of 3 after one warm-up: the same protocol as ``bench.py: cpu_baseline``.  The reference is models.output2env.output2env + models.renderingLayer.forwardEnv
(models.py:391-404, 461-522) exactly as wrapperBRDFLight.py:177,194 call them."""
from __future__ import annotations

import json
import os
import time

import torch

from oracle import ref_import as RI
from oracle import sg_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


MODES = ("forward_only", "fwd_bwd_sg", "fwd_bwd_all_grads")      # BASELINE.md section 3 (same names as bench.py)


def main():
    if not RI.available():
        raise SystemExit("reference not mounted")
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    imH, imW, R, C, K, eh, ew = 240, 320, 120, 160, 12, 8, 16
    inp = O.synthetic_inputs(1, imH, imW, R, C, K, eh, ew, seed=20202)
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight")
    x = {k: inp[k].clone() for k in names}
    sg = [x[k].requires_grad_(True) for k in ("axis", "lamb", "weight")]
    brdf = [x[k] for k in ("albedo", "normal", "rough")]
    g = torch.Generator().manual_seed(99)
    cts = [torch.randn((1, 3, R, C, eh, ew), generator=g) * 1e-3, torch.randn((1, 3, R, C), generator=g), torch.randn((1, 3, R, C), generator=g)]
    o2e, rl = RI.make_layers(K, R, C, eh, ew)

    def reference():
        env, _, _, _ = o2e.output2env(x["axis"], x["lamb"], x["weight"])
        d, s = rl.forwardEnv(x["albedo"], x["normal"], x["rough"], env)
        return [env, d, s]

    def port():               # the bounded-memory per-lobe formulation the parity tests use
        return list(O.render_from_sg(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew))

    def port_broadcast():     # the reference's own tensor formulation (whole-image broadcast temporaries): what bench.py's cpu_baseline times
        return list(O.render_from_sg_broadcast(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew))

    def timed(fwd, mode):
        def run():
            if mode == "forward_only":
                with torch.no_grad():
                    fwd()
                return
            wrt = sg + (brdf if mode == "fwd_bwd_all_grads" else [])
            for t in brdf:
                t.requires_grad_(mode == "fwd_bwd_all_grads")
            torch.autograd.grad(fwd(), wrt, grad_outputs=cts)
        t = best_of(run)
        for t_ in brdf:
            t_.requires_grad_(False)
        return t

    res = {m: {"reference_seconds": round(timed(reference, m), 3), "port_seconds": round(timed(port, m), 3), "broadcast_port_seconds": round(timed(port_broadcast, m), 3)}
           for m in MODES}
    for m in MODES:
        r = res[m]
        r["reference_Mpix_per_s"] = round(imH * imW / r["reference_seconds"] / 1e6, 4)
        r["broadcast_port_Mpix_per_s"] = round(imH * imW / r["broadcast_port_seconds"] / 1e6, 4)
        r["broadcast_port_over_reference_speed"] = round(r["reference_seconds"] / r["broadcast_port_seconds"], 3)
        r["port_over_reference_speed"] = round(r["reference_seconds"] / r["port_seconds"], 3)
    model = "unknown"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    sgm = res["fwd_bwd_sg"]
    out = {"where": "authoring container (no GPU; /root/reference mounted)", "cpu": model, "cores": cores, "torch": torch.__version__,
           "sample": "1 image of BASELINE configs[1] (240x320 -> 120x160, SGNum 12, 8x16), fp32, best of 3 per mode",
           "modes": res,
           # the metric's mode (fwd + bwd w.r.t. the SG parameters) at the top level, as rounds 3-4 recorded it
           "reference_seconds": sgm["reference_seconds"], "port_seconds": sgm["port_seconds"],
           "reference_Mpix_per_s": sgm["reference_Mpix_per_s"], "port_Mpix_per_s": round(imH * imW / sgm["port_seconds"] / 1e6, 4),
           "port_over_reference_speed": sgm["port_over_reference_speed"],
           "broadcast_port_seconds": sgm["broadcast_port_seconds"], "broadcast_port_Mpix_per_s": sgm["broadcast_port_Mpix_per_s"],
           "broadcast_port_over_reference_speed": sgm["broadcast_port_over_reference_speed"],
           "note": "bench.py's cpu_baseline times the BROADCAST port (kind 'port'), or the unmodified reference itself where it is mounted (kind 'reference')"}
    path = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
