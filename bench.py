#!/usr/bin/env python
"""Headline benchmark: shaded Mpix/s (forward + backward) of the SG x microfacet render layer.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM
(trainLight mode):

    fused forward  (sgr_fused_fwd: SG -> env image written, diffuse, specular)
    fused backward (sgr_fused_bwd_sg: dense env cotangent + diffuse/specular cotangents -> SG grads)

through the package's autograd functions (i.e. through the drop-in boundary, not around it).
Workload = BASELINE.json configs[1] per GPU: batch 16, 240x320 BRDF maps, 120x160 env grid,
12 SG lobes, 8x16 directions.  For N > 1 the driver launches one rank per GPU with torchrun;
images shard across ranks (weak scaling: 16 images per GPU) and the timed step additionally runs the
render loss (LSregressDiffSpec + masked L2, wrapperBRDFLight.py:170-207) with its only collective --
the all-reduce of the loss numerator/denominator pair over RCCL -- between
forward and backward, so the N-GPU number contains the exchange north_star names.  The N = 1 line
carries the same with-loss step as `config.Mpix_per_s_with_render_loss` for a like-for-like ratio.

Timing: `--reps` (default 15) repetitions of the K-step loop, each bracketed by barrier + device
synchronise on both sides and maximised over ranks; the line reports the MEDIAN repetition
(`ms_per_step`) and lists all of them (`config.ms_per_step_repetitions`).

Informational legs (single process only; they can never fail the contract line): the cascade-0 light objective fused / unfused
(`ms_per_step_light_objective_*`), and BASELINE config 3 -- the synthetic trainLight step: decoder-head activations ->
light objective -> backward -> Adam over the 103 MB of decoder outputs (`config3`: eager, and the whole step replayed from a HIP
graph with the fused capturable Adam).  Round 2's HIP-graph replay legs of the two- and few-kernel steps are gone: with nothing but
long kernels in the step there is no launch gap to remove, and a replay pays ~10-40 us of graph-launch latency per step that
eager launches hide behind the running kernel (driver, round 2: 0.469 vs 0.422 ms).

`roofline` is the HBM roofline north_star names (algorithmic bytes / live kernel time); `roofline_valu` is the resource that
actually binds the fused kernels -- VALU issue -- from the SQ counters of the same workload (profiles/sq.json, tools/pmc_sq.sh).

Rank 0 prints ONE JSON line; see README/DESIGN.md for the field definitions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_shaded_px(K: int, J: int, q: int) -> dict:
    """BRDF maps 7q, SG params 7K, env image 3J, outputs 6 (x4 bytes)."""
    b_brdf, b_sg, b_env, b_out = 7 * q * 4, 7 * K * 4, 3 * J * 4, 6 * 4
    return dict(fwd_env=b_brdf + b_sg + b_env + b_out,          # fused forward writing the env image
                fwd_noenv=b_brdf + b_sg + b_out,
                bwd_sg=b_brdf + b_sg + b_out + b_env + b_sg,    # reads g_env (dense), g_d/g_s; writes SG grads
                )


def _claim_stdout():
    """The contract: rank 0 prints ONE JSON line.  RCCL prints a five-line banner ("RCCL version ... Librccl path ...") to the C-level stdout
    when its first communicator is created -- buffered, so it lands AFTER the line at exit -- and other native libraries may print as well.
    Everything this process writes to file descriptor 1 from here on goes to stderr; the returned function writes to the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        sys.stdout.flush()
        os.write(real, (line + "\n").encode())
    return emit


def main() -> None:
    emit = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=300, help="untimed steps first (0.15 s at config 2: the GPU's clocks ramp up over the first ~100 ms of load)")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--reps", type=int, default=15, help="repetitions of the K-step timed loop; the median is reported (fifteen: with a short --warmup the first three repetitions still run on ramping clocks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-only", action="store_true", help="skip the informational legs (two streams, HIP graph, light objective, baselines): profiling runs")
    ap.add_argument("--no-env", action="store_true", help="render-only variant (env image never materialised)")
    ap.add_argument("--graph-leg", action="store_true", help="with --layer-only: still time the with-render-loss step replayed from a HIP graph (sgr.capture_step) -- the batch sweep's launch-bound sizes")
    ap.add_argument("--no-config5", action="store_true", help="skip the compact BASELINE configs[4] leg of the default run")
    ap.add_argument("--pmc-workload", default="layer", choices=("layer", "objective", "objective_heads"),
                    help="counter-collection runs (tools/pmc_traffic.sh, tools/pmc_sq.sh): 'objective' / 'objective_heads' run ONLY the fused light objective "
                         "+ its backward (the trainLight step's kernels: fwd_pk*gt* and sg_bwd_recon_pk_kernel; _heads: decoder outputs in, premap 3) "
                         "for --warmup + --steps iterations and print a short record instead of the contract line")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="BASELINE.json configs index: 2 = headline (default); 5 = 480x640, SGNum 24, 16x32 stress (batch 4)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torchrun launch with WORLD_SIZE={args.gpus} (got {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render layer has no CPU path")
    # SGR_BENCH_BACKEND=gloo (rehearsal only): the N > 1 flow -- barriers, the sharded render loss in the timed step, max over ranks, rank 0's
    # line -- on a box with fewer GPUs than ranks: ranks share devices and the collectives go through the host.  Such a line says so
    # (`config.rehearsal`) and is no measurement; the driver's runs use the default, nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("SGR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:       # input generation is CPU work: do not oversubscribe the host with world x all-core thread pools
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    from oracle import sg_oracle as O      # bench-only: synthetic inputs + the cpu_baseline leg

    _lib.load()
    bn, imH, imW, R, C, K, eh, ew = args.batch, 240, 320, 120, 160, 12, 8, 16
    if args.config == 5:      # BASELINE configs[4]: env grid 240x320 assumed
        bn, imH, imW, R, C, K, eh, ew = (4 if args.batch == 16 else args.batch), 480, 640, 240, 320, 24, 16, 32
    J, q = eh * ew, (imH // R) * (imW // C)
    need_env = not args.no_env

    # different images on every rank (seed offset), generated on the CPU like SURVEY 8d prescribes
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=20202 + 1000 * rank)
    x = {k: v.to(dev) for k, v in inp.items()}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    g = torch.Generator().manual_seed(99 + rank)
    ct_env = (torch.randn((bn, 3, R, C, eh, ew), generator=g) * 1e-3).to(dev) if need_env else None
    ct_d = torch.randn((bn, 3, R, C), generator=g).to(dev)
    ct_s = torch.randn((bn, 3, R, C), generator=g).to(dev)
    layer = pkg.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    group = dist.group.WORLD if world > 1 else None

    if args.pmc_workload != "layer":      # counter-collection workload: the objective's kernels alone
        heads = args.pmc_workload == "objective_heads"
        ind1 = torch.ones(bn, 1, 1, 1, device=dev)
        if heads:
            gh = torch.Generator().manual_seed(7)
            sg = [(torch.randn(s, generator=gh) * 0.5).to(dev).requires_grad_(True) for s in ((bn, 3 * K, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
        else:
            sg = [x["axis"], x["lamb"], x["weight"]]
        for _ in range(args.warmup + args.steps):
            obj = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], x["im"], x["seg"], x["env_gt"], ind1, 1.0, 10.0,
                                      decoder_outputs=heads)[0]
            torch.autograd.grad(obj, sg)
        torch.cuda.synchronize()
        emit(json.dumps({"pmc_workload": args.pmc_workload, "config": args.config, "batch": bn, "iterations": args.warmup + args.steps}))
        return

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=need_env)
        if i is not None:
            ev[i][1].record()
        outs, cts = ([env, d, s], [ct_env, ct_d, ct_s]) if need_env else ([d, s], [ct_d, ct_s])
        grads = torch.autograd.grad(outs, [x["axis"], x["lamb"], x["weight"]], grad_outputs=cts)
        if i is not None:
            ev[i][2].record()
        return grads

    # the same step with the render loss in the loop (LSregressDiffSpec + masked L2 kernels, wrapperBRDFLight.py:170-207
    # around the layer); sharded: the all-reduce of [num, den] sits between forward and backward
    def step_with_loss(i=None):
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=need_env)
        err, _ = pkg.render_loss(d, s, x["im"], x["seg"], R, C, group=group)
        if need_env:
            grads = torch.autograd.grad([err, env], [x["axis"], x["lamb"], x["weight"]], grad_outputs=[None, ct_env])
        else:
            grads = torch.autograd.grad([err], [x["axis"], x["lamb"], x["weight"]])
        return grads

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, with_events=False):
        """EXACTLY args.steps steps between two barrier + synchronise brackets; seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn(i if with_events else None)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    def median(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    for _ in range(args.warmup):
        step()
        step_with_loss()

    reps = max(1, args.reps)
    fwd_acc = bwd_acc = 0.0
    plain_dts, loss_dts = [], []
    for _ in range(reps):
        plain_dts.append(timed(step, with_events=True))
        fwd_acc += sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
        bwd_acc += sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    for _ in range(reps):
        loss_dts.append(timed(step_with_loss))
    fwd_ms, bwd_ms = fwd_acc / reps, bwd_acc / reps
    plain_ms, loss_step_ms = median(plain_dts) / args.steps * 1e3, median(loss_dts) / args.steps * 1e3
    # the contract line: single GPU = the layer's forward + backward; sharded = the same plus the loss and its all-reduce
    headline_dts = plain_dts if world == 1 else loss_dts
    dt = median(headline_dts)

    # informational: the whole cascade-0 light objective (render loss + 10 x env reconstruction loss,
    # wrapperBRDFLight.py:167-207) -- fused (env image never written, sgr.light_objective) and unfused.  Single process only:
    # a leg that failed on one rank would leave the others in its barrier
    obj_ms = obj_unfused_ms = obj_fwd_only_ms = None
    cfg3 = None
    # The informational legs keep their own loop lengths: at least 60 timed iterations after 10 untimed ones, whatever --steps / --warmup say.
    # With the driver's `--steps 20 --warmup 5` a 20-iteration loop of a 0.5 ms step is 10 ms -- shorter than the ~100 ms the clocks take
    # to settle after the previous leg -- and read 3-8 % high (config 3: 0.78 vs 0.71 ms on one box).  The headline honours the flags exactly.
    leg_steps, leg_warmup = max(args.steps, 60), 10
    if world == 1 and need_env and not args.layer_only:
        ind = torch.ones(bn, 1, 1, 1, device=dev)

        def clear():
            for k in ("axis", "lamb", "weight"):
                x[k].grad = None

        def step_obj_fused():
            obj = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"],
                                      x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0)[0]
            obj.backward()
            clear()

        def step_obj_forward_only():      # evaluation loops (testLight.py drives the same wrapper without a backward): no gradient kernel is launched
            with torch.no_grad():
                pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0)

        def step_obj_unfused():
            env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
            err, _ = pkg.render_loss(d, s, x["im"], x["seg"], R, C)
            rec = pkg.recon_loss(env, x["env_gt"], x["seg"], ind, R, C)
            (err + 10.0 * rec).backward()
            clear()

        def loop_ms(fn, n):
            for _ in range(leg_warmup):
                fn()
            barrier()
            t2 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            return (time.perf_counter() - t2) / n * 1e3

        try:
            obj_ms = loop_ms(step_obj_fused, leg_steps)
            obj_unfused_ms = loop_ms(step_obj_unfused, leg_steps)
            obj_fwd_only_ms = loop_ms(step_obj_forward_only, leg_steps)
        except Exception as exc:       # informational legs: never fail the bench over them
            obj_ms = obj_unfused_ms = obj_fwd_only_ms = None
            print(f"# light-objective legs skipped: {str(exc)[:160]}", file=sys.stderr)

        # BASELINE config 3: the synthetic trainLight cascade-0 step (trainLight.py:203-244 around wrapperBRDFLight.py:158-207):
        # learnable decoder outputs -> output activations (sgr.light_heads) -> light objective -> backward -> Adam
        try:
            cfg3 = config3_legs(pkg, layer, x, ind, bn, R, C, K, leg_steps, barrier, leg_warmup)
        except Exception as exc:
            cfg3 = {"error": str(exc)[:200]}

    # informational: the with-render-loss step replayed from a HIP graph (sgr.capture_step): what a launch-bound caller -- the reference's
    # default batch of 5, trainLight.py:28 -- gets by capturing its step
    graph_loss_ms = None
    if world == 1 and (args.graph_leg or not args.layer_only):
        try:
            captured = pkg.capture_step(step_with_loss)
            captured.replay()
            barrier()
            t2 = time.perf_counter()
            for _ in range(max(args.steps, 60)):
                captured.replay()
            barrier()
            graph_loss_ms = (time.perf_counter() - t2) / max(args.steps, 60) * 1e3
            del captured
        except Exception as exc:
            print(f"# graph-replay leg skipped: {str(exc)[:160]}", file=sys.stderr)

    # informational, LAST (it creates and destroys a process group): the same with-loss step and the fused objective through RCCL with a
    # world of one rank -- init_process_group("nccl"), the all-reduces of the device-resident [num, den] vectors between the loss passes
    # (wrapperBRDFLight.py:192,205-207 under batch sharding) -- so the collectives' latency on the step is a measured number.
    # N > 1 is the driver's to run.
    rccl1 = None
    if world == 1 and need_env and not args.layer_only and args.config == 2:
        try:
            rccl1 = rccl_world1_legs(pkg, layer, x, ct_env, R, C, leg_steps, dev)
        except Exception as exc:
            rccl1 = {"error": str(exc)[:200]}

    cfg5 = None
    if world == 1 and need_env and not args.layer_only and args.config == 2 and not args.no_config5:
        try:
            for k in list(x):
                x[k] = None      # config 2's inputs are done with: 1.2 GB back to the allocator before config 5's 6 GB
            ct_env = None
            torch.cuda.empty_cache()
            cfg5 = config5_leg(pkg, dev)
        except Exception as exc:
            cfg5 = {"error": str(exc)[:200]}

    if rank == 0:
        P = bn * R * C                      # shaded (env-grid) pixels per GPU per step
        img_px = bn * imH * imW
        ms_per_step = dt / args.steps * 1e3
        value = world * img_px / (dt / args.steps) / 1e6
        bpp = algorithmic_bytes_per_shaded_px(K, J, q)
        fwd_bytes = P * (bpp["fwd_env"] if need_env else bpp["fwd_noenv"])
        bwd_bytes = P * (bpp["bwd_sg"] if need_env else bpp["bwd_sg"] - 3 * J * 4)
        fwd_gbps = fwd_bytes / (fwd_ms * 1e-3) / 1e9
        bwd_gbps = bwd_bytes / (bwd_ms * 1e-3) / 1e9
        dom = ("bwd", bwd_ms, bwd_bytes, bwd_gbps) if bwd_ms >= fwd_ms else ("fwd", fwd_ms, fwd_bytes, fwd_gbps)
        # HBM bytes per launch of the dominant kernel from the PMC passes (tools/pmc_traffic.sh -> tools/parse_pmc.py ->
        # profiles/traffic.json), recorded per WORKLOAD (config, batch, env written or not): a figure measured on another
        # workload is not reported
        tkey = f"config{args.config}_batch{bn}_{'env' if need_env else 'noenv'}"
        want = [r"sg_bwd_pk_kernel<"] if dom[0] == "bwd" else [r"fwd_pk_half_kernel<", r"fwd_pk_kernel<"]
        traffic, dom_name, traffic_stale = pmc_traffic(tkey, want)
        if dom_name is None:
            dom_name = ("sg_bwd" if dom[0] == "bwd" else "fwd") + " kernel (no PMC record for this workload)"
        # the resource that actually binds the fused kernels: VALU issue.  SQ counters of the same workload (tools/pmc_sq.sh ->
        # profiles/sq.json): SQ_ACTIVE_INST_VALU counts, per SIMD quad, the cycles a VALU instruction is in flight; x 4 / SIMDs
        # against the kernel's duration in shader-clock cycles is the fraction of issue cycles used.  Round 5: the duration is the
        # per-shader-engine SQ_BUSY_CYCLES (capped at kernel_ms x 2.4 GHz), not GRBM_GUI_ACTIVE / XCDs, which under the counter mode
        # also counts the launch's pre- and post-amble and implied clocks above the part's maximum (tools/parse_sq.py).
        # Both records are OFFLINE (counter passes cannot run inside the timed loop) and stamped with the hash of the kernel sources
        # they were measured on: a record from other sources is reported as stale and does not decide `limited_by`
        valu = valu_roofline(tkey, want, dom[1])
        limited_by = limited(valu, dom[3] / HBM_PEAK_GBPS)
        mpix = lambda ms: round(world * img_px / (ms * 1e-3) / 1e6, 1)
        out = {
            "metric": "Mpix/s shaded (fwd+bwd), 240x320x12-SG render layer" if args.config == 2 else "Mpix/s shaded (fwd+bwd), 480x640x24-SG 16x32 render layer (stress config)",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{1 if args.config == 2 else 4}] per GPU: batch {bn} x {imH}x{imW} BRDF maps -> "
                                   f"{R}x{C} env grid, SGNum={K}, {eh}x{ew} directions; fused fwd "
                                   f"({'env image written' if need_env else 'render only'}) + fused bwd (SG grads), trainLight mode",
                       "shaded_px_per_step_per_gpu": P, "image_px_per_step_per_gpu": img_px, "q": q,
                       "Mshade_per_s": round(world * P / (dt / args.steps) / 1e6, 1),
                       "timed_step": "fwd + bwd of the layer" if world == 1 else "fwd + render loss (all-reduce of [num, den] over RCCL) + bwd",
                       "repetitions": reps, "statistic": "median repetition of the K-step loop, max over ranks",
                       "ms_per_step_repetitions": [round(t / args.steps * 1e3, 4) for t in headline_dts],
                       "ms_per_step_layer_only": round(plain_ms, 4), "Mpix_per_s_layer_only": mpix(plain_ms),
                       "ms_per_step_with_render_loss": round(loss_step_ms, 4), "Mpix_per_s_with_render_loss": mpix(loss_step_ms),
                       "ms_per_step_with_render_loss_graph_replay": None if graph_loss_ms is None else round(graph_loss_ms, 4),
                       "Mpix_per_s_with_render_loss_graph_replay": None if graph_loss_ms is None else mpix(graph_loss_ms),
                       "informational_legs": {"timed_iterations": max(args.steps, 60), "untimed_iterations": 10,
                                              "note": "objective / config-3 / RCCL / graph legs: their own loop lengths, independent of --steps / --warmup (the headline honours the flags exactly)"},
                       "rccl_world1": rccl1,
                       "config5": cfg5,
                       "ms_per_step_light_objective_fused": None if obj_ms is None else round(obj_ms, 4),
                       "ms_per_step_light_objective_unfused": None if obj_unfused_ms is None else round(obj_unfused_ms, 4),
                       "ms_per_step_light_objective_forward_only": None if obj_fwd_only_ms is None else round(obj_fwd_only_ms, 4),
                       "config3": cfg3,
                       "parallelism": f"batch-sharded x{world}",
                       "rehearsal": None if backend == "nccl" else f"backend {backend}, {ndev} GPU(s) for {world} ranks: the N > 1 flow only, not a measurement"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(dom[3], 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(dom[3] / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_record_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": dom[2], "avg_launch_ms": round(dom[1], 4),
                         "limited_by": limited_by},
            "roofline_valu": valu,
            "kernels": {"forward (sgr_fused_fwd)": {"ms": round(fwd_ms, 4), "GBps": round(fwd_gbps, 1), "frac": round(fwd_gbps / HBM_PEAK_GBPS, 4),
                                                       "bytes": fwd_bytes},
                        "backward (sgr_fused_bwd_sg)": {"ms": round(bwd_ms, 4), "GBps": round(bwd_gbps, 1), "frac": round(bwd_gbps / HBM_PEAK_GBPS, 4),
                                                           "bytes": bwd_bytes}},
        }
        if world == 1 and not args.no_cpu_baseline and not args.layer_only and args.config == 2:
            out["cpu_baseline"] = cpu_baseline(O, 240, 320, 120, 160, 12, 8, 16)
            # port vs the UNMODIFIED reference on the same cores, measured where the reference exists (the authoring
            # container; oracle/calibrate_port_vs_reference.py -> profiles/cpu_calibration.json)
            try:
                out["cpu_baseline"]["calibration"] = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
            except Exception:
                out["cpu_baseline"]["calibration"] = None
            try:
                out["eager_gpu_baseline"] = eager_gpu_baseline(O, dev, 240, 320, 120, 160, 12, 8, 16)
            except Exception as exc:       # informational leg: never fail the bench over it
                out["eager_gpu_baseline"] = {"error": str(exc)[:200]}
        emit(json.dumps(out))

    if world > 1:
        dist.destroy_process_group()


def rccl_world1_legs(pkg, layer, x, ct_env, R, C, steps, dev) -> dict:
    """The sharded code path over RCCL with ONE rank (single-process bench only): `render_loss(group=WORLD)` between the layer's forward and
    backward, and `light_objective(group=WORLD)` (three stage operators, two all-reduces) -- against the same steps without a group, timed
    back to back.  What it measures is the collectives' enqueue + latency on the critical path; the wire (xGMI) is not involved."""
    import socket
    created = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        sg = [x["axis"], x["lamb"], x["weight"]]
        ind = torch.ones(x["albedo"].shape[0], 1, 1, 1, device=dev)

        def with_loss(group):
            env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], need_env=True)
            err, _ = pkg.render_loss(d, s, x["im"], x["seg"], R, C, group=group)
            torch.autograd.grad([err, env], sg, grad_outputs=[None, ct_env])

        def objective(group):
            obj = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0, group=group)[0]
            torch.autograd.grad(obj, sg)

        def loop_ms(fn, group):
            for _ in range(5):
                fn(group)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn(group)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        W = dist.group.WORLD
        out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        # alternate the two routes so that both see the same clocks
        a = [loop_ms(with_loss, None), loop_ms(with_loss, W), loop_ms(with_loss, None), loop_ms(with_loss, W)]
        b = [loop_ms(objective, None), loop_ms(objective, W), loop_ms(objective, None), loop_ms(objective, W)]
        out["ms_per_step_with_render_loss"] = round(min(a[0], a[2]), 4)
        out["ms_per_step_with_render_loss_sharded_world1"] = round(min(a[1], a[3]), 4)
        out["ms_per_step_light_objective"] = round(min(b[0], b[2]), 4)
        out["ms_per_step_light_objective_sharded_world1"] = round(min(b[1], b[3]), 4)
        out["note"] = ("one rank over the nccl (= RCCL) backend: one all-reduce of [num, den] per render loss, two per light objective, on device tensors; "
                       "N > 1 has never been timed by the builder (one GPU per box) -- the scaling curve is the driver's")
        return out
    finally:
        if created:
            dist.destroy_process_group()


def config5_leg(pkg, dev, steps=25, warmup=12, reps=3) -> dict:
    """BASELINE configs[4] (480x640 -> 240x320 env grid assumed, SGNum 24, 16x32 directions, batch 4) as a compact leg of the default run:
    the layer's forward + backward with per-kernel HBM rooflines from events on the launch stream, plus the offline counter records of
    that workload (profiles/traffic.json / sq.json, stamped with the kernel sources they were measured on).  Inputs come from the device
    generator (same distributions as SURVEY.md 8d; the parity of config 5 is tests/test_gpu_fullsize.py's business, not this leg's)."""
    bn, imH, imW, R, C, K, eh, ew = 4, 480, 640, 240, 320, 24, 16, 32
    J, q = eh * ew, 4
    g = torch.Generator(device=dev).manual_seed(20205)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    albedo, rough = rnd(bn, 3, imH, imW), rnd(bn, 1, imH, imW) * 2.0 - 1.0
    n = rn(bn, 3, imH, imW)
    n[:, 2] = n[:, 2].abs() + 0.5
    normal = n / n.norm(dim=1, keepdim=True)
    a = rn(bn, K, 3, R, C)
    sg = [(a / a.norm(dim=2, keepdim=True)).requires_grad_(True), rnd(bn, K, R, C).requires_grad_(True), rnd(bn, 3 * K, R, C).requires_grad_(True)]
    cts = [rn(bn, 3, R, C, eh, ew) * 1e-3, rn(bn, 3, R, C), rn(bn, 3, R, C)]
    del n, a
    layer = pkg.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        outs = layer.forwardSG(albedo, normal, rough, sg[0], sg[1], sg[2], need_env=True)
        if i is not None:
            ev[i][1].record()
        torch.autograd.grad(list(outs), sg, grad_outputs=cts)
        if i is not None:
            ev[i][2].record()

    for _ in range(warmup):
        step()
    dts, fwd, bwd = [], 0.0, 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / steps)
        fwd += sum(e[0].elapsed_time(e[1]) for e in ev) / steps / reps
        bwd += sum(e[1].elapsed_time(e[2]) for e in ev) / steps / reps
    dt = sorted(dts)[len(dts) // 2]
    P, bpp = bn * R * C, algorithmic_bytes_per_shaded_px(K, J, q)
    tkey = f"config5_batch{bn}_env"
    out = {"workload": f"BASELINE configs[4]: batch {bn} x {imH}x{imW} -> {R}x{C} env grid (assumed, SURVEY.md 8d), SGNum={K}, {eh}x{ew} directions; fused fwd (env written) + fused bwd (SG grads)",
           "value": round(bn * imH * imW / dt / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "warmup": warmup, "repetitions": reps,
           "data": "synthetic (device generator)", "kernels": {}}
    for tag, ms, nbytes, pats in (("forward (sgr_fused_fwd)", fwd, P * bpp["fwd_env"], [r"fwd_pk_half_kernel<"]), ("backward (sgr_fused_bwd_sg)", bwd, P * bpp["bwd_sg"], [r"sg_bwd_pk_kernel<"])):
        gbps = nbytes / (ms * 1e-3) / 1e9
        traffic, name, t_stale = pmc_traffic(tkey, pats)
        valu = valu_roofline(tkey, pats, ms)
        out["kernels"][tag] = {"roofline": {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                                            "traffic": traffic, "traffic_record_stale": t_stale, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 4),
                                            "limited_by": limited(valu, gbps / HBM_PEAK_GBPS)},
                               "roofline_valu": valu}
    return out


def csrc_sha16() -> str:
    """Hash of the kernel sources the loaded library was built from (same function as tools/parse_sq.py / parse_pmc.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "inverserenderingofindoorscene_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inl")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _records(fname, tkey):
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.isfile(path):
        return {}, None
    try:
        recs = json.load(open(path)).get(tkey, {})
    except Exception:
        return {}, None
    stamp = recs.get("_csrc_sha16")
    stale = None if stamp is None else (stamp != csrc_sha16())      # None: an unstamped record from before round 4
    return {k: v for k, v in recs.items() if not k.startswith("_")}, stale


def pmc_traffic(tkey, patterns):
    """HBM bytes per launch of the first kernel matching `patterns` from the PMC passes (tools/pmc_traffic.sh -> tools/parse_pmc.py ->
    profiles/traffic.json), recorded per WORKLOAD: a figure measured on another workload is not reported.  -> (bytes, kernel, stale)"""
    import re
    recs, stale = _records("traffic.json", tkey)
    for pat in patterns:
        hits = sorted((n for n in recs if re.search("::" + pat, n)), key=lambda n: -recs[n]["hbm_bytes"])
        if hits:
            return recs[hits[0]]["hbm_bytes"], hits[0].split("::")[-1], stale
    return None, None, None


def valu_roofline(tkey, patterns, live_ms):
    """`roofline_valu` from profiles/sq.json[tkey] for the first kernel matching `patterns`."""
    import re
    recs, stale = _records("sq.json", tkey)
    for pat in patterns:
        hits = [n for n in recs if re.search("::" + pat, n)]
        if not hits:
            continue
        n = max(hits, key=lambda h: recs[h].get("valu_busy_cycles_per_simd", 0))
        r = recs[n]
        busy, total = r["valu_busy_cycles_per_simd"], r["kernel_cycles"]
        return {"bound": "valu_issue", "kernel": n.split("::")[-1], "achieved": round(busy), "peak": round(total),
                "unit": "SIMD issue cycles per launch (SQ_ACTIVE_INST_VALU x 4 / SIMDs vs " + r.get("kernel_cycles_source", "GRBM_GUI_ACTIVE / XCDs") + ")",
                "frac": round(busy / total, 4), "frac_at_max_clock": r.get("frac_at_max_clock"), "effective_clock_GHz": r.get("effective_clock_GHz"),
                "issue_rate_busy_cycles_per_us": r.get("issue_rate_per_us"), "per_wave_cycle_shares": r.get("per_wave_cycle_shares"),
                "valu_instructions_per_wave": r.get("valu_insts_per_wave"),
                "transcendental_share": r.get("trans_share"), "pmc_kernel_ms": r.get("kernel_ms"), "live_kernel_ms": round(live_ms, 4),
                "record": "offline", "record_stale": stale,
                "source": "profiles/sq.json (tools/pmc_sq.sh on this workload; counters are per launch, not re-measured live; "
                          "record_stale = the kernel sources changed since it was taken)"}
    return None


def limited(valu, hbm_frac) -> str:
    if valu is None or valu.get("frac") is None:
        return "hbm"
    if valu.get("record_stale"):
        return "hbm? (the VALU record is stale)"
    return "valu_issue" if valu["frac"] > hbm_frac else "hbm"


def kernel_ms_from_profiler(fn, n=10) -> dict:
    """Average device time (ms) per launch of every sgr:: kernel over n calls of fn (torch.profiler / roctracer timestamps)."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    out = {}
    for e in prof.key_averages():
        if "sgr::" in e.key and e.count:
            tot = getattr(e, "device_time_total", None)
            if tot is None:
                tot = getattr(e, "cuda_time_total", 0.0)
            out[e.key] = tot / e.count * 1e-3
    return out


def objective_rooflines(kernel_ms, tkey, P, K, J, q) -> dict:
    """`roofline` + `roofline_valu` of the two heavy kernels of the trainLight step (the fused objective): algorithmic bytes per launch
    (the forward reads the ground-truth env where the layer's writes the predicted one -- 2008 B per shaded pixel at
    config 2 -- and the backward reads it again where the layer's reads the cotangent -- 2344 B) over the live per-launch time."""
    import re
    bpp = algorithmic_bytes_per_shaded_px(K, J, q)
    out = {}
    for tag, pats, nbytes in (("objective_forward", [r"fwd_pk_half_gt_kernel<", r"fwd_pk_kernel<"], P * bpp["fwd_env"]),
                              ("objective_backward", [r"sg_bwd_recon_pk_kernel<"], P * bpp["bwd_sg"])):
        hit = None
        for pat in pats:
            names = [n for n in kernel_ms if re.search("::" + pat, n)]
            if names:
                hit = max(names, key=lambda n: kernel_ms[n])
                break
        if hit is None:
            continue
        ms = kernel_ms[hit]
        gbps = nbytes / (ms * 1e-3) / 1e9
        traffic, _, t_stale = pmc_traffic(tkey, pats)
        valu = valu_roofline(tkey, pats, ms)
        out[tag] = {"roofline": {"bound": "hbm", "kernel": hit.split("sgr::", 1)[-1].split("(")[0], "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_record_stale": t_stale,
                                 "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 4),
                                 "timing": "torch.profiler device time per launch over the config-3 loop", "limited_by": limited(valu, gbps / HBM_PEAK_GBPS)},
                    "roofline_valu": valu}
    return out


def config3_legs(pkg, layer, x, ind, bn, R, C, K, steps, barrier, warmup=10) -> dict:
    """BASELINE config 3 at config-2 shapes: decoder-output tensors as parameters, sgr.light_objective with the decoders' output
    activations as the prologue of its kernels (and, for comparison, sgr.light_heads as a standalone pass each way),
    backward, Adam (trainLight.py:178-181: lr 1e-4 scaled, betas (0.5, 0.999)).  Eager, and the whole step replayed from one
    HIP graph (fused + capturable Adam: the optimizer's ~30 foreach launches become one kernel, the ~25 small launches of
    the step lose their gaps)."""
    dev = x["albedo"].device
    g = torch.Generator().manual_seed(7)
    shapes = ((bn, 3 * K, R, C), (bn, K, R, C), (bn, 3 * K, R, C))
    out = {}
    for mode in ("eager", "eager_standalone_heads", "hipgraph"):
        params = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(dev)) for s in shapes]
        opt = torch.optim.Adam(params, lr=1e-3, betas=(0.5, 0.999), fused=True, capturable=(mode == "hipgraph"))
        prologue = mode != "eager_standalone_heads"

        def one():
            opt.zero_grad(set_to_none=True)
            if prologue:      # the decoders' output activations run inside the objective's two heavy kernels (premap 3)
                total = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], params[0], params[1], params[2], x["im"], x["seg"],
                                            x["env_gt"], ind, 1.0, 10.0, decoder_outputs=True)[0]
            else:             # round 2's route: a standalone pass each way
                axis, lam, w, _ = pkg.light_heads(params[0], params[1], params[2])
                total = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], axis, lam, w, x["im"], x["seg"], x["env_gt"], ind, 1.0, 10.0)[0]
            total.backward()
            opt.step()
            return total

        if mode != "hipgraph":
            for _ in range(warmup):
                one()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            barrier()
            out["ms_per_step_config3" if prologue else "ms_per_step_config3_standalone_heads"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
            # the two heavy kernels of this step against their rooflines (round 4: the kernels the real trainLight step runs, not the
            # layer's); per-launch device time from the profiler, counters from the PMC passes on the same workload
            try:
                J, q = layer.envHeight * layer.envWidth, (x["albedo"].shape[2] // R) * (x["albedo"].shape[3] // C)
                tag = f"config2_batch{bn}_objective" + ("_heads" if prologue else "")
                out["kernels_config3" if prologue else "kernels_config3_standalone_heads"] = objective_rooflines(kernel_ms_from_profiler(one), tag, bn * R * C, K, J, q)
            except Exception as exc:
                out["kernels_config3_error"] = str(exc)[:160]
        else:
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(3):
                        one()
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_total = one()
                graph.replay()
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps):
                    graph.replay()
                barrier()
                out["ms_per_step_config3_hipgraph"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
                out["objective_after_replays"] = round(float(static_total.item()), 6)
                del graph
            except Exception as exc:
                out["ms_per_step_config3_hipgraph"] = None
                out["hipgraph_error"] = str(exc)[:160]
    out["step"] = ("light_objective(decoder_outputs=True: heads as the kernels' prologue) -> backward -> Adam(fused) over 3 decoder-output tensors "
                   f"({sum(torch.Size(s).numel() for s in shapes) * 4 / 1e6:.0f} MB), batch {bn}")
    return out


MODES = ("forward_only", "fwd_bwd_sg", "fwd_bwd_all_grads")      # the three timings asked of every baseline


def _baseline_runner(fwd, x, cts):
    """`fwd(x) -> (env, diffuse, spec)`; returns {mode: callable} for the three modes (models.py:371-389 + 461-522 forward under no_grad;
    + autograd backward w.r.t. the SG parameters -- trainLight mode, the metric's; + w.r.t. the BRDF maps as well)."""
    sg = [x["axis"], x["lamb"], x["weight"]]
    brdf = [x["albedo"], x["normal"], x["rough"]]

    def forward_only():
        with torch.no_grad():
            fwd(x)

    def fwd_bwd_sg():
        for t in brdf:
            t.requires_grad_(False)
        torch.autograd.grad(list(fwd(x)), sg, grad_outputs=cts)

    def fwd_bwd_all():
        for t in brdf:
            t.requires_grad_(True)
        torch.autograd.grad(list(fwd(x)), sg + brdf, grad_outputs=cts)
        for t in brdf:
            t.requires_grad_(False)

    return dict(zip(MODES, (forward_only, fwd_bwd_sg, fwd_bwd_all)))


def _time_modes(runners, sync, budget_s, max_reps):
    """Best-of timing of each mode inside a wall-clock budget per mode (one warm-up; at least one timed run)."""
    out = {}
    for mode in MODES:
        fn = runners[mode]
        t0 = time.perf_counter()
        fn(); sync()
        warm = time.perf_counter() - t0
        times = [warm] if warm > 0.5 * budget_s else []      # pathological host: the warm-up is the sample
        t_end = time.perf_counter() + budget_s
        while len(times) < 1 or (time.perf_counter() < t_end and len(times) < max_reps):
            t0 = time.perf_counter()
            fn(); sync()
            times.append(time.perf_counter() - t0)
        out[mode] = (min(times), len(times))
    return out


def eager_gpu_baseline(O, dev, imH, imW, R, C, K, eh, ew) -> dict:
    """The torch port of the reference algorithm in the reference's own tensor formulation (whole-batch broadcast
    temporaries, oracle.render_from_sg_broadcast) run eagerly ON THE GPU through PyTorch-ROCm's aten kernels: the
    GPU-vs-GPU baseline BASELINE.md asks for (the reference itself cannot travel to the GPU box).  Bounded sample:
    four images; forward only, forward + backward (SG grads: the metric's mode, `value`), forward + backward (all six gradients)."""
    n = 4
    inp = O.synthetic_inputs(n, imH, imW, R, C, K, eh, ew, seed=20202)
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight")
    x = {k: inp[k].to(dev) for k in names}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    g = torch.Generator().manual_seed(99)
    cts = [(torch.randn((n, 3, R, C, eh, ew), generator=g) * 1e-3).to(dev), torch.randn((n, 3, R, C), generator=g).to(dev),
           torch.randn((n, 3, R, C), generator=g).to(dev)]
    fwd = lambda x_: O.render_from_sg_broadcast(x_["albedo"], x_["normal"], x_["rough"], x_["axis"], x_["lamb"], x_["weight"], eh, ew)
    res = _time_modes(_baseline_runner(fwd, x, cts), torch.cuda.synchronize, budget_s=2.0, max_reps=3)
    mp = lambda t: round(n * imH * imW / t / 1e6, 2)
    best = res["fwd_bwd_sg"][0]
    return {"value": mp(best), "unit": "Mpix/s", "kind": "port",
            "modes": {m: {"Mpix_per_s": mp(res[m][0]), "ms_per_image": round(res[m][0] / n * 1e3, 2), "runs": res[m][1]} for m in MODES},
            "sample": f"{n} images of the same workload, eager PyTorch-ROCm on this GPU running the torch port "
                      f"of the reference algorithm in its broadcast formulation (oracle.render_from_sg_broadcast), best of <= 3 per mode; "
                      f"value = fwd+bwd (SG grads), {best / n * 1e3:.2f} ms per image"}


def cpu_baseline(O, imH, imW, R, C, K, eh, ew) -> dict:
    """The reference's CPU path for the same step, timed on this box's host cores on a bounded sample: ONE image of the
    workload (1/16 of a step), fp32, in the three modes BASELINE.md section 3 names -- forward only, forward + backward (SG grads: the
    metric's mode, `value`), forward + backward (all six gradients).  Where the reference checkout is mounted (the authoring
    container) that is the UNMODIFIED models.output2env.output2env + models.renderingLayer.forwardEnv (kind "reference");
    on the GPU box, where it is not, the torch port in the reference's OWN tensor formulation -- whole-image broadcast
    temporaries, oracle.render_from_sg_broadcast (kind "port"; profiles/cpu_calibration.json: its speed against the reference's on the
    same cores, per mode; the bounded-memory per-lobe formulation the parity tests use is 1.5x slower and is not timed here)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))          # the port is memory-bound; more threads only oversubscribe
    torch.set_num_threads(cores)
    inp = O.synthetic_inputs(1, imH, imW, R, C, K, eh, ew, seed=20202)
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight")
    x = {k: inp[k].clone() for k in names}
    for k in ("axis", "lamb", "weight"):
        x[k].requires_grad_(True)
    g = torch.Generator().manual_seed(99)
    cts = [torch.randn((1, 3, R, C, eh, ew), generator=g) * 1e-3, torch.randn((1, 3, R, C), generator=g), torch.randn((1, 3, R, C), generator=g)]

    kind, what = "port", "torch fp32 CPU port of the reference algorithm in its broadcast formulation (oracle.render_from_sg_broadcast)"
    ref_layers = None
    try:
        from oracle import ref_import as RI
        if RI.available():
            ref_layers = RI.make_layers(K, R, C, eh, ew)
            kind, what = "reference", "the UNMODIFIED reference (models.output2env.output2env + models.renderingLayer.forwardEnv, imported from the mounted checkout)"
    except Exception:
        ref_layers = None

    def fwd(x_):
        if ref_layers is not None:
            env, _, _, _ = ref_layers[0].output2env(x_["axis"], x_["lamb"], x_["weight"])
            d, s = ref_layers[1].forwardEnv(x_["albedo"], x_["normal"], x_["rough"], env)
            return env, d, s
        return O.render_from_sg_broadcast(x_["albedo"], x_["normal"], x_["rough"], x_["axis"], x_["lamb"], x_["weight"], eh, ew)

    res = _time_modes(_baseline_runner(fwd, x, cts), lambda: None, budget_s=8.0, max_reps=4)      # <= ~30 s of CPU work in all
    best, runs = res["fwd_bwd_sg"]
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    mp = lambda t: round(imH * imW / t / 1e6, 4)
    return {"value": mp(best), "unit": "Mpix/s", "cores": torch.get_num_threads(), "kind": kind,
            "modes": {m: {"Mpix_per_s": mp(res[m][0]), "seconds_per_image": round(res[m][0], 3), "runs": res[m][1]} for m in MODES},
            "sample": f"1 image (1/16 of a step) of the same workload, {what}, best of <= 4 per mode; value = fwd+bwd (SG grads), best of {runs}; {best:.3f} s per image",
            "cpu": model}


if __name__ == "__main__":
    main()
