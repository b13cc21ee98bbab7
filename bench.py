#!/usr/bin/env python
"""Headline benchmark: shaded Mpix/s (forward + backward) of the SG x microfacet render layer.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM
(trainLight mode):

    fused forward  (sgr_fused_fwd: SG -> env image written, diffuse, specular)
    fused backward (sgr_fused_bwd_sg: dense env cotangent + diffuse/specular cotangents -> SG grads)

through the package's autograd functions (i.e. through the drop-in boundary, not around it).
Workload = BASELINE.json configs[1] per GPU: batch 16, 240x320 BRDF maps, 120x160 env grid,
12 SG lobes, 8x16 directions.

N > 1 (round 6: launch-proof).  `python bench.py --gpus N` WITHOUT torchrun re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); under
torchrun (the driver's form) it uses the RANK / LOCAL_RANK / WORLD_SIZE it finds.  Images shard across ranks (weak scaling:
--batch images per GPU, default 16) and the timed step additionally runs the render loss (LSregressDiffSpec + masked L2,
wrapperBRDFLight.py:170-207) with its only collective -- the all-reduce of the loss numerator / denominator pair over RCCL
 -- between forward and backward, so the N-GPU number contains the exchange north_star names.
Like-for-like anchors, flat in `config` at EVERY N: `Mpix_layer_only` (the N = 1 headline's step, no collective) and
`Mpix_with_loss` (the N > 1 headline's step); the N = 1 line also carries `scaling_anchor_Mpix_per_s` (= its with-loss figure,
what an N-GPU `value` divides by) and leaves it in `.bench_anchor.json` for the N > 1 runs that follow on the same box.
N > 1 legs (all ranks): `cfg4_*` = BASELINE configs[3] (8 images per GPU, batch 64 at N = 8), `strong16_*` = 16 images split
N ways (SURVEY 8d strong scaling), `obj_ms` = the sharded light objective, `n_ranks_seen` = an RCCL all-reduce of ones.

Timing: `--reps` (default 15) repetitions of the K-step loop, each bracketed by barrier + device
synchronise on both sides and maximised over ranks; the line reports the MEDIAN repetition (`ms_per_step`).

THE LINE (round 6): <= 6 KB, `config` holds flat scalars with keys <= 40 characters and strings <= 120 (what the driver's
record keeps); every nested object of rounds 1-5 (per-kernel rooflines, VALU records, per-mode baselines, repetitions) goes to
`bench_detail.json` beside this file (`config.detail_file`).  tests/test_bench_contract.py asserts the limits.

Informational legs (single process only; they can never fail the contract line): the cascade-0 light objective fused / unfused /
forward-only, BASELINE config 3 (the synthetic trainLight step incl. Adam; eager and HIP-graph replay), config 5, RCCL with a
world of one (c10d route and the extension's in-stream ncclAllReduce), and the COLD column: the layer step and the objective step
rotating through four input sets (2.4 GB, ten times the 256 MB Infinity Cache) -- what a loader handing over fresh batches sees.

`roofline` is the HBM roofline north_star names (algorithmic bytes / live kernel time); `roofline_valu` is the resource that
actually binds the fused kernels -- VALU issue -- from the SQ counters of the same workload (profiles/sq.json, tools/pmc_sq.sh).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LINE_LIMIT = 6000        # bytes of the contract line (the driver's record keeps an 8.7 KB tail)
KEY_LIMIT, STR_LIMIT = 40, 120
DETAIL_FILE = os.environ.get("SGR_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
ANCHOR_FILE = os.path.join(ROOT, ".bench_anchor.json")


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=300, help="untimed steps first (0.15 s at config 2: the GPU's clocks ramp up over the first ~100 ms of load)")
    ap.add_argument("--batch", "--per-gpu-batch", dest="batch", type=int, default=16, help="images per GPU (weak scaling; BASELINE configs[3] = 8)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --batch is the TOTAL, split evenly over the ranks (SURVEY.md 8d config 4: 16 images 2/4/8 ways)")
    ap.add_argument("--reps", type=int, default=15, help="repetitions of the K-step timed loop; the median is reported (fifteen: with a short --warmup the first three repetitions still run on ramping clocks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-only", action="store_true", help="skip the informational legs (light objective, config 3 / 4 / 5, cold column, HIP graph, baselines): profiling and rehearsal runs")
    ap.add_argument("--no-env", action="store_true", help="render-only variant (env image never materialised)")
    ap.add_argument("--graph-leg", action="store_true", help="with --layer-only: still time the with-render-loss step replayed from a HIP graph (sgr.capture_step) -- the batch sweep's launch-bound sizes")
    ap.add_argument("--no-config5", action="store_true", help="skip the compact BASELINE configs[4] leg of the default run")
    ap.add_argument("--pmc-workload", default="layer", choices=("layer", "objective", "objective_heads"),
                    help="counter-collection runs (tools/pmc_traffic.sh, tools/pmc_sq.sh): 'objective' / 'objective_heads' run ONLY the fused light objective "
                         "+ its backward (the trainLight step's kernels: fwd_pk*gt* and sg_bwd_recon_pk_kernel; _heads: decoder outputs in, premap 3) "
                         "for --warmup + --steps iterations and print a short record instead of the contract line")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="BASELINE.json configs index: 2 = headline (default); 5 = 480x640, SGNum 24, 16x32 stress (batch 4)")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(gpus: int, argv) -> list:
    """`python bench.py --gpus N ...` outside torchrun -> the driver's own launch form for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(args) -> None:
    """Round 5 raised SystemExit here ("needs a torchrun launch"): the same command the driver uses at N = 1, with --gpus 8, died before
    touching a GPU.  Now the process BECOMES the torchrun launcher (exec: same stdout, same exit code), after checking that the box has the
    GPUs -- RCCL needs one per rank; SGR_BENCH_BACKEND=gloo is the flagged rehearsal on fewer."""
    backend = os.environ.get("SGR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the render layer has no CPU path")
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {ndev} GPU(s) visible and RCCL needs one per rank "
                         "(SGR_BENCH_BACKEND=gloo rehearses the N > 1 flow on fewer GPUs; such a line is flagged config.rehearsal)")
    cmd = self_launch_command(args.gpus, sys.argv[1:])
    print("# bench.py: --gpus %d outside torchrun -> %s" % (args.gpus, " ".join(cmd[1:8])), file=sys.stderr)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


class Workload:
    """One rank's batch of synthetic inputs (generated on the CPU, seed offset per rank) and the steps timed on it."""

    def __init__(self, pkg, O, dev, bn, dims, seed, need_env=True, group=None):
        imH, imW, R, C, K, eh, ew = dims
        self.pkg, self.dev, self.bn, self.R, self.C, self.need_env, self.group = pkg, dev, bn, R, C, need_env, group
        inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=seed)
        self.x = {k: v.to(dev) for k, v in inp.items()}
        for k in ("axis", "lamb", "weight"):
            self.x[k].requires_grad_(True)
        g = torch.Generator().manual_seed(99 + seed % 1000)
        self.ct_env = (torch.randn((bn, 3, R, C, eh, ew), generator=g) * 1e-3).to(dev) if need_env else None
        self.ct_d = torch.randn((bn, 3, R, C), generator=g).to(dev)
        self.ct_s = torch.randn((bn, 3, R, C), generator=g).to(dev)
        self.layer = pkg.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
        self.ind = torch.ones(bn, 1, 1, 1, device=dev)

    def sg(self, x=None):
        x = x or self.x
        return [x["axis"], x["lamb"], x["weight"]]

    def step(self, ev=None, x=None, ct_env=None):
        """the N = 1 headline: fused forward + fused backward of the layer"""
        x = x or self.x
        ct_env = self.ct_env if ct_env is None else ct_env
        if ev is not None:
            ev[0].record()
        env, d, s = self.layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=self.need_env)
        if ev is not None:
            ev[1].record()
        outs, cts = ([env, d, s], [ct_env, self.ct_d, self.ct_s]) if self.need_env else ([d, s], [self.ct_d, self.ct_s])
        grads = torch.autograd.grad(outs, self.sg(x), grad_outputs=cts)
        if ev is not None:
            ev[2].record()
        return grads

    def step_with_loss(self, group="default"):
        """the same step with the render loss in the loop (LSregressDiffSpec + masked L2 kernels, wrapperBRDFLight.py:170-207 around the
        layer); sharded: the all-reduce of [num, den] sits between forward and backward -- the N > 1 headline"""
        x, group = self.x, (self.group if group == "default" else group)
        env, d, s = self.layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=self.need_env)
        err, _ = self.pkg.render_loss(d, s, x["im"], x["seg"], self.R, self.C, group=group)
        if self.need_env:
            return torch.autograd.grad([err, env], self.sg(), grad_outputs=[None, self.ct_env])
        return torch.autograd.grad([err], self.sg())

    def step_objective(self, group="default", x=None):
        """the cascade-0 light objective (render loss + 10 x env reconstruction loss, wrapperBRDFLight.py:167-207), fused, fwd + bwd"""
        x, group = x or self.x, (self.group if group == "default" else group)
        obj = self.pkg.light_objective(self.layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"],
                                       x["im"], x["seg"], x["env_gt"], self.ind, 1.0, 10.0, group=group)[0]
        return torch.autograd.grad(obj, self.sg(x))

    def release(self):
        self.x = self.ct_env = self.ct_d = self.ct_s = None


def clip(v, n=STR_LIMIT):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                      # does not return
    emit = _claim_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without torchrun: bench.py launches itself)")
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
    n_ranks_seen = 1
    if world > 1:       # input generation is CPU work: do not oversubscribe the host with world x all-core thread pools
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that dies leaves the others in a collective: ten minutes, then they fail too instead of holding the box
        kw = dict(timeout=datetime.timedelta(minutes=10))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                  # the first collective: every rank is there, over the backend the line names
        n_ranks_seen = int(ones.item())

    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    from oracle import sg_oracle as O      # bench-only: synthetic inputs + the cpu_baseline leg

    _lib.load()
    bn, imH, imW, R, C, K, eh, ew = args.batch, 240, 320, 120, 160, 12, 8, 16
    if args.config == 5:      # BASELINE configs[4]: env grid 240x320 assumed
        bn, imH, imW, R, C, K, eh, ew = (4 if args.batch == 16 else args.batch), 480, 640, 240, 320, 24, 16, 32
    if args.strong:
        if bn % world:
            raise SystemExit(f"--strong: {bn} images do not split evenly over {world} ranks")
        bn //= world
    dims = (imH, imW, R, C, K, eh, ew)
    J, q = eh * ew, (imH // R) * (imW // C)
    need_env = not args.no_env
    group = dist.group.WORLD if world > 1 else None

    # different images on every rank (seed offset), generated on the CPU like SURVEY 8d prescribes
    wl = Workload(pkg, O, dev, bn, dims, 20202 + 1000 * rank, need_env, group)
    x, layer = wl.x, wl.layer
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    if args.pmc_workload != "layer":      # counter-collection workload: the objective's kernels alone
        heads = args.pmc_workload == "objective_heads"
        if heads:
            gh = torch.Generator().manual_seed(7)
            sg = [(torch.randn(s, generator=gh) * 0.5).to(dev).requires_grad_(True) for s in ((bn, 3 * K, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
        else:
            sg = wl.sg()
        for _ in range(args.warmup + args.steps):
            obj = pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], sg[0], sg[1], sg[2], x["im"], x["seg"], x["env_gt"], wl.ind, 1.0, 10.0,
                                      decoder_outputs=heads)[0]
            torch.autograd.grad(obj, sg)
        torch.cuda.synchronize()
        emit(json.dumps({"pmc_workload": args.pmc_workload, "config": args.config, "batch": bn, "iterations": args.warmup + args.steps}))
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, events=None):
        """EXACTLY `steps` steps between two barrier + synchronise brackets; seconds, max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(events[i]) if events is not None else fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    for _ in range(args.warmup):
        wl.step()
        wl.step_with_loss()

    reps = max(1, args.reps)
    fwd_acc = bwd_acc = 0.0
    plain_dts, loss_dts = [], []
    for _ in range(reps):
        plain_dts.append(timed(wl.step, args.steps, ev))
        fwd_acc += sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
        bwd_acc += sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    for _ in range(reps):
        loss_dts.append(timed(wl.step_with_loss, args.steps))
    fwd_ms, bwd_ms = fwd_acc / reps, bwd_acc / reps
    plain_ms, loss_step_ms = median(plain_dts) / args.steps * 1e3, median(loss_dts) / args.steps * 1e3
    # the contract line: single GPU = the layer's forward + backward; sharded = the same plus the loss and its all-reduce
    headline_dts = plain_dts if world == 1 else loss_dts
    dt = median(headline_dts)

    # The informational legs keep their own loop lengths: at least 60 timed iterations after 10 untimed ones, whatever --steps / --warmup say.
    # With the driver's `--steps 20 --warmup 5` a 20-iteration loop of a 0.5 ms step is 10 ms -- shorter than the ~100 ms the clocks take
    # to settle after the previous leg -- and read 3-8 % high (config 3: 0.78 vs 0.71 ms on one box).  The headline honours the flags exactly.
    leg_steps, leg_warmup = max(args.steps, 60), 10

    def loop_ms(fn, n=leg_steps, warm=leg_warmup):
        for _ in range(warm):
            fn()
        return timed(fn, n) / n * 1e3

    flat, detail = {}, {}      # flat: scalars for `config` (keys <= 40 chars); detail: everything nested -> bench_detail.json
    legs = need_env and not args.layer_only and args.config == 2

    # ---- N > 1 legs: every rank takes part, in the same order (a leg that one rank skipped would leave the others in its barrier) ----------
    if world > 1 and legs:
        flat["obj_ms"] = round(loop_ms(wl.step_objective), 4)                   # the sharded light objective: 2 all-reduces per step
        if not args.strong:
            if bn != 8:      # BASELINE configs[3]: 8 images per GPU (batch 64 at N = 8), render loss + all-reduce in the step
                w4 = Workload(pkg, O, dev, 8, dims, 24202 + 1000 * rank, True, group)
                ms = loop_ms(w4.step_with_loss)
                flat["cfg4_ms"], flat["cfg4_Mpix_per_s"], flat["cfg4_global_batch"] = round(ms, 4), round(world * 8 * imH * imW / (ms * 1e-3) / 1e6, 1), 8 * world
                w4.release()
            if 16 % world == 0:      # SURVEY 8d strong scaling: the N = 1 batch of 16 split N ways
                ws = Workload(pkg, O, dev, 16 // world, dims, 25202 + 1000 * rank, True, group)
                ms = loop_ms(ws.step_with_loss)
                flat["strong16_ms"], flat["strong16_Mpix_per_s"] = round(ms, 4), round(16 * imH * imW / (ms * 1e-3) / 1e6, 1)
                ws.release()

        # the extension's in-stream all-reduce over N ranks (sgr.enable_native_allreduce: its own RCCL communicator, ncclAllReduce on the
        # current HIP stream).  OPT-IN (SGR_BENCH_NATIVE_AR=1): the builder's boxes have one GPU, so this route has run with a world of one
        # only (rccl1_*_native_ms in the N = 1 line, tests/test_gpu_rccl_world1.py) and the driver's scaling run must not depend on it
        if os.environ.get("SGR_BENCH_NATIVE_AR") == "1" and backend == "nccl":
            pkg.enable_native_allreduce(group)
            flat["native_ms_with_loss"] = round(loop_ms(wl.step_with_loss), 4)
            flat["native_obj_ms"] = round(loop_ms(wl.step_objective), 4)
            pkg.disable_native_allreduce(group)

    # ---- single-process informational legs -------------------------------------------------------------------------------------------------
    cfg3 = cfg5 = rccl1 = None
    if world == 1 and legs:
        def clear():
            for k in ("axis", "lamb", "weight"):
                x[k].grad = None

        def step_obj_forward_only():      # evaluation loops (testLight.py drives the same wrapper without a backward): no gradient kernel is launched
            with torch.no_grad():
                pkg.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"], wl.ind, 1.0, 10.0)

        def step_obj_unfused():
            env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
            err, _ = pkg.render_loss(d, s, x["im"], x["seg"], R, C)
            rec = pkg.recon_loss(env, x["env_gt"], x["seg"], wl.ind, R, C)
            (err + 10.0 * rec).backward()
            clear()

        try:
            flat["obj_ms"] = round(loop_ms(wl.step_objective), 4)
            flat["obj_unfused_ms"] = round(loop_ms(step_obj_unfused), 4)
            flat["obj_fwd_only_ms"] = round(loop_ms(step_obj_forward_only), 4)
        except Exception as exc:       # informational legs: never fail the bench over them
            print(f"# light-objective legs skipped: {str(exc)[:160]}", file=sys.stderr)

        # BASELINE config 3: the synthetic trainLight cascade-0 step (trainLight.py:203-244 around wrapperBRDFLight.py:158-207):
        # learnable decoder outputs -> output activations (sgr.light_heads) -> light objective -> backward -> Adam
        try:
            cfg3 = config3_legs(pkg, layer, x, wl.ind, bn, R, C, K, leg_steps, barrier, leg_warmup)
        except Exception as exc:
            cfg3 = {"error": str(exc)[:200]}
        detail["config3"] = cfg3
        flat["cfg3_ms"] = cfg3.get("ms_per_step_config3")
        flat["cfg3_ms_graph"] = cfg3.get("ms_per_step_config3_hipgraph")
        flat["cfg3_ms_standalone_heads"] = cfg3.get("ms_per_step_config3_standalone_heads")
        for tag, kk in (("obj", "kernels_config3_standalone_heads"), ("cfg3", "kernels_config3")):      # obj_*: the plain objective's kernels; cfg3_*: heads as prologue
            ks = cfg3.get(kk) or {}
            for side, name in (("fwd", "objective_forward"), ("bwd", "objective_backward")):
                r = (ks.get(name) or {}).get("roofline")
                if r:
                    flat[f"{tag}_{side}_frac"], flat[f"{tag}_{side}_us"] = r["frac"], round(r["avg_launch_ms"] * 1e3, 1)
                    if r.get("traffic"):
                        flat[f"{tag}_{side}_traffic_ratio"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)

        # BASELINE configs[3] at one rank: 8 images, render loss in the step -- the per-GPU work of the 8-GPU batch of 64
        try:
            if bn != 8:
                w4 = Workload(pkg, O, dev, 8, dims, 24202, True, None)
                ms = loop_ms(w4.step_with_loss)
                flat["cfg4_ms"], flat["cfg4_Mpix_per_s"], flat["cfg4_global_batch"] = round(ms, 4), round(8 * imH * imW / (ms * 1e-3) / 1e6, 1), 8
                w4.release()
                del w4
        except Exception as exc:
            print(f"# config-4 leg skipped: {str(exc)[:160]}", file=sys.stderr)

        # the COLD column: the same two steps rotating through four input sets (4 x 0.6 GB of inputs against 256 MB of Infinity Cache), so that
        # no step finds its inputs in a cache -- what a data loader handing over a fresh batch every step sees (profiles/r05l_*: forwardEnv
        # alone loses 45 % cold, the objective's forward 11 %)
        try:
            flat.update(cold_legs(wl, leg_steps, loop_ms))
        except Exception as exc:
            print(f"# cold legs skipped: {str(exc)[:160]}", file=sys.stderr)
            torch.cuda.empty_cache()

    # informational: the with-render-loss step replayed from a HIP graph (sgr.capture_step): what a launch-bound caller -- the reference's
    # default batch of 5, trainLight.py:28 -- gets by capturing its step
    if world == 1 and (args.graph_leg or legs):
        try:
            captured = pkg.capture_step(wl.step_with_loss)
            captured.replay()
            flat["ms_with_loss_graph"] = round(timed(captured.replay, leg_steps) / leg_steps * 1e3, 4)
            del captured
        except Exception as exc:
            print(f"# graph-replay leg skipped: {str(exc)[:160]}", file=sys.stderr)

    # informational, LAST of the config-2 legs (it creates and destroys a process group): the with-loss step and the fused objective through
    # RCCL with a world of one rank -- the c10d route (dist.all_reduce between the stage operators) and the extension's in-stream route
    # (ncclAllReduce enqueued on the current HIP stream by sgrender::allreduce_sum_, no c10d call in the step) -- so the collectives' cost
    # on the step is a measured number.  N > 1 is the driver's to run.
    if world == 1 and legs:
        try:
            rccl1 = rccl_world1_legs(pkg, wl, leg_steps, dev)
        except Exception as exc:
            rccl1 = {"error": str(exc)[:200]}
        detail["rccl_world1"] = rccl1
        for k_out, k_in in (("rccl1_loss_base_ms", "ms_per_step_with_render_loss"), ("rccl1_loss_ms", "ms_per_step_with_render_loss_sharded_world1"),
                            ("rccl1_loss_native_ms", "ms_per_step_with_render_loss_native_world1"),
                            ("rccl1_obj_base_ms", "ms_per_step_light_objective"), ("rccl1_obj_ms", "ms_per_step_light_objective_sharded_world1"),
                            ("rccl1_obj_native_ms", "ms_per_step_light_objective_native_world1")):
            flat[k_out] = rccl1.get(k_in)

    if world == 1 and legs and not args.no_config5:
        try:
            wl.release()      # config 2's inputs are done with: 1.2 GB back to the allocator before config 5's 6 GB
            x = None
            torch.cuda.empty_cache()
            cfg5 = config5_leg(pkg, dev)
        except Exception as exc:
            cfg5 = {"error": str(exc)[:200]}
        detail["config5"] = cfg5
        flat["cfg5_Mpix_per_s"], flat["cfg5_ms"] = cfg5.get("value"), cfg5.get("ms_per_step")
        for side, name in (("fwd", "forward (sgr_fused_fwd)"), ("bwd", "backward (sgr_fused_bwd_sg)")):
            r = ((cfg5.get("kernels") or {}).get(name) or {}).get("roofline")
            if r:
                flat[f"cfg5_{side}_frac"] = r["frac"]

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
        # against the kernel's duration in shader-clock cycles (per-shader-engine SQ_BUSY_CYCLES, tools/parse_sq.py) is the fraction of
        # issue cycles used.  Both records are OFFLINE (counter passes cannot run inside the timed loop) and stamped with the hash of
        # the kernel sources they were measured on: a record from other sources is reported as stale and does not decide `limited_by`
        valu = valu_roofline(tkey, want, dom[1])
        limited_by = limited(valu, dom[3] / HBM_PEAK_GBPS)
        mpix = lambda ms: round(world * img_px / (ms * 1e-3) / 1e6, 1)
        cfg_idx = 1 if args.config == 2 else 4
        config = {
            "workload": clip(f"BASELINE configs[{cfg_idx}]/GPU: {bn}x{imH}x{imW} maps -> {R}x{C} env grid, K={K}, {eh}x{ew} dirs; fused fwd "
                             f"({'env written' if need_env else 'render only'}) + bwd (SG grads)"),
            "timed_step": "layer fwd + bwd" if world == 1 else "fwd + render loss (all-reduce [num,den] over RCCL) + bwd",
            "parallelism": f"batch-sharded x{world}", "n_ranks_seen": n_ranks_seen, "backend": backend if world > 1 else None,
            "batch_per_gpu": bn, "global_batch": bn * world, "strong": bool(args.strong),
            "shaded_px_per_step_per_gpu": P, "image_px_per_step_per_gpu": img_px, "q": q,
            "Mshade_per_s": round(world * P / (dt / args.steps) / 1e6, 1),
            "repetitions": reps, "statistic": "median repetition of the K-step loop, max over ranks",
            "ms_layer_only": round(plain_ms, 4), "Mpix_layer_only": mpix(plain_ms),
            "ms_with_loss": round(loss_step_ms, 4), "Mpix_with_loss": mpix(loss_step_ms),
            "fwd_us": round(fwd_ms * 1e3, 1), "bwd_us": round(bwd_ms * 1e3, 1),
            "fwd_frac": round(fwd_gbps / HBM_PEAK_GBPS, 4), "bwd_frac": round(bwd_gbps / HBM_PEAK_GBPS, 4),
            "legs_timed_iterations": leg_steps, "legs_untimed_iterations": leg_warmup,
        }
        # the like-for-like anchor of an N-GPU `value` (whose step carries the render loss): the N = 1 with-loss figure
        anchor_key = f"cfg{args.config}_b{bn}_{'env' if need_env else 'noenv'}"
        if world == 1:
            config["scaling_anchor_Mpix_per_s"] = mpix(loss_step_ms)
            if not args.strong:
                _write_anchor(anchor_key, dict(Mpix_with_loss=mpix(loss_step_ms), Mpix_layer_only=mpix(plain_ms), cfg4_Mpix_per_s=flat.get("cfg4_Mpix_per_s"),
                                               obj_ms=flat.get("obj_ms")))
        else:
            a = _read_anchor(f"cfg{args.config}_b{16 if args.strong else bn}_{'env' if need_env else 'noenv'}")
            config["scaling_anchor_Mpix_per_s"] = a.get("Mpix_with_loss")
            config["anchor_Mpix_layer_only"] = a.get("Mpix_layer_only")
            config["anchor_cfg4_Mpix_per_s"] = a.get("cfg4_Mpix_per_s")
            config["anchor_source"] = a.get("source")
        config.update({k: v for k, v in flat.items()})
        config["rehearsal"] = None if backend == "nccl" else clip(f"backend {backend}, {ndev} GPU(s) for {world} ranks: the N > 1 flow only, not a measurement")
        config["detail_file"] = os.path.basename(DETAIL_FILE)
        out = {
            "metric": "Mpix/s shaded (fwd+bwd), 240x320x12-SG render layer" if args.config == 2 else "Mpix/s shaded (fwd+bwd), 480x640x24-SG 16x32 render layer (stress config)",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": config,
            "roofline": {"bound": "hbm", "kernel": clip(dom_name, 80), "achieved": round(dom[3], 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(dom[3] / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_record_stale": traffic_stale,
                         "algorithmic_bytes_per_launch": dom[2], "avg_launch_ms": round(dom[1], 4),
                         "limited_by": limited_by},
        }
        if valu is not None:      # compact; the full record (per-wave cycle shares, instruction counts, the source note) -> bench_detail.json
            out["roofline_valu"] = {k: valu.get(k) for k in ("bound", "kernel", "frac", "frac_at_max_clock", "effective_clock_GHz", "pmc_kernel_ms",
                                                             "live_kernel_ms", "record", "record_stale")}
            out["roofline_valu"]["kernel"] = clip(out["roofline_valu"]["kernel"], 80)
        detail["roofline_valu"] = valu
        detail["ms_per_step_repetitions"] = [round(t / args.steps * 1e3, 4) for t in headline_dts]
        detail["ms_layer_only_repetitions"] = [round(t / args.steps * 1e3, 4) for t in plain_dts]
        detail["ms_with_loss_repetitions"] = [round(t / args.steps * 1e3, 4) for t in loss_dts]
        detail["kernels"] = {"forward (sgr_fused_fwd)": {"ms": round(fwd_ms, 4), "GBps": round(fwd_gbps, 1), "frac": round(fwd_gbps / HBM_PEAK_GBPS, 4), "bytes": fwd_bytes},
                             "backward (sgr_fused_bwd_sg)": {"ms": round(bwd_ms, 4), "GBps": round(bwd_gbps, 1), "frac": round(bwd_gbps / HBM_PEAK_GBPS, 4), "bytes": bwd_bytes}}
        if world == 1 and not args.no_cpu_baseline and not args.layer_only and args.config == 2:
            cb = cpu_baseline(O, 240, 320, 120, 160, 12, 8, 16)
            # port vs the UNMODIFIED reference on the same cores, measured where the reference exists (the authoring
            # container; oracle/calibrate_port_vs_reference.py -> profiles/cpu_calibration.json)
            try:
                cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
            except Exception:
                cal = None
            detail["cpu_baseline"] = dict(cb, calibration=cal)
            ratio = ((cal or {}).get("modes", {}).get("fwd_bwd_sg", {}) or {}).get("broadcast_port_over_reference_speed")
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": clip(cb["sample_short"]), "cpu": clip(cb["cpu"], 60),
                                   "fwd_only_Mpix_per_s": cb["modes"]["forward_only"]["Mpix_per_s"],
                                   "all_grads_Mpix_per_s": cb["modes"]["fwd_bwd_all_grads"]["Mpix_per_s"]}
            if cb["kind"] == "port" and ratio:      # an ESTIMATE: the ratio was measured on the authoring container's CPU, not this one
                out["cpu_baseline"]["port_over_reference_speed"] = ratio
                out["cpu_baseline"]["reference_estimate_Mpix_per_s"] = round(cb["value"] / ratio, 4)
            try:
                eg = eager_gpu_baseline(O, dev, 240, 320, 120, 160, 12, 8, 16)
                detail["eager_gpu_baseline"] = eg
                out["eager_gpu_baseline"] = {"value": eg["value"], "unit": eg["unit"], "kind": eg["kind"]}
            except Exception as exc:       # informational leg: never fail the bench over it
                out["eager_gpu_baseline"] = {"error": str(exc)[:120]}
        detail["line"] = out
        try:
            with open(DETAIL_FILE, "w") as fh:
                json.dump(detail, fh, indent=1)
        except OSError as exc:
            print(f"# bench_detail.json not written: {exc}", file=sys.stderr)
        emit(fit_line(out))

    if world > 1:
        dist.destroy_process_group()


def fit_line(out: dict) -> str:
    """The contract line within LINE_LIMIT bytes: None-valued optional `config` keys go first, then the optional top-level objects."""
    line = json.dumps(out)
    if len(line) <= LINE_LIMIT:
        return line
    cfg = out["config"]
    for k in [k for k, v in cfg.items() if v is None]:
        del cfg[k]
    for k in ("eager_gpu_baseline", "roofline_valu"):
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        out.pop(k, None)
    return json.dumps(out)


def _host_id() -> str:
    """hostname + boot id: GPU boxes of one pool share a hostname, a boot id they do not"""
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        boot = ""
    return socket.gethostname() + "/" + boot


def _write_anchor(key, rec):
    if os.environ.get("SGR_BENCH_NO_ANCHOR"):      # test runs (three-step loops) must not leave figures a later N > 1 run would quote
        return
    try:
        try:
            allrec = json.load(open(ANCHOR_FILE))
        except Exception:
            allrec = {}
        allrec[key] = dict(rec, host=_host_id(), time=time.time())
        with open(ANCHOR_FILE, "w") as fh:
            json.dump(allrec, fh)
    except OSError:
        pass


def _read_anchor(key) -> dict:
    """The N = 1 figures an N > 1 line is read against: left by an N = 1 run of this bench on the same host within the last two hours (the
    driver runs N = 1, 2, 4, 8 back to back); otherwise nothing -- never a figure from another box."""
    try:
        rec = json.load(open(ANCHOR_FILE)).get(key)
        if rec and rec.get("host") == _host_id() and time.time() - rec.get("time", 0) < 7200:
            return dict(rec, source="N=1 run of this bench on this host, %d s earlier" % int(time.time() - rec["time"]))
    except Exception:
        pass
    return {}


def cold_legs(wl, steps, loop_ms, nsets=4) -> dict:
    """`ms_per_step_cold` / `obj_ms_cold`: the layer step and the fused-objective step with the inputs rotating through `nsets` copies at different
    addresses (0.6 GB per set for the layer step incl. the env cotangent, 0.6 GB for the objective incl. the ground-truth env), against the warm
    figure taken back to back with the same loop length."""
    names = ("albedo", "normal", "rough", "axis", "lamb", "weight", "im", "seg", "env_gt")
    sets = []
    for _ in range(nsets):
        xs = {k: wl.x[k].detach().clone() for k in names}
        for k in ("axis", "lamb", "weight"):
            xs[k].requires_grad_(True)
        sets.append((xs, wl.ct_env.clone()))
    state = {"i": 0}

    def nxt():
        state["i"] = (state["i"] + 1) % nsets
        return sets[state["i"]]

    def layer_cold():
        xs, ct = nxt()
        wl.step(None, xs, ct)

    def obj_cold():
        xs, _ = nxt()
        wl.step_objective(None, xs)

    out = {}
    warm = loop_ms(wl.step, steps)
    out["ms_per_step_cold"], out["ms_per_step_warm_same_loop"] = round(loop_ms(layer_cold, steps), 4), round(warm, 4)
    warm_o = loop_ms(lambda: wl.step_objective(None), steps)
    out["obj_ms_cold"], out["obj_ms_warm_same_loop"] = round(loop_ms(obj_cold, steps), 4), round(warm_o, 4)
    out["cold_input_sets"] = nsets
    del sets
    torch.cuda.empty_cache()
    return out


def rccl_world1_legs(pkg, wl, steps, dev) -> dict:
    """The sharded code path over RCCL with ONE rank (single-process bench only): `render_loss(group=WORLD)` between the layer's forward and
    backward, and `light_objective(group=WORLD)` (three stage operators, two all-reduces) -- through c10d (`dist.all_reduce`) and through the
    extension's own communicator (`sgr.enable_native_allreduce`: ncclAllReduce enqueued on the current HIP stream by
    `torch.ops.sgrender.allreduce_sum_`) -- against the same steps without a group, timed back to back.  What it measures is the collectives'
    enqueue + latency on the critical path; the wire (xGMI) is not involved."""
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
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
        # alternate the routes so that all see the same clocks
        a = [loop_ms(wl.step_with_loss, None), loop_ms(wl.step_with_loss, W), loop_ms(wl.step_with_loss, None), loop_ms(wl.step_with_loss, W)]
        b = [loop_ms(wl.step_objective, None), loop_ms(wl.step_objective, W), loop_ms(wl.step_objective, None), loop_ms(wl.step_objective, W)]
        out["ms_per_step_with_render_loss"] = round(min(a[0], a[2]), 4)
        out["ms_per_step_with_render_loss_sharded_world1"] = round(min(a[1], a[3]), 4)
        out["ms_per_step_light_objective"] = round(min(b[0], b[2]), 4)
        out["ms_per_step_light_objective_sharded_world1"] = round(min(b[1], b[3]), 4)
        try:      # the in-stream route: same group, the collectives now enqueued by the extension itself
            pkg.enable_native_allreduce(W)
            c = [loop_ms(wl.step_with_loss, W), loop_ms(wl.step_with_loss, None), loop_ms(wl.step_with_loss, W)]
            d = [loop_ms(wl.step_objective, W), loop_ms(wl.step_objective, None), loop_ms(wl.step_objective, W)]
            out["ms_per_step_with_render_loss_native_world1"] = round(min(c[0], c[2]), 4)
            out["ms_per_step_light_objective_native_world1"] = round(min(d[0], d[2]), 4)
            out["ms_per_step_with_render_loss_2"], out["ms_per_step_light_objective_2"] = round(c[1], 4), round(d[1], 4)
        except Exception as exc:
            out["native_error"] = str(exc)[:200]
        finally:
            pkg.disable_native_allreduce(W)
        out["note"] = ("one rank over the nccl (= RCCL) backend: one all-reduce of [num, den] per render loss, two per light objective, on device tensors; "
                       "'sharded' = through c10d, 'native' = ncclAllReduce on the current stream from the extension")
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
            "sample_short": f"1 image (1/16 step) of the workload, fp32, fwd+bwd(SG grads), best of {runs}: {best:.3f} s; kind={kind} (see detail file)",
            "modes": {m: {"Mpix_per_s": mp(res[m][0]), "seconds_per_image": round(res[m][0], 3), "runs": res[m][1]} for m in MODES},
            "sample": f"1 image (1/16 of a step) of the same workload, {what}, best of <= 4 per mode; value = fwd+bwd (SG grads), best of {runs}; {best:.3f} s per image",
            "cpu": model}


if __name__ == "__main__":
    main()
