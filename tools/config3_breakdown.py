"""Per-step kernel-time breakdown of the config-3 training step from a rocprofv3 kernel_stats.csv (development tool):
   python tools/config3_breakdown.py <kernel_stats.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
groups = {"objective forward (fwd_pk_kernel + folds)": ("fwd_pk_kernel", "fwd_pk_half_gt", "recon_fold0"), "objective backward (sg_bwd_recon_pk + fold)": ("sg_bwd_recon", "recon_fold1"),
          "decoder heads (heads_fwd / heads_bwd)": ("heads_",), "render loss (stages, finalize, bwd)": ("loss_",), "Adam (fused multi-tensor)": ("FusedAdam", "multi_tensor"),
          "rescale / set_scalar": ("rescale_kernel", "set_scalar")}
tot = sum(float(r["TotalDurationNs"]) for r in rows)
acc = {k: 0.0 for k in groups}
other = []
for r in rows:
    for g, pats in groups.items():
        if any(p in r["Name"] for p in pats):
            acc[g] += float(r["TotalDurationNs"]); break
    else:
        other.append((float(r["TotalDurationNs"]), r["Name"][:90], r["Calls"]))
print(f"config-3 step: kernel time per step {tot / steps / 1e3:.1f} us over {steps} steps (the first 3 steps include warm-up launches)")
for g, v in acc.items():
    print(f"  {g:52s} {v / steps / 1e3:8.1f} us/step  {100 * v / tot:5.1f} %")
print(f"  {'other torch kernels (pooling, fills, elementwise glue)':52s} {sum(o[0] for o in other) / steps / 1e3:8.1f} us/step  {100 * sum(o[0] for o in other) / tot:5.1f} %")
for d, n, c in sorted(other, reverse=True)[:6]:
    print(f"      {d / steps / 1e3:7.1f} us/step  x{c:>5s}  {n}")
