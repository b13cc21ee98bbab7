"""Summarise rocprofv3 --pmc SQ passes (tools/pmc_sq.sh) into per-kernel VALU-issue figures per launch.

  valu_busy_cycles_per_simd = SQ_ACTIVE_INST_VALU x 4 / SIMDs      (the counter ticks per SIMD quad while a VALU instruction is in flight)
  kernel_cycles             = SQ_BUSY_CYCLES / shader engines      (round 5: the clock cycles a shader engine has waves resident, averaged over
                                                                    the 32 engines -- rocprofiler-sdk counter_defs.yaml: "per-shader engine basis
                                                                    in clock cycles"), capped at kernel_ms x 2.4 GHz
  valu_insts_per_wave       = SQ_INSTS_VALU / SQ_WAVES

Rounds 3-4 divided by GRBM_GUI_ACTIVE / XCDs.  Under rocprofv3's counter mode that register also counts the serialised launch's pre- and
post-amble: for a 0.14 ms kernel it read 0.38 M "cycles" -- a 2.65 GHz clock on a part whose maximum is 2.4, 5-10 GHz for the 7 us loss
kernels -- so every busy fraction was under-stated, the short kernels' most (the round-4 review's finding).  The per-engine SQ_BUSY_CYCLES implies
1.72-2.05 GHz for every kernel on record, in line with the DVFS behaviour MI355X_MICROARCH.md describes and with the s_memtime / s_memrealtime
ratio of tools/wavetrace.  Because a shader engine that has drained early stops counting, SQ_BUSY_CYCLES can only UNDER-state the kernel's cycles
(-> `frac` is an upper bound of the busy fraction); `frac_at_max_clock` = busy cycles / (kernel_ms x 2.4 GHz) is the lower bound (it charges the
clock the part did not run at to the kernel), and `issue_rate_per_us` = busy cycles per SIMD per microsecond is the clock-free figure to
compare kernels by.

Writes {kernel: {...}} under the workload tag into the target json (profiles/sq.json: what bench.py's `roofline_valu` reads).
  tools/parse_sq.py <pmc dir> <tag> <target.json>         after a tools/pmc_sq.sh run
  tools/parse_sq.py --reprocess <target.json>             recompute every record of the file from its stored raw counters (no GPU)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SIMDS, XCDS, SES = 1024, 8, 32
MAX_CLOCK_GHZ = 2.4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMENT = ("VALU-issue figures per launch from SQ counters (tools/pmc_sq.sh + tools/parse_sq.py), keyed by workload then kernel; "
           "kernel_cycles = SQ_BUSY_CYCLES / 32 shader engines (capped at kernel_ms x 2.4 GHz), see tools/parse_sq.py; "
           "bench.py's roofline_valu reports a figure only for the workload it was measured on")


def csrc_sha16(root):
    """Hash of the kernel sources (csrc/*.hip, *.inl, *.h): counter records are stamped with it, and bench.py reports a record as
    stale when the sources it was measured on are not the ones the loaded library was built from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "inverserenderingofindoorscene_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inl")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def record(m, kernel_ms):
    """One kernel's figures from its averaged raw counters `m` and its traced duration."""
    if "SQ_ACTIVE_INST_VALU" not in m:
        return None
    busy = m["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS
    cap = kernel_ms * MAX_CLOCK_GHZ * 1e6 if kernel_ms else None
    if "SQ_BUSY_CYCLES" in m:
        cycles, source = m["SQ_BUSY_CYCLES"] / SES, "SQ_BUSY_CYCLES / 32 shader engines"
    elif "GRBM_GUI_ACTIVE" in m:
        cycles, source = m["GRBM_GUI_ACTIVE"] / XCDS, "GRBM_GUI_ACTIVE / 8 XCDs (no SQ_BUSY_CYCLES in this record)"
    else:
        return None
    capped = bool(cap and cycles > cap)
    if capped:
        cycles, source = cap, source + ", capped at kernel_ms x 2.4 GHz"
    rec = {"valu_busy_cycles_per_simd": busy, "kernel_cycles": cycles, "kernel_cycles_source": source,
           "valu_insts_per_wave": round(m["SQ_INSTS_VALU"] / max(m.get("SQ_WAVES", 1.0), 1.0), 1) if "SQ_INSTS_VALU" in m else None,
           "waves": m.get("SQ_WAVES"), "kernel_ms": kernel_ms,
           "trans_share": round(m["SQ_INSTS_VALU_TRANS_F32"] / m["SQ_INSTS_VALU"], 4) if m.get("SQ_INSTS_VALU_TRANS_F32") and m.get("SQ_INSTS_VALU") else None,
           "counters": {c: round(x, 1) for c, x in m.items()}}
    rec["frac"] = round(busy / cycles, 4)
    rec["effective_clock_GHz"] = round(cycles / (kernel_ms * 1e6), 3) if kernel_ms else None
    rec["frac_at_max_clock"] = round(busy / cap, 4) if cap else None
    rec["issue_rate_per_us"] = round(busy / (kernel_ms * 1e3), 1) if kernel_ms else None
    # where a wave's resident cycles go, when the stall pass (tools/pmc_sq.sh pass c) was collected: all in quad-cycles summed over waves
    if m.get("SQ_WAVE_CYCLES"):
        wc = m["SQ_WAVE_CYCLES"]
        rec["per_wave_cycle_shares"] = {k: round(m[c] / wc, 4) for k, c in (
            ("issuing_any", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("issuing_scalar", "SQ_ACTIVE_INST_SCA"),
            ("issuing_lds", "SQ_ACTIVE_INST_LDS"), ("issuing_vmem", "SQ_ACTIVE_INST_VMEM"), ("waiting_any", "SQ_WAIT_ANY"),
            ("waiting_on_inst", "SQ_WAIT_INST_ANY"), ("waiting_on_lds", "SQ_WAIT_INST_LDS")) if c in m}
        rec["mean_resident_waves_per_simd"] = round(wc * 4.0 / SIMDS / (m["SQ_BUSY_CYCLES"] / SES), 3) if m.get("SQ_BUSY_CYCLES") else None
    return rec


def show(k, rec):
    print(f"{k:72s} VALU-busy {rec['frac']:.3f} (>= {rec['frac_at_max_clock']} at 2.4 GHz)  ({rec['valu_busy_cycles_per_simd']:.0f} of {rec['kernel_cycles']:.0f} cycles)  "
          f"{rec['valu_insts_per_wave']} VALU instr/wave  trans share {rec['trans_share']}  {rec['kernel_ms']} ms  clock {rec['effective_clock_GHz']} GHz  "
          f"issue rate {rec['issue_rate_per_us']} busy cycles/us")
    if rec.get("per_wave_cycle_shares"):
        print(f"{'':72s} per-wave cycle shares {rec['per_wave_cycle_shares']}  resident waves/SIMD {rec.get('mean_resident_waves_per_simd')}")


def reprocess(target):
    allrec = json.load(open(target))
    for tag, recs in allrec.items():
        if tag.startswith("_") or not isinstance(recs, dict):
            continue
        for k, r in list(recs.items()):
            if k.startswith("_") or "counters" not in r:
                continue
            new = record(r["counters"], r.get("kernel_ms"))
            if new is not None:
                recs[k] = new
                show(tag + " " + k.split("::")[-1][:44], new)
    allrec["_comment"] = COMMENT
    json.dump(allrec, open(target, "w"), indent=1)


def main():
    if sys.argv[1] == "--reprocess":
        return reprocess(sys.argv[2])
    root, tag, target = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "sgr::" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, "a", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "sgr::" in k:
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    out = {}
    for k, v in sorted(acc.items()):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        rec = record(m, round(sum(dur[k]) / len(dur[k]), 4) if dur.get(k) else None)
        if rec is None:
            continue
        out[k] = rec
        show(k, rec)
    try:
        allrec = json.load(open(target))
    except Exception:
        allrec = {}
    allrec["_comment"] = COMMENT
    out["_csrc_sha16"] = csrc_sha16(ROOT)
    allrec[tag] = out
    json.dump(allrec, open(target, "w"), indent=1)


if __name__ == "__main__":
    main()
