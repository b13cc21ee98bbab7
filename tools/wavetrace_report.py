#!/usr/bin/env python3
"""Summarise tools/wavetrace output: per kernel, the span, per-SIMD busy time and how the last round was placed."""
import sys, collections
runs = collections.OrderedDict()
for line in open(sys.argv[1]):
    if line.startswith('#') or not line.strip():
        if line.startswith('#'): print(line.strip())
        continue
    f = line.split()
    k, i, t0, t1, xcc, se, sh, cu, simd = f[:9]
    tp = int(f[9]) if len(f) > 9 else 0
    cyc = int(f[10]) if len(f) > 10 else 0
    runs.setdefault(k, []).append((int(t0), int(t1), (int(xcc), int(se), int(sh), int(cu), int(simd)), tp, cyc))
for k, ws in runs.items():
    span = max(w[1] for w in ws)
    dur = sorted(w[1] - w[0] for w in ws)
    simds = collections.defaultdict(list)
    for t0, t1, s, tp, cyc in ws: simds[s].append((t0, t1))
    if any(w[4] for w in ws):      # round 5: shader clock under load = s_memtime cycles / s_memrealtime ticks (10 ns), per wave
        clk = sorted(w[4] / max(w[1] - w[0], 1) / 10.0 for w in ws)
        tot = sum(w[4] for w in ws) / max(sum(w[1] - w[0] for w in ws), 1) / 10.0
        print(f"   shader clock over the waves' lifetimes (GHz): mean {tot:.3f}  min/10%/med/90%/max {clk[0]:.3f}/{clk[len(clk)//10]:.3f}/{clk[len(clk)//2]:.3f}/{clk[9*len(clk)//10]:.3f}/{clk[-1]:.3f}")
    pro = sorted(w[3] for w in ws)
    print(f"   prologue (loads + pre-map + frame) min/med/90%/max {pro[0]/100:.1f}/{pro[len(pro)//2]/100:.1f}/{pro[9*len(pro)//10]/100:.1f}/{pro[-1]/100:.1f} us")
    n = len(simds)
    per = [len(v) for v in simds.values()]
    # concurrency profile: number of resident waves over time, sampled
    ev = sorted([(w[0], 1) for w in ws] + [(w[1], -1) for w in ws])
    print(f"{k}: waves {len(ws)}  SIMDs seen {n}  span {span/100:.1f} us  wave duration min/med/max {dur[0]/100:.1f}/{dur[len(dur)//2]/100:.1f}/{dur[-1]/100:.1f} us  waves per SIMD min/max {min(per)}/{max(per)}")
    # busy SIMD fraction in 10 slices
    slices = 12
    out = []
    for si in range(slices):
        a, b = span * si / slices, span * (si + 1) / slices
        busy = 0; two = 0
        for v in simds.values():
            ov = sum(max(0, min(t1, b) - max(t0, a)) for t0, t1 in v)
            busy += min(ov, b - a) / (b - a)
            two += ov / (b - a)
        out.append(f"{busy/n:.2f}/{two/n:.2f}")
    print("   SIMD busy fraction / mean resident waves per time slice:", ' '.join(out))
    # end times of SIMDs
    ends = sorted(max(t1 for _, t1 in v) for v in simds.values())
    print(f"   SIMD finish time percentiles (us): 10% {ends[n//10]/100:.1f}  50% {ends[n//2]/100:.1f}  90% {ends[9*n//10]/100:.1f}  max {ends[-1]/100:.1f}")
    late = [w for w in ws if w[0] > 0.6 * span]
    cnt = collections.Counter(w[2] for w in late)
    print(f"   waves starting after 60% of the span: {len(late)} on {len(cnt)} SIMDs (max {max(cnt.values()) if cnt else 0} on one SIMD)")
