#!/usr/bin/env python3
"""Turns the counter passes of tools/fetch_calib (rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum, see tools/gpu_final.sh) into the table
DESIGN.md quotes: per access pattern the KNOWN useful bytes, the raw FETCH_SIZE (KiB -> bytes), the number of memory-side read requests and
bytes per request implied, and the factor raw -> useful.     tools/fetch_calib_report.py <gpurun_out dir>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
GiB = float(1 << 30)
img = 32 * 3 * 19200 * 128 * 4.0
useful = {"wide16": GiB, "dword4": GiB, "half_lines": GiB / 2, "half_far_first": GiB / 2, "half_far_second": GiB / 2,
          "dma_rows<0>": img, "dma_rows<320>": img, "dma_rows<1280>": img}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("calib_fetch", "calib_rdreq"):
    for f in glob.glob(f"{root}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'pattern':18s} {'useful MB':>10s} {'FETCH_SIZE raw MB':>18s} {'x2 / useful':>12s} {'RDREQ':>10s} {'32B':>6s} {'useful B / request':>19s}")
for k, u in useful.items():
    c = acc.get(k)
    if not c:
        continue
    mean = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else float("nan")
    raw = mean("FETCH_SIZE") * 1024.0
    rq = mean("TCC_EA0_RDREQ_sum")
    print(f"{k:18s} {u / 1e6:10.1f} {raw / 1e6:18.1f} {2 * raw / u:12.3f} {rq:10.0f} {mean('TCC_EA0_RDREQ_32B_sum'):6.0f} {u / rq:19.1f}")
print("reading: every pattern issues 128-byte requests (RDREQ x 128 = bytes moved; none of 32 B) and FETCH_SIZE tallies each at 64: the x2 of\n"
      "tools/parse_pmc.py holds for 4 B/lane, 16 B/lane, half-line and LDS-DMA row reads alike.  half_lines moves the WHOLE line for the half it\n"
      "uses (x2 / useful = 2.0; its time equals wide16's per byte moved): a half line whose other half arrives after the line left the L2 is a real\n"
      "second fetch, not a counting artefact.")
