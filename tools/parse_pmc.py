"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

Units / corrections (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> the read side is doubled.  WRITE_SIZE is used as reported.
Writes <dir>/traffic_raw.json: {kernel: {"fetch_bytes": .., "write_bytes": .., "hbm_bytes": ..}}; with a workload tag
and a target file as 2nd / 3rd argument the table is also merged into that file under the tag
(profiles/traffic.json: {"config2_batch16_env": {kernel: ..}, ..} -- what bench.py's roofline.traffic reads)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root):
    """Hash of the kernel sources (csrc/*.hip, *.inl, *.h): counter records are stamped with it, and bench.py reports a record as
    stale when the sources it was measured on are not the ones the loaded library was built from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(root, "inverserenderingofindoorscene_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inl")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


root = sys.argv[1]
acc = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for cname in acc:
    for f in glob.glob(os.path.join(root, cname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == cname:
                acc[cname][row["Kernel_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    if "sgr::" not in k:
        continue
    fk = acc["FETCH_SIZE"].get(k, [])
    wk = acc["WRITE_SIZE"].get(k, [])
    fetch = 2.0 * 1024.0 * sum(fk) / max(len(fk), 1)          # KiB -> B, x2 gfx950 correction
    write = 1024.0 * sum(wk) / max(len(wk), 1)
    short = k.split("(")[0].replace("void ", "")
    out[short] = {"fetch_bytes": round(fetch), "write_bytes": round(write), "hbm_bytes": round(fetch + write),
                  "launches_sampled": [len(fk), len(wk)]}
    print(f"{short:70s} fetch {fetch/1e6:9.1f} MB  write {write/1e6:9.1f} MB  total {(fetch+write)/1e6:9.1f} MB per launch")
json.dump(out, open(os.path.join(root, "traffic_raw.json"), "w"), indent=1)
if len(sys.argv) > 3:
    tag, target = sys.argv[2], sys.argv[3]
    try:
        allrec = json.load(open(target))
    except Exception:
        allrec = {}
    out["_csrc_sha16"] = csrc_sha16(ROOT)
    allrec[tag] = out
    json.dump(allrec, open(target, "w"), indent=1)
