#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a bench run into profiles/r02_roofline_traffic.json:
per kernel key, average HBM-side bytes per launch.  On gfx950 FETCH_SIZE counts 64-B requests as 32 B: doubled
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Units of both counters: KB (x1024 B).
usage: python tools/pmc_traffic_json.py <fetch_dir> <write_dir> <out.json> key=substring[,grid] ..."""
import csv, glob, json, sys

def avg(d, counter, sub, grid):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and sub in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid):
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)

fetch_dir, write_dir, out = sys.argv[1:4]
rec = {}
for spec in sys.argv[4:]:
    key, sub = spec.split("=", 1)
    sub, _, grid = sub.partition(",")
    f, nf = avg(fetch_dir, "FETCH_SIZE", sub, grid or None)
    w, nw = avg(write_dir, "WRITE_SIZE", sub, grid or None)
    if f is None or w is None:
        continue
    rec[key] = {"traffic_bytes": 2 * f * 1024 + w * 1024, "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
                "launches": [nf, nw],
                "note": f"PMC per launch: 2*FETCH_SIZE {2 * f * 1024 / 1e6:.0f} MB + WRITE_SIZE {w * 1024 / 1e6:.0f} MB"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
