#!/usr/bin/env python3
"""Raw per-kernel counter dump of a rocprofv3 --pmc run: every dispatch of a (kernel name, grid) averaged; the layer kernel's
launches split into the two populations of a step (with / without the next layer's QKV epilogue), as tools/pmc_summary.py does.
usage: python tools/pmc_raw.py <dir> [name-filter]"""
import csv, glob, re, collections, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"_ZN5icrec\d+", "", r["Kernel_Name"]).replace("void icrec::", "")[:44] + " g" + r["Grid_Size"]
        if flt and flt not in n:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": n, "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    groups = collections.OrderedDict()
    for d in disp.values():
        key = d["name"]
        if "ffn_fused2_kernel" in key:
            t = sorted(x["us"] for x in disp.values() if x["name"] == key)
            if t[-1] > 1.15 * t[0]:
                key += " [with next QKV]" if d["us"] > (t[0] + t[-1]) / 2 else " [last layer: no QKV]"
        groups.setdefault(key, []).append(d)
    for n, ds in groups.items():
        us = sum(d["us"] for d in ds) / len(ds)
        names = [k for k in ds[0] if k not in ("name", "us")]
        print(f"{n:70s} x{len(ds):3d} {us:9.1f}us  " + "  ".join(f"{k}={sum(d.get(k, 0.0) for d in ds) / len(ds):.4g}" for k in names))
