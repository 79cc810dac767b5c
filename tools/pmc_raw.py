#!/usr/bin/env python3
"""Raw per-kernel counter dump of a rocprofv3 --pmc run (last dispatch of each kernel name + grid):
usage: python tools/pmc_raw.py <dir> [name-filter]"""
import csv, glob, re, collections, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"_ZN5icrec\d+", "", r["Kernel_Name"]).replace("void icrec::", "")[:44] + " g" + r["Grid_Size"]
        if flt and flt not in n:
            continue
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(n, {})[r["Counter_Name"]] = (float(r["Counter_Value"]), us)
    for n, c in agg.items():
        us = next(iter(c.values()))[1]
        print(f"{n:60s} {us:9.1f}us  " + "  ".join(f"{k}={v[0]:.4g}" for k, v in c.items()))
