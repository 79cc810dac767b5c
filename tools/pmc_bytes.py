#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE (KB per launch) per kernel+grid from `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` output directories.  On gfx950 FETCH_SIZE counts 64-B units as 32 B: double it
(calibrated on add_ln_kernel, see profiles/r01_pmc_hbm_bytes.txt); WRITE_SIZE is exact.
usage: python tools/pmc_bytes.py <dir> [<dir> ...]"""
import csv, glob, re, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            n = re.sub(r"_ZN5icrec\d+", "", r["Kernel_Name"]).replace("void icrec::", "")[:58] + " grid=" + r["Grid_Size"]
            agg.setdefault((r["Counter_Name"], n), []).append(float(r["Counter_Value"]))
        cur = None
        for (c, n), v in sorted(agg.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
            if sum(v) / len(v) < 1000:
                continue
            if c != cur:
                print(c)
                cur = c
            print(f"  {n:82s} n={len(v):3d} avg_KB={sum(v) / len(v):12.1f}")
