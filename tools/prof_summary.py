#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name calls / total / avg / share.
usage: python tools/prof_summary.py <dir-or-csv> [min_duration_us]"""
import csv, glob, os, re, sys

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
agg = {}
for f in files:
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d < min_us:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void icrec::", "").replace("icrec::", "")
        key = (name, r["Grid_Size_X"], r["Grid_Size_Y"])
        a = agg.setdefault(key, [0, 0.0, r["VGPR_Count"], r["LDS_Block_Size"]])
        a[0] += 1
        a[1] += d
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':70s} {'grid':>14s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'share':>6s} vgpr lds")
for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} {gx+'x'+gy:>14s} {a[0]:6d} {a[1]/1e3:9.3f} {a[1]/a[0]:9.1f} {100*a[1]/tot:5.1f}% {a[2]} {a[3]}")
