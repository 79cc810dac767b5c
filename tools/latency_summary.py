#!/usr/bin/env python3
"""Summarise the last single-request iteration of a `rocprofv3 --kernel-trace -- python3 tools/latency_trace.py` run:
per kernel family, launches and microseconds; plus the span of the iteration.
usage: python tools/latency_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "embed_ln" in r["Kernel_Name"]]
it = rows[idx[-2]:idx[-1]]
fam = collections.OrderedDict()
for r in it:
    n = r["Kernel_Name"]
    for key in ("embed_ln", "linear_x3", "linear_kernel", "attention", "add_ln", "pool_norm", "normalize_rows", "stream_search", "search_kernel", "merge_kernel"):
        if key in n:
            break
    else:
        key = n[:30]
    d = fam.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, us) in fam.items():
    print(f"{k:16s} x{c:3d} {us:8.1f} us  ({us / c:5.1f} each)")
print(f"kernels {len(it)}  span {(int(rows[idx[-1]]['Start_Timestamp']) - int(it[0]['Start_Timestamp'])) / 1e3:.1f} us")
