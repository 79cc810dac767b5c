#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc run that collected GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS.

Every dispatch of a (kernel, grid, kernel-argument variant) is averaged.  Round 3's version printed the LAST dispatch of each
kernel only - for the layer kernel that is the last layer's launch, which has no QKV epilogue (0.89 ms where the other five
launches of a step take 1.20 ms): the figure looked like a discrepancy between the counter pass and the kernel trace.  The
layer kernel's launches are therefore split by duration class (with / without the next layer's QKV projection)."""
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"_ZN5icrec\d+", "", r["Kernel_Name"]).replace("void icrec::", "")[:40] + " g" + r["Grid_Size"]
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": n, "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.OrderedDict()
for d in disp.values():
    key = d["name"]
    if "ffn_fused2_kernel" in key:  # with / without the QKV epilogue: two populations of one symbol
        med = sorted(x["us"] for x in disp.values() if x["name"] == key)
        split = (med[0] + med[-1]) / 2
        if med[-1] > 1.15 * med[0]:
            key += " [with next QKV]" if d["us"] > split else " [last layer: no QKV]"
    groups.setdefault(key, []).append(d)
for n, ds in groups.items():
    us = sum(d["us"] for d in ds) / len(ds)
    if us < 8:
        continue
    avg = lambda k: sum(d.get(k, 0.0) for d in ds) / len(ds)
    clk = avg("GRBM_GUI_ACTIVE") / 8 / us / 1e3
    mf = avg("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (us * 1e-6 * clk * 1e9) if clk else 0
    wc = avg("SQ_WAVE_CYCLES") or 1
    g = lambda k: avg(k) / wc
    print(f"{n:74s} x{len(ds):3d} {us:8.1f}us clk~{clk:.2f}GHz mfma_busy={mf:5.1%} wait_any={g('SQ_WAIT_ANY'):5.1%} "
          f"wait_inst={g('SQ_WAIT_INST_ANY'):5.1%} wait_lds={g('SQ_WAIT_INST_LDS'):5.1%} valu={g('SQ_ACTIVE_INST_VALU'):5.1%}")
