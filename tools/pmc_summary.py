#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc run that collected GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS (last dispatch of each kernel)."""
import csv, glob, re, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"_ZN5icrec\d+", "", r["Kernel_Name"]).replace("void icrec::", "")[:40] + " g" + r["Grid_Size"]
    k = (int(r["Dispatch_Id"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (d, n, us), c in reversed(list(agg.items())):
    if n in seen or us < 8:
        continue
    seen.add(n)
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (us * 1e-6 * clk * 1e9) if clk else 0
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    g = lambda k: c.get(k, 0) / wc
    print(f"{n:52s} {us:8.1f}us clk~{clk:.2f}GHz mfma_busy={mf:5.1%} wait_any={g('SQ_WAIT_ANY'):5.1%} "
          f"wait_inst={g('SQ_WAIT_INST_ANY'):5.1%} wait_lds={g('SQ_WAIT_INST_LDS'):5.1%} valu={g('SQ_ACTIVE_INST_VALU'):5.1%}")
