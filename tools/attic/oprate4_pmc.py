#!/usr/bin/env python3
"""tools/oprate4_pmc.py DIR -- table of the SQ counters rocprofv3 --pmc collected over tools/oprate4 (one dispatch per
probe and waves-per-SIMD setting, in oprate4's order 1, 2, 3, 4, 6, 8).  Per dispatch: VALU wave-instructions, and per
instruction the quad-cycle counters SQ_ACTIVE_INST_VALU / SQ_WAIT_INST_ANY / SQ_WAIT_ANY / SQ_WAVE_CYCLES (x 4 = cycles,
MI355X_MICROARCH.md) and SQ_BUSY_CYCLES."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = defaultdict(dict)
order = []
for r in csv.DictReader(open(f)):
    k = int(r["Dispatch_Id"])
    if k not in rows:
        order.append(k)
    rows[k]["name"] = r["Kernel_Name"].split("(")[0]
    rows[k]["grid"] = int(r["Grid_Size"])
    rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows[k]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
W = [1, 2, 3, 4, 6, 8]
seen = defaultdict(int)
print("%-10s %3s %12s %9s | per VALU instruction, cycles (quad-cycle counters x 4): %8s %9s %9s %9s | %s" % ("probe", "w", "INSTS_VALU", "ms", "ACTIVE", "WAIT_INST", "WAIT_ANY", "WAVE_CYC", "BUSY_CYCLES/instr (per-SE sum)"))
for k in order:
    r = rows[k]
    w = r["grid"] // 256 // 256
    n = r.get("SQ_INSTS_VALU", 0.0) or 1.0
    g = lambda c: 4.0 * r.get(c, float("nan")) / n
    print("%-10s %3d %12.4g %9.3f | %62.2f %9.2f %9.2f %9.2f | %.3f" % (r["name"], w, n, r["ns"] / 1e6, g("SQ_ACTIVE_INST_VALU"), g("SQ_WAIT_INST_ANY"), g("SQ_WAIT_ANY"),
                                                                          g("SQ_WAVE_CYCLES"), r.get("SQ_BUSY_CYCLES", float("nan")) / n))
